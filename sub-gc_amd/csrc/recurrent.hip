// The teacher-forced recurrence of the train decoder issued from C (reference loop: AttModel.py:157-175 around TopDownCore.forward,
// :400-431).  Nothing here launches a kernel of its own: both entry points walk the steps and call the per-step entry points of
// decoder.hip / attention_group.hip / gemm_*.hip exactly as the host-language loop did (functions.py / functions_packed.py keep that
// loop for scheduled sampling and for the FLOP-accounting pass) -- one library crossing per direction instead of ~10 per step.
#include "common.h"

namespace {

struct Win {            // a column window of a plane stack: see subgc_lstm_bwd_planes
    const float* p;
    int64_t ld;
    int n;
    int64_t stride;
    int rows;
};

inline const char* at(const void* base, int64_t elems, int esz) { return static_cast<const char*>(base) + elems * esz; }
inline char* at(void* base, int64_t elems, int esz) { return static_cast<char*>(base) + elems * esz; }

int check_common(const SubgcRecurrence* a, const char* what) {
    SUBGC_REQUIRE(a, "%s: null argument block", what);
    SUBGC_REQUIRE(a->S > 0 && a->T >= 0 && a->T <= 4096 && a->R > 0 && a->A > 0 && a->n_alpha > 0, "%s: bad sizes", what);
    SUBGC_REQUIRE(a->T == 0 || (a->m && a->row0), "%s: null step tables", what);
    SUBGC_REQUIRE(a->T == 0 || (a->n_m >= a->T + 1 && a->n_row0 >= a->T + 1), "%s: m[] and row0[] need T + 1 = %d entries (got %d, %d)", what,
                  a->T + 1, a->n_m, a->n_row0);
    SUBGC_REQUIRE(!a->shared || (a->rows_map && a->B > 0 && a->g > 0 && a->Nn > 0), "%s: shared attention sets need rows_map, B, g, Nn", what);
    SUBGC_REQUIRE(a->shared || a->off, "%s: per-sentence attention sets need their offsets", what);
    for (int t = 0; t < a->T; ++t)
        SUBGC_REQUIRE(a->m[t] > 0 && a->m[t] <= a->S && (t == 0 || a->m[t] <= a->m[t - 1]) && a->row0[t] >= 0,
                      "%s: live rows must be positive, non-increasing and <= S (step %d: %d)", what, t, a->m[t]);
    return SUBGC_OK;
}

}  // namespace

SUBGC_API int subgc_recurrence_sizeof(void) { return (int)sizeof(SubgcRecurrence); }

namespace {
int check_fwd(const SubgcRecurrence* a) {
    if (int rc = check_common(a, "recurrence_fwd")) return rc;
    if (a->T == 0) return SUBGC_OK;
    SUBGC_REQUIRE(a->H1 && a->H2 && a->Hout && a->Wc1 && a->Wc2 && a->Wq && a->pre && a->Gx && a->Gf && a->C1 && a->C2 && a->G1 && a->G2 && a->AH &&
                      a->AL && a->QP && a->u && a->v && a->w_a && a->b_a && a->lens && a->hout_off,
                  "recurrence_fwd: null pointer");
    return SUBGC_OK;
}

// one time step of the forward recurrence (all launches of step t on `stream`)
int fwd_step(const SubgcRecurrence* a, int t, void* workspace, size_t ws_bytes, void* stream) {
    const int R = a->R, A = a->A, S = a->S, esz = a->bf16 ? 2 : 4;
    const int64_t R4 = 4 * (int64_t)R;
    const int cell_bits = (a->bf16 ? 1 : 0) | (a->bf16 ? 2 : 0);          // bit 0: x and w are bf16, bit 1: the h destinations are bf16
    const int attn_bits = (a->bf16 ? 1 : 0) | (a->uv_b16 ? 2 : 0);        // bit 0: ctx destination bf16, bit 1: u / v bf16
    const int m = a->m[t];
    const int mn = a->m[t + 1] > 0 ? a->m[t + 1] : 1;                 // row limit 0 would mean "all": one dummy row goes to the slack
    const int64_t o = a->row0[t], o1 = a->row0[t + 1];
    char* const H1t = at(a->H1, o * a->ldH1, esz);
    char* const H2t = at(a->H2, o * a->ldH2, esz);
    char* const H1n = at(a->H1, o1 * a->ldH1, esz);
    char* const H2n = at(a->H2, o1 * a->ldH2, esz);
    float* const C1p = a->C1 + (int64_t)t * S * R;
    float* const C2p = a->C2 + (int64_t)t * S * R;
    // attention LSTM: [h2_{t-1} | h1_{t-1}] . Wc1^T + x->gates + fc->gates; h1_t -> H2[t][:, R:2R] and the next step's H1[:, R:]
    int rc = subgc_lstm_fwd_gemm(H1t, a->ldH1, a->Wc1, a->ldW1, 2 * R, a->pre, R4, a->Gx + o * R4, R4, a->Gf, R4, a->b1i, a->b1h, C1p,
                                 C1p + (int64_t)S * R, H2t + (int64_t)R * esz, a->ldH2, H1n + (int64_t)R * esz, a->ldH1, nullptr, 1.f, nullptr, 0,
                                 a->G1 + o * R4, m, R, m, mn, cell_bits, a->gemm_flags, workspace, ws_bytes, stream);
    if (rc != SUBGC_OK) return rc;
    // attention query h2att(h1_t): left as split-K planes, summed (+ bias) by the attention kernel into AH
    int nq = 0;
    rc = a->bf16 ? subgc_gemm_bf16_planes(0, 1, m, A, R, reinterpret_cast<const uint16_t*>(H2t + (int64_t)R * esz), a->ldH2,
                                          static_cast<const uint16_t*>(a->Wq), a->ldWq, a->QP, a->qp_bytes, &nq, stream)
                 : subgc_gemm_f32_planes(0, 1, m, A, R, reinterpret_cast<const float*>(H2t + (int64_t)R * esz), a->ldH2,
                                         static_cast<const float*>(a->Wq), a->ldWq, a->QP, a->qp_bytes, &nq, a->gemm_flags, stream);
    if (rc != SUBGC_OK) return rc;
    const int64_t sq = (int64_t)m * A;
    rc = a->shared ? subgc_attn_fwd_group_q(a->u, a->v, a->QP, nq, sq, a->bq, a->AH + o * A, a->w_a, a->b_a, a->rows_map, a->lens, m, a->B, a->g,
                                            a->Nn, H2t, a->ldH2, a->AL + o * a->n_alpha, a->n_alpha, A, R, attn_bits, stream)
                   : subgc_attn_fwd_q(a->u, a->v, a->QP, nq, sq, a->bq, a->AH + o * A, a->w_a, a->b_a, a->off, a->lens, H2t, a->ldH2,
                                      a->AL + o * a->n_alpha, a->n_alpha, m, A, R, attn_bits, stream);
    if (rc != SUBGC_OK) return rc;
    // language LSTM: [ctx_t | h1_t | h2_{t-1}] . Wc2^T; h2_t -> the next step's H1[:, :R] and H2[:, 2R:], dropout(h2_t) -> Hout
    rc = subgc_lstm_fwd_gemm(H2t, a->ldH2, a->Wc2, a->ldW2, 3 * R, a->pre, R4, nullptr, 0, nullptr, 0, a->b2i, a->b2h, C2p, C2p + (int64_t)S * R,
                             H1n, a->ldH1, H2n + 2 * (int64_t)R * esz, a->ldH2, a->k_out ? a->k_out + (int64_t)t * S * R : nullptr, a->keep_scale,
                             at(a->Hout, a->hout_off[t], esz), a->ld_hout, a->G2 + o * R4, m, R, mn, mn, cell_bits, a->gemm_flags, workspace,
                             ws_bytes, stream);
    if (rc != SUBGC_OK) return rc;
    return SUBGC_OK;
}
}  // namespace

SUBGC_API int subgc_recurrence_fwd(const SubgcRecurrence* a, void* workspace, size_t ws_bytes, void* stream) {
    if (int rc = check_fwd(a)) return rc;
    for (int t = 0; t < a->T; ++t)
        if (int rc = fwd_step(a, t, workspace, ws_bytes, stream)) return rc;
    return SUBGC_OK;
}


namespace {
int check_bwd(const SubgcRecurrence* a) {
    if (int rc = check_common(a, "recurrence_bwd")) return rc;
    if (a->T == 0) return SUBGC_OK;
    SUBGC_REQUIRE(a->Wc1 && a->Wc2 && a->Wq && a->C1 && a->C2 && a->G1 && a->G2 && a->AH && a->AL && a->u && a->v && a->w_a && a->lens && a->dHout &&
                      a->dhout_off && a->dP1 && a->dP2 && a->dAH && (a->du || a->dE) && a->dWa && a->dBa && a->PA && a->PB && a->PC && a->dC1_in &&
                      a->dC1_out && a->dC2_in && a->dC2_out,
                  "recurrence_bwd: null pointer");
    SUBGC_REQUIRE(!a->dE || !a->shared, "recurrence_bwd: deferred d(u) (dE) is the per-sentence form");
    SUBGC_REQUIRE(a->dv || a->dCtx, "recurrence_bwd: either dv (accumulated per step) or dCtx (kept for one dv_accum pass)");
    SUBGC_REQUIRE(!a->shared || !a->dv, "recurrence_bwd: shared attention sets always defer d(v)");
    return SUBGC_OK;
}

struct BwdState {       // what one backward step hands to the next (earlier) one
    float *c1_in, *c1_out, *c2_in, *c2_out;
    Win sA, sC;         // planes of the previous (later) step's dP2.Wc2 / dP1.Wc1
};

BwdState bwd_begin(const SubgcRecurrence* a) {
    return BwdState{a->dC1_in, a->dC1_out, a->dC2_in, a->dC2_out, Win{nullptr, 0, 0, 0, 0}, Win{nullptr, 0, 0, 0, 0}};
}

// one time step of the backward recurrence (all launches of step t on `stream`)
int bwd_step(const SubgcRecurrence* a, BwdState& st, int t, void* stream) {
    const int R = a->R, A = a->A, S = a->S, esz = a->bf16 ? 2 : 4;
    const int64_t R4 = 4 * (int64_t)R;
    const int dbits = (a->bf16 ? 1 : 0) | (a->uv_b16 ? 2 : 0);            // bit 0: dah destination bf16, bit 1: u / v bf16
    auto win = [](const Win& w, int64_t col0) { return Win{w.n ? w.p + col0 : nullptr, w.ld, w.n, w.stride, w.rows}; };
    auto planes = [&](const void* dy, int64_t ld_dy, int K, const void* W, int64_t ldw, int N, int m, float* buf, size_t bytes, int* n) {
        return a->bf16 ? subgc_gemm_bf16_planes(0, 0, m, N, K, static_cast<const uint16_t*>(dy), ld_dy, static_cast<const uint16_t*>(W), ldw, buf, bytes, n,
                                                stream)
                       : subgc_gemm_f32_planes(0, 0, m, N, K, static_cast<const float*>(dy), ld_dy, static_cast<const float*>(W), ldw, buf, bytes, n,
                                               a->gemm_flags, stream);
    };
    const int m = a->m[t];
    const int64_t o = a->row0[t];
    const float* C1p = a->C1 + (int64_t)t * S * R;
    const float* C2p = a->C2 + (int64_t)t * S * R;
    char* const dP2t = at(a->dP2, o * R4, esz);
    char* const dP1t = at(a->dP1, o * R4, esz);
    char* const dAHt = at(a->dAH, o * A, esz);
    // language cell: d(h2_t) = d(Hout_t) through the dropout mask + the later step's d(h2_prev) windows
    Win w0 = win(st.sC, 0), w1 = win(st.sA, 2 * (int64_t)R);
    {
        Win src[3] = {{nullptr, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0}};
        int k = 0;
        if (w0.n > 0) src[k++] = w0;
        if (w1.n > 0) src[k++] = w1;
        int rc = subgc_lstm_bwd_planes(a->G2 + o * R4, C2p, C2p + (int64_t)S * R, src[0].p, src[0].ld, src[0].n, src[0].stride, src[0].rows, src[1].p,
                                       src[1].ld, src[1].n, src[1].stride, src[1].rows, src[2].p, src[2].ld, src[2].n, src[2].stride, src[2].rows,
                                       a->dHout + a->dhout_off[t], a->ld_dhout, a->k_out ? a->k_out + (int64_t)t * S * R : nullptr, a->keep_scale,
                                       st.c2_in, dP2t, st.c2_out, m, R, a->bf16, stream);
        if (rc != SUBGC_OK) return rc;
    }
    int n = 0;
    int rc = planes(dP2t, R4, (int)R4, a->Wc2, a->ldW2, 3 * R, m, a->PA, a->pa_bytes, &n);      // -> [dctx | dh1 | dh2_prev]
    if (rc != SUBGC_OK) return rc;
    st.sA = Win{a->PA, 3 * (int64_t)R, n, (int64_t)m * 3 * R, m};
    float* const keep = a->dv ? nullptr : a->dCtx + o * R;
    rc = a->shared ? subgc_attn_bwd_group(a->u, a->v, a->AH + o * A, a->w_a, a->rows_map, a->lens, m, a->B, a->g, a->Nn, a->AL + o * a->n_alpha,
                                          a->n_alpha, st.sA.p, st.sA.ld, dAHt, a->du, a->dWa + o * A, a->dBa + o, A, R, dbits, keep, R, st.sA.n, st.sA.stride,
                                          a->du_planes, a->du_plane_stride, stream)
                   : a->dE ? subgc_attn_bwd_planes_de(a->u, a->v, a->AH + o * A, a->w_a, a->off, a->lens, a->AL + o * a->n_alpha, a->n_alpha, st.sA.p, st.sA.ld,
                                                      st.sA.n, st.sA.stride, dAHt, a->dE + o * a->n_alpha, a->dv, a->dWa + o * A, a->dBa + o, m, A, R, dbits,
                                                      keep, R, stream)
                           : subgc_attn_bwd_planes(a->u, a->v, a->AH + o * A, a->w_a, a->off, a->lens, a->AL + o * a->n_alpha, a->n_alpha, st.sA.p, st.sA.ld,
                                                   st.sA.n, st.sA.stride, dAHt, a->du, a->dv, a->dWa + o * A, a->dBa + o, m, A, R, dbits, keep, R, stream);
    if (rc != SUBGC_OK) return rc;
    int nb = 0;
    rc = planes(dAHt, A, A, a->Wq, a->ldWq, R, m, a->PB, a->pb_bytes, &nb);                      // h1_t also feeds the attention query
    if (rc != SUBGC_OK) return rc;
    {
        Win src[3] = {{nullptr, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0}, {nullptr, 0, 0, 0, 0}};
        int k = 0;
        Win x0 = win(st.sA, R), x1{a->PB, (int64_t)R, nb, (int64_t)m * R, m}, x2 = win(st.sC, R);
        if (x0.n > 0) src[k++] = x0;
        if (x1.n > 0) src[k++] = x1;
        if (x2.n > 0) src[k++] = x2;
        rc = subgc_lstm_bwd_planes(a->G1 + o * R4, C1p, C1p + (int64_t)S * R, src[0].p, src[0].ld, src[0].n, src[0].stride, src[0].rows, src[1].p,
                                   src[1].ld, src[1].n, src[1].stride, src[1].rows, src[2].p, src[2].ld, src[2].n, src[2].stride, src[2].rows, nullptr,
                                   0, nullptr, 1.f, st.c1_in, dP1t, st.c1_out, m, R, a->bf16, stream);
        if (rc != SUBGC_OK) return rc;
    }
    rc = planes(dP1t, R4, (int)R4, a->Wc1, a->ldW1, 2 * R, m, a->PC, a->pc_bytes, &n);          // -> [dh2_prev | dh1_prev]
    if (rc != SUBGC_OK) return rc;
    st.sC = Win{a->PC, 2 * (int64_t)R, n, (int64_t)m * 2 * R, m};
    std::swap(st.c1_in, st.c1_out);
    std::swap(st.c2_in, st.c2_out);
    return SUBGC_OK;
}
}  // namespace

SUBGC_API int subgc_recurrence_bwd(const SubgcRecurrence* a, void* stream) {
    if (int rc = check_bwd(a)) return rc;
    BwdState st = bwd_begin(a);
    for (int t = a->T - 1; t >= 0; --t)
        if (int rc = bwd_step(a, st, t, stream)) return rc;
    return SUBGC_OK;
}

