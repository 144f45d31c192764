// The ROW-LOCAL middle of a train-decoder step as ONE launch (reference: TopDownCore.forward, AttModel.py:400-431, and
// Attention.forward, :445-471, at the batch sizes of training).  OPT-IN (ops.FUSE_MID / SubgcRecurrence.fuse_mid): correct, measured
// SLOWER than the three launches it replaces -- kept as the evidence behind that statement (profiles/r05_mid_probe.txt).
//
// A step of the recurrence is  product -> cell -> query product -> attention -> product -> cell.  The two gate products are
// all-to-all over the hidden dimension (every output column needs the whole input row of every sentence) and stay chip-wide GEMM
// launches.  Everything BETWEEN them is local to a sentence row: the attention LSTM's cell update of row s needs the gate
// pre-activations of row s, the query h2att(h1[s]) needs h1[s], the attention of sentence s needs its query and its own node set,
// and the context goes to row s of the next product's operand.  Cutting the forward step at the all-to-all seams only gives
//
//     [gate product 1] [cell 1 + query + attention]  [gate product 2] [cell 2]                                   6 -> 4 launches
//
// and the split-K planes of the 512-wide query product, the query rows' round trip and two launch boundaries per step disappear.
// A workgroup (16 waves) owns RB consecutive sentence rows (RB = ceil(rows / 256): one workgroup per CU, all CUs busy) and walks the
// phases with its rows' state in LDS:
//   1. cell: gate planes + x->gates + fc->gates + biases -> (i, f, g, o), c, h1 -- h1 to its two operand slots, the gates to G1 and h1 to
//      LDS (the query product's operand: fp32, or bf16 under compute_dtype = bf16 with the rounding the operand slot gets);
//   2. query: q = h1 Wq^T + b.  bf16 operands: the matrix pipe as a reduction engine (gemm_skinny.hip's scheme) -- the <= 16 rows are ONE
//      MFMA operand tile, a wave owns 2 x 16 query columns over the whole K, no cross-wave reduction.  fp32 operands: plain FMAs on the
//      K-MAJOR weight (the fp32 matrix pipe runs at the VALU's rate and would spend 80 % of it on the padding of a 16-row tile).
//      Either way every workgroup streams the WHOLE weight (1 MB bf16 / 2 MB fp32) from L2;
//   3. attention over the rows' node sets as attention_vec.hip does it (same arithmetic order), the (sentence, node) pairs of all RB
//      rows flattened so that every wave has four rows of u in flight; the context goes to the product operand slot.
// What the stamps say (tools/mid_probe.py): phase 1 is bound by the CHIP's memory bandwidth (159 MB per step at 1280 rows: 28 us fused or
// not); phase 2 saturates the L2s (256 CUs x the same megabytes = ~13 TB/s aggregate: 20 us bf16 / 33 us fp32, against 15 / 19 us for the
// chip-wide split-K product); phase 3 of a workgroup alone on its CU is no faster than the 1280-workgroup launch.  The backward
// (attention backward -> d(query) Wq -> cell backward) has the same three problems and was not built.
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MID_WAVES = 16, MID_THREADS = MID_WAVES * 64;      // 4 waves per SIMD: every phase is a chain of dependent memory / VALU latencies
constexpr int MID_MAXRB = 16;             // rows of a workgroup (bf16: the rows of the one MFMA operand tile)
constexpr int MID_MAXRB_F32 = 4;          // fp32 operands: the product runs on the VALU with RB accumulator sets per thread
constexpr int MAXLEN = 512;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void fma4(float4& a, float s, const float4 b) { a.x += s * b.x; a.y += s * b.y; a.z += s * b.z; a.w += s * b.w; }
template <bool UV16>
__device__ __forceinline__ float4 ldx(const void* base, int64_t i) {
    return UV16 ? subgc_load4_bf(static_cast<const uint16_t*>(base) + i) : ld4(static_cast<const float*>(base) + i);
}
// gate non-linearities on the hardware exp / rcp units (as subgc_tanh: absolute error <= 5e-7); the stand-alone cell kernels use libm --
// inside this launch a workgroup is alone on its CU and libm's ~40-instruction chains are exposed
__device__ __forceinline__ float fast_sigmoid(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }

__device__ __forceinline__ size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
__device__ __forceinline__ void stamp(long long* st, int k) {
    if (st && threadIdx.x == 0) st[(size_t)blockIdx.x * 8 + k] = (long long)wall_clock64();
}

// ---- the skinny product of phase 2 (both directions), bf16 operands: out[m, n] = sum_k act[m, k] W[n, k] for m < 16 rows held in LDS ---
// W rows are K-contiguous ([N, K], leading dimension ldw: Wq for the forward, WqT for the backward), K is walked in 32-wide steps,
// KP = K rounded up to 32 (the LDS rows are zero there and the W addresses clamped: a padded step adds 0 x finite).  The matrix pipe as a
// reduction engine (gemm_skinny.hip): lane (r16 = lane % 16, kq = lane / 16) holds W row n0 + r16 and activation row min(r16, nv - 1), k =
// k0 + kq * 8 .. + 7; result acc[i][v] = out[m = r16][n = tile i * 16 + 4 kq + v].  A wave owns NT tiles over the whole K (no cross-wave
// reduction).  Every workgroup of the launch streams the SAME W: the waves start their K walk at different steps (rot) so that the
// chip's requests of one instant spread over the row instead of hammering the same lines of the same L2 channels.
template <int NT, int D>
__device__ __forceinline__ void skinny_mfma_b16(const uint16_t* W, int64_t ldw, int N, int K, int KP, const unsigned char* hs, int hp_bytes, int nv,
                                                int tile0, int rot, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const int steps = KP >> 5;
    const uint16_t* wrow[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) wrow[i] = W + (int64_t)min((tile0 + i) * 16 + r16, N - 1) * ldw;      // tiles past N: clamped, never stored
    const unsigned char* arow = hs + (size_t)min(r16, nv - 1) * hp_bytes;
    float4 ring[D][NT];
    auto step_of = [&](int i) { int s = i + rot; return s >= steps ? s - steps : s; };
    auto issue = [&](int slot, int i) {
        const int k = step_of(i) * 32 + kq * 8, kc = k < K ? k : 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) ring[slot][j] = *reinterpret_cast<const float4*>(wrow[j] + kc);
    };
    auto consume = [&](int slot, int i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ((size_t)step_of(i) * 32 + kq * 8) * 2);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot][j]), a, acc[j], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < steps) issue(d, d);
    int st = 0;
    for (; st + D <= steps; st += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            consume(d, st + d);
            if (st + d + D < steps) issue(d, st + d + D);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (st + d < steps) consume(d, st + d);
}

// ---- the same product with fp32 operands, on the VALU: out[r, n] = sum_k act[r, k] W[k, n] for r < RB <= 4 rows held in LDS --------------
// The fp32 matrix pipe runs at the VALU's rate (64 FLOP / clk / SIMD), and with 3 rows in a 16-row operand tile it would waste 80 % of it:
// plain FMAs on the rows that exist are 4-5 x faster here.  W is K-MAJOR ([K, N], leading dimension ldw: WqT for the forward, Wq for the
// backward) so that a wave reads 1 KB contiguous per instruction: thread (cq = t % NQP, kg = t / NQP) owns the four columns 4 cq .. 4 cq + 3
// and every KG-th block of four k; the KG partial sums meet in LDS (part [KG][RB][NP]) and are added in kg order.  k blocks start at a
// workgroup-dependent offset (the hot-spot argument above).
template <int NQP>                        // column quads per k-group, padded to whole waves: 128 (N <= 512) or 256 (N <= 1024)
__device__ __forceinline__ void skinny_valu_f32(const float* W, int64_t ldw, int N, int K, const float* hs, int hp, int nv, float* part, int NP, int rot) {
    constexpr int KG = MID_THREADS / NQP, D = 2;
    const int t = threadIdx.x, cq = t % NQP, kg = t / NQP;
    const int nkb = K >> 2, rounds = (nkb + KG - 1) / KG;                 // K % 4 == 0; every k-group walks `rounds` blocks (its last may not exist)
    const int rot0 = rot % rounds;
    const bool live = cq * 4 < N;
    const float* wc = W + (live ? cq * 4 : 0);
    float4 acc[MID_MAXRB_F32];
#pragma unroll
    for (int r = 0; r < MID_MAXRB_F32; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ring[D][4];
    auto block_of = [&](int i) { int p = i + rot0; if (p >= rounds) p -= rounds; return kg + p * KG; };
    auto issue = [&](int slot, int i) {
        const int raw = block_of(i), kb = raw < nkb ? raw : kg;              // a block past the end: a valid address, its product is dropped
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ring[slot][kk] = ld4(wc + (int64_t)(kb * 4 + kk) * ldw);
    };
    auto consume = [&](int slot, int i) {
        const int raw = block_of(i);
        const bool in = raw < nkb;
        const int kb = in ? raw : kg;
#pragma unroll
        for (int r = 0; r < MID_MAXRB_F32; ++r) {
            if (r < nv) {
                float4 h = *reinterpret_cast<const float4*>(hs + (size_t)r * hp + kb * 4);
                if (!in) h = make_float4(0.f, 0.f, 0.f, 0.f);
                fma4(acc[r], h.x, ring[slot][0]); fma4(acc[r], h.y, ring[slot][1]); fma4(acc[r], h.z, ring[slot][2]); fma4(acc[r], h.w, ring[slot][3]);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < rounds) issue(d, d);
    int st = 0;
    for (; st + D <= rounds; st += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            consume(d, st + d);
            if (st + d + D < rounds) issue(d, st + d + D);
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (st + d < rounds) consume(d, st + d);
    if (live) {
#pragma unroll
        for (int r = 0; r < MID_MAXRB_F32; ++r)
            if (r < nv) st4(part + ((size_t)kg * MID_MAXRB_F32 + r) * NP + cq * 4, acc[r]);
    }
}
template <int NQP>
__device__ __forceinline__ float4 valu_partial_sum(const float* part, int NP, int r, int c4) {        // sum over the k-groups, in order
    constexpr int KG = MID_THREADS / NQP;
    float4 s = ld4(part + ((size_t)0 * MID_MAXRB_F32 + r) * NP + c4 * 4);
#pragma unroll
    for (int g = 1; g < KG; ++g) add4(s, ld4(part + ((size_t)g * MID_MAXRB_F32 + r) * NP + c4 * 4));
    return s;
}

struct MidFwd {
    // cell (lstm_fwd_kernel's arguments)
    const float* g0; int64_t ld0; int parts; int64_t plane;
    const float* g1; int64_t ld1; const float* g2; int64_t ld2; const float* b0; const float* b1;
    const float* c_prev; float* c;
    void* h; int64_t ldh; int rows_h; void* h2; int64_t ldh2; int rows_h2;
    float* gates;
    // query product: bf16 -> Wq [A, R] (K-contiguous rows), fp32 -> WqT [R, A] (K-major)
    const void* Wq; int64_t ldw; const float* bq; float* q_out;
    // attention (attn_fwd_vec_kernel's arguments)
    const void* u; const void* v; const float* w_a; const float* b_a; const int32_t* off; const int32_t* len;
    void* ctx; int64_t ldctx; float* alpha; int n_stride;
    int m, R, A, RB, KP, HP;             // HP: LDS row pitch of the h rows in ELEMENTS (KP + pad)
    long long* stamps;                   // debug (tools/mid_probe.py): [workgroups][8] wall-clock stamps at the phase boundaries, or NULL
};

template <bool B16, bool UV16>
__global__ __launch_bounds__(MID_THREADS) void mid_fwd_kernel(const MidFwd a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int s0 = blockIdx.x * a.RB;
    const int nv = min(a.RB, a.m - s0);                                       // this workgroup's rows (>= 1)
    const int R = a.R, A = a.A, RV = R >> 2, A4 = A >> 2, R4 = R >> 2;
    constexpr int ESZ = B16 ? 2 : 4;
    unsigned char* const hs = lds;                                            // [RB][HP] h1 rows (fp32 / bf16), zero in [R, KP)
    float* const qs = reinterpret_cast<float*>(lds + al16((size_t)a.RB * a.HP * ESZ));     // [RB][A] query rows
    float* const es = qs + (size_t)a.RB * A;                                  // [RB][n_stride] scores, then attention weights
    int* const pre = reinterpret_cast<int*>(es + (size_t)a.RB * a.n_stride);  // [RB + 1] prefix of the set lengths, then [RB] first rows
    int* const m0s = pre + MID_MAXRB + 1;
    float* const part = reinterpret_cast<float*>(lds + al16((size_t)((unsigned char*)(m0s + MID_MAXRB) - lds)));   // fp32: [KG][4][A] partial sums

    stamp(a.stamps, 0);
    if (t <= a.RB) {                                                          // set lengths of the rows (attention phase)
        int acc = 0;
        for (int r = 0; r < t; ++r) acc += r < nv ? min(a.len[s0 + r], MAXLEN) : 0;
        pre[t] = acc;
        if (t < a.RB) m0s[t] = t < nv ? a.off[s0 + t] : 0;
    }
    for (int e = t; e < nv * (a.KP - R); e += MID_THREADS) {                  // zero tail of the LDS rows: columns [R, KP)
        const int r = e / (a.KP - R), c = R + e % (a.KP - R);
        if (B16) reinterpret_cast<uint16_t*>(hs)[(size_t)r * a.HP + c] = 0;
        else reinterpret_cast<float*>(hs)[(size_t)r * a.HP + c] = 0.f;
    }

    // ---- phase 1: the cell update of rows s0 .. s0 + nv - 1 (lstm_fwd_kernel<4>'s sums in its order; sigmoid / tanh on the fast units)
    for (int idx = t; idx < nv * RV; idx += MID_THREADS) {
        const int r = idx / RV, j = (idx - r * RV) << 2, s = s0 + r;
        const int64_t q = (int64_t)s * R + j;
        float4 pa[4];
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.c_prev) cp = ld4(a.c_prev + q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = k * R + j;
            float4 x = ld4(a.g0 + (int64_t)s * a.ld0 + col);
            for (int pt = 1; pt < a.parts; ++pt) add4(x, ld4(a.g0 + pt * a.plane + (int64_t)s * a.ld0 + col));
            if (a.g1) add4(x, ld4(a.g1 + (int64_t)s * a.ld1 + col));
            if (a.g2) add4(x, ld4(a.g2 + (int64_t)s * a.ld2 + col));
            if (a.b0) add4(x, ld4(a.b0 + col));
            if (a.b1) add4(x, ld4(a.b1 + col));
            pa[k] = x;
        }
        const float pi[4] = {pa[0].x, pa[0].y, pa[0].z, pa[0].w}, pf[4] = {pa[1].x, pa[1].y, pa[1].z, pa[1].w};
        const float pg[4] = {pa[2].x, pa[2].y, pa[2].z, pa[2].w}, po[4] = {pa[3].x, pa[3].y, pa[3].z, pa[3].w};
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
        float cn[4], hn[4], ig[4], fg[4], gg[4], og[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ig[e] = fast_sigmoid(pi[e]); fg[e] = fast_sigmoid(pf[e]); gg[e] = subgc_tanh(pg[e]); og[e] = fast_sigmoid(po[e]);
            cn[e] = fg[e] * cpv[e] + ig[e] * gg[e];
            hn[e] = og[e] * subgc_tanh(cn[e]);
        }
        st4(a.c + q, make_float4(cn[0], cn[1], cn[2], cn[3]));
        if (s < a.rows_h) subgc_store_act<4>(a.h, (int64_t)s * a.ldh + j, hn, B16);
        if (a.h2 && s < a.rows_h2) subgc_store_act<4>(a.h2, (int64_t)s * a.ldh2 + j, hn, B16);
        if (a.gates) {
            float* gp = a.gates + (int64_t)s * 4 * R + j;
            st4(gp, make_float4(ig[0], ig[1], ig[2], ig[3])); st4(gp + R, make_float4(fg[0], fg[1], fg[2], fg[3]));
            st4(gp + 2 * R, make_float4(gg[0], gg[1], gg[2], gg[3])); st4(gp + 3 * R, make_float4(og[0], og[1], og[2], og[3]));
        }
        if (B16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(hs) + (size_t)r * a.HP + j) = subgc_pack4(hn[0], hn[1], hn[2], hn[3]);
        else st4(reinterpret_cast<float*>(hs) + (size_t)r * a.HP + j, make_float4(hn[0], hn[1], hn[2], hn[3]));
    }
    __syncthreads();
    stamp(a.stamps, 1);

    // ---- phase 2: q = h1 Wq^T + bq for the nv rows
    const int rot = (blockIdx.x * 5 + wave * 2);
    if (B16) {
        constexpr int NT = 2;                                                 // 16 waves x 2 tiles x 16 columns = 512
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        skinny_mfma_b16<NT, 8>(static_cast<const uint16_t*>(a.Wq), a.ldw, A, R, a.KP, hs, a.HP * 2, nv, wave * NT, rot % (a.KP >> 5), acc);
        const int r16 = lane & 15, kq = lane >> 4;
        if (r16 < nv) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int n = (wave * NT + i) * 16 + 4 * kq + v;
                    if (n < A) qs[(size_t)r16 * A + n] = acc[i][v] + a.bq[n];
                }
        }
        __syncthreads();
    } else {
        skinny_valu_f32<128>(static_cast<const float*>(a.Wq), a.ldw, A, R, reinterpret_cast<const float*>(hs), a.HP, nv, part, A, blockIdx.x * 3);
        __syncthreads();
        for (int e = t; e < nv * A4; e += MID_THREADS) {
            const int r = e / A4, c4 = e - r * A4;
            float4 sum = valu_partial_sum<128>(part, A, r, c4);
            add4(sum, ld4(a.bq + c4 * 4));
            st4(qs + (size_t)r * A + c4 * 4, sum);
        }
        __syncthreads();
    }
    stamp(a.stamps, 2);
    if (a.q_out)                                                              // the summed query rows, kept for the backward
        for (int e = t; e < nv * A4; e += MID_THREADS) {
            const int r = e / A4, c4 = e - r * A4;
            st4(a.q_out + (int64_t)(s0 + r) * A + c4 * 4, *reinterpret_cast<const float4*>(qs + (size_t)r * A + c4 * 4));
        }

    // ---- phase 3: attention of the nv sentences over their node sets (attn_fwd_vec_kernel's arithmetic)
    constexpr int CA = 2, NCH = 4;                                            // A <= 512: two float4 chunks of a score row per lane
    const int ltot = pre[nv];
    float4 w[CA];
#pragma unroll
    for (int c = 0; c < CA; ++c) w[c] = (lane + c * 64 < A4) ? ld4(a.w_a + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float ba = a.b_a[0];
    for (int base = 0; base < ltot; base += MID_WAVES * NCH) {                // (sentence, node) pairs: wave-strided, four per wave in flight
        float4 x[NCH][CA];
        int rr[NCH], ii[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int it = base + wave + MID_WAVES * k;
            int r = 0;
            for (int q = 1; q < nv; ++q) r += it >= pre[q] ? 1 : 0;
            rr[k] = r; ii[k] = it - pre[r];
            const int64_t row = (int64_t)m0s[r] + ii[k];
#pragma unroll
            for (int c = 0; c < CA; ++c)
                x[k][c] = (it < ltot && lane + c * 64 < A4) ? ldx<UV16>(a.u, row * A + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float sc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            float acc = 0.f;
            if (base + wave + MID_WAVES * k < ltot) {
#pragma unroll
                for (int c = 0; c < CA; ++c) {
                    const float4 q = (lane + c * 64 < A4) ? *reinterpret_cast<const float4*>(qs + (size_t)rr[k] * A + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc += w[c].x * subgc_tanh(x[k][c].x + q.x) + w[c].y * subgc_tanh(x[k][c].y + q.y) + w[c].z * subgc_tanh(x[k][c].z + q.z) +
                           w[c].w * subgc_tanh(x[k][c].w + q.w);
                }
            }
            sc[k] = acc;
        }
        wave_sum_n<NCH>(sc);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (base + wave + MID_WAVES * k < ltot) es[(size_t)rr[k] * a.n_stride + ii[k]] = sc[k] + ba;
        }
    }
    __syncthreads();
    stamp(a.stamps, 3);
    for (int r = wave; r < nv; r += MID_WAVES) {                              // softmax of row r: one wave, every lane the same sums (in index order)
        const int l = pre[r + 1] - pre[r];
        float* e = es + (size_t)r * a.n_stride;
        float mx = -INFINITY;
        for (int i = 0; i < l; ++i) mx = fmaxf(mx, e[i]);
        float den = 0.f;
        for (int i = 0; i < l; ++i) den += expf(e[i] - mx);
        float al[(MAXLEN + 63) / 64];
#pragma unroll
        for (int c = 0; c < (MAXLEN + 63) / 64; ++c) al[c] = (lane + c * 64 < l) ? expf(e[lane + c * 64] - mx) / den : 0.f;
#pragma unroll
        for (int c = 0; c < (MAXLEN + 63) / 64; ++c) {
            const int i = lane + c * 64;
            if (i < l) e[i] = al[c];
            if (a.alpha && i < a.n_stride) a.alpha[(int64_t)(s0 + r) * a.n_stride + i] = al[c];
        }
    }
    __syncthreads();
    stamp(a.stamps, 4);
    for (int idx = t; idx < nv * R4; idx += MID_THREADS) {                    // context rows: thread = float4 column chunk, eight node rows in flight
        const int r = idx / R4, c4 = idx - r * R4;
        const int l = pre[r + 1] - pre[r];
        const float* e = es + (size_t)r * a.n_stride;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t vp = (int64_t)m0s[r] * R + c4 * 4;
        int i = 0;
        for (; i + 8 <= l; i += 8) {
            float4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ldx<UV16>(a.v, vp + (int64_t)(i + k) * R);
#pragma unroll
            for (int k = 0; k < 8; ++k) fma4(acc, e[i + k], x[k]);
        }
        for (; i + 4 <= l; i += 4) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = ldx<UV16>(a.v, vp + (int64_t)(i + k) * R);
#pragma unroll
            for (int k = 0; k < 4; ++k) fma4(acc, e[i + k], x[k]);
        }
        for (; i < l; ++i) fma4(acc, e[i], ldx<UV16>(a.v, vp + (int64_t)i * R));
        const float o[4] = {acc.x, acc.y, acc.z, acc.w};
        subgc_store_act<4>(a.ctx, (int64_t)(s0 + r) * a.ldctx + c4 * 4, o, B16);
    }
    __syncthreads();
    stamp(a.stamps, 5);
}

// y[c, r] = x[r, c] (fp32): 64x64 tiles through LDS -- the K-major copy of the query weight the fp32 forms stream
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? x[(int64_t)(r0 + r) * ldx + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < rows) y[(int64_t)(c0 + c) * ldy + r0 + r] = tile[r][c];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace subgc {

// rows per workgroup: one workgroup per CU when the batch allows it (every workgroup streams ALL of Wq whatever its row count, so fewer,
// fatter workgroups would only lengthen the cell and attention phases of the busiest CU)
int mid_rows_per_wg(int m) {
    int rb = (m + 255) / 256;
    return rb < 1 ? 1 : (rb > MID_MAXRB ? MID_MAXRB : rb);
}

// -100: the fused form does not cover the shape (the caller issues the three launches)
int mid_fwd(const float* g0, int64_t ld0, int parts, int64_t plane, const float* g1, int64_t ld1, const float* g2, int64_t ld2, const float* b0,
            const float* b1, const float* c_prev, float* c, void* h, int64_t ldh, int rows_h, void* h2, int64_t ldh2, int rows_h2, float* gates,
            const void* Wq, int64_t ldw, const float* bq, float* q_out, const void* u, const void* v, const float* w_a, const float* b_a,
            const int32_t* off, const int32_t* len, void* ctx, int64_t ldctx, float* alpha, int n_stride, int m, int R, int A, int b16, int uv16,
            int flags, hipStream_t s, long long* stamps) {
    if (m <= 0) return SUBGC_OK;
    if (b16 != uv16) return -100;
    if (R % 8 || A % 4 || A > 512 || n_stride > MAXLEN || n_stride < 0 || (int64_t)m > (int64_t)256 * (b16 ? MID_MAXRB : MID_MAXRB_F32)) return -100;
    if (!b16 && ldw < A) return -100;                                     // fp32: the K-MAJOR weight WqT [R, A]
    if (ld0 % 4 || plane % 4 || ld1 % 4 || ld2 % 4 || ldh % 4 || ldh2 % 4 || ldctx % 4 || ldw % (b16 ? 8 : 4)) return -100;
    if (!aligned16(g0) || !aligned16(g1) || !aligned16(g2) || !aligned16(b0) || !aligned16(b1) || !aligned16(c_prev) || !aligned16(c) || !aligned16(h) ||
        !aligned16(h2) || !aligned16(gates) || !aligned16(Wq) || !aligned16(bq) || !aligned16(q_out) || !aligned16(u) || !aligned16(v) || !aligned16(w_a) ||
        !aligned16(ctx))
        return -100;
    if (rows_h <= 0 || rows_h > m) rows_h = m;
    if (rows_h2 <= 0 || rows_h2 > m) rows_h2 = m;
    const int RB = mid_rows_per_wg(m), KP = (R + 31) / 32 * 32, HP = KP + (b16 ? 16 : 8);
    MidFwd a{g0, ld0, parts < 1 ? 1 : parts, plane, g1, ld1, g2, ld2, b0, b1, c_prev, c, h, ldh, rows_h, h2, ldh2, rows_h2, gates,
             Wq, ldw, bq, q_out, u, v, w_a, b_a, off, len, ctx, ldctx, alpha, n_stride, m, R, A, RB, KP, HP, stamps};
    size_t lds = ((size_t)RB * HP * (b16 ? 2 : 4) + 15) / 16 * 16 + (size_t)RB * A * 4 + (size_t)RB * n_stride * 4 + (2 * MID_MAXRB + 2) * 4 + 16;
    if (!b16) lds += (size_t)(MID_THREADS / 128) * MID_MAXRB_F32 * A * 4;     // the k-groups' partial sums
    if (flags & 1) lds = std::max<size_t>(lds, 84 * 1024);                    // more than half a CU's LDS: one workgroup per CU
    if (lds > 150 * 1024) return -100;
    const int wgs = (m + RB - 1) / RB;
    ProfScope prof(SUBGC_FAM_MID, s, 2.0 * m * (double)A * R);
    if (b16) {
        if (int rc = raise_lds_cached((const void*)mid_fwd_kernel<true, true>, lds, "mid_fwd")) return rc;
        hipLaunchKernelGGL((mid_fwd_kernel<true, true>), dim3(wgs), dim3(MID_THREADS), lds, s, a);
    } else {
        if (int rc = raise_lds_cached((const void*)mid_fwd_kernel<false, false>, lds, "mid_fwd")) return rc;
        hipLaunchKernelGGL((mid_fwd_kernel<false, false>), dim3(wgs), dim3(MID_THREADS), lds, s, a);
    }
    return check_launch("subgc_mid_fwd");
}

}  // namespace subgc

// C ABI: cell update + attention query + attention of one decoder step's rows in one launch (see the file comment).  The arguments are
// those of subgc_lstm_fwd (gate pre-activations as `parts` planes `plane_stride` floats apart: the split-K planes of the gate product, or
// parts = 1), of the query product (Wq [A, R] K-contiguous, bias bq, q_out [m, A] = the summed query rows the backward reads) and of
// subgc_attn_fwd.  bf16_bits: bit 0 = h / h2 / ctx destinations and Wq are bf16, bit 1 = u and v are bf16 (both or neither).
// flags: bit 0 = request enough LDS that a CU holds one workgroup.
SUBGC_API int subgc_mid_fwd(const float* g0, int64_t ld0, int parts, int64_t plane_stride, const float* g1, int64_t ld1, const float* g2, int64_t ld2,
                            const float* b0, const float* b1, const float* c_prev, float* c, void* h, int64_t ldh, int rows_h, void* h2, int64_t ldh2,
                            int rows_h2, float* gates, const void* Wq, int64_t ldWq, const float* bq, float* q_out, const void* u, const void* v,
                            const float* w_a, const float* b_a, const int32_t* off, const int32_t* len, void* ctx, int64_t ldctx, float* alpha,
                            int n_stride, int m, int R, int A, int bf16_bits, int flags, int64_t* debug_stamps, void* stream) {
    SUBGC_REQUIRE(m >= 0 && R > 0 && A > 0, "mid_fwd: bad sizes");
    if (m == 0) return SUBGC_OK;
    SUBGC_REQUIRE(g0 && c && h && Wq && bq && u && v && w_a && b_a && off && len && ctx, "mid_fwd: null pointer");
    const int rc = subgc::mid_fwd(g0, ld0, parts, plane_stride, g1, ld1, g2, ld2, b0, b1, c_prev, c, h, ldh, rows_h, h2, ldh2, rows_h2, gates, Wq, ldWq, bq,
                                  q_out, u, v, w_a, b_a, off, len, ctx, ldctx, alpha, n_stride, m, R, A, bf16_bits & 1, (bf16_bits >> 1) & 1, flags,
                                  (hipStream_t)stream, reinterpret_cast<long long*>(debug_stamps));
    if (rc != -100) return rc;
    subgc::set_error("mid_fwd: shape not covered (needs rnn_size %% 8 == 0, att_hid_size %% 4 == 0 and <= 512, <= 4096 rows, 16-byte aligned rows, "
                     "u / v stored like the operands): R=%d A=%d m=%d", R, A, m);
    return SUBGC_EINVAL;
}

SUBGC_API int subgc_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= rows, "transpose_f32: bad sizes");
    if (rows == 0 || cols == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && y, "transpose_f32: null pointer");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols);
    return subgc::check_launch("subgc_transpose_f32");
}
