// Decoder-side kernels of the Sub-GC hot path: ragged attention layout, word-embedding gather,
// fused LSTM gate math, the per-step attention (scores + softmax + context) as a wavefront-
// primitive kernel, row log-softmax / masked NLL, greedy / top-k token choice, dropout masks and
// the fused clip + Adam step over a flat bucket.  All HBM- or latency-bound; fp32.
//
// Reference op sites: AttModel.py:16-36,348-354 (clip/pack), :332 (embed), :411-413,421-423
// (LSTMCell), :445-471 (Attention), :336,340 (log_softmax), :295-316 (token choice),
// misc/utils.py:115-124 (criterion), :174-200,234-235 (clip-norm, Adam).
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

// ------------------------------------------------------------------ ragged rows (pack)
__global__ __launch_bounds__(1024) void pack_rows_kernel(const int32_t* __restrict__ len, const int64_t* __restrict__ idx,
                                                         int64_t idx_stride, const int32_t* __restrict__ img, int S, int N,
                                                         int32_t* __restrict__ off, int32_t* __restrict__ total,
                                                         int32_t* __restrict__ src_row, int32_t* __restrict__ sent_of) {
    __shared__ int part[1024];
    __shared__ int carry_s;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < S; base += 1024) {
        const int s = base + t;
        const int l = s < S ? max(0, min(len[s], N)) : 0;
        part[t] = l;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan
            const int v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        const int excl = carry_s + part[t] - l;
        if (s < S) off[s] = excl;
        __syncthreads();
        if (t == 1023) carry_s += part[1023];
        __syncthreads();
    }
    const int tot = carry_s;
    if (t == 0) total[0] = tot;
    // rows: one thread per (sentence, slot)
    for (int64_t q = t; q < (int64_t)S * N; q += 1024) {
        const int s = (int)(q / N), i = (int)(q % N);
        const int l = max(0, min(len[s], N));
        if (i < l) {
            int64_t n = idx[(int64_t)s * idx_stride + i];
            n = n < 0 ? 0 : (n >= N ? N - 1 : n);
            const int m = off[s] + i;
            src_row[m] = img[s] * N + (int)n;
            sent_of[m] = s;
        }
        if (q >= tot) { src_row[q] = -1; sent_of[q] = -1; }
    }
}

// ------------------------------------------------------------------ embedding (+ReLU +dropout)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                        int64_t tok_stride, const uint8_t* __restrict__ keep, float scale,
                                                        void* __restrict__ out, int n, int E, int rows, int b16) {
    const int r = blockIdx.x;
    int64_t w = tok[(int64_t)r * tok_stride];
    w = w < 0 ? 0 : (w >= rows ? rows - 1 : w);
    for (int c = threadIdx.x; c < E; c += blockDim.x) {
        float v = fmaxf(table[w * E + c], 0.f);
        if (keep) v = keep[(int64_t)r * E + c] ? v * scale : 0.f;
        const float o[1] = {v};
        subgc_store_act<1>(out, (int64_t)r * E + c, o, b16);
    }
}
// plain row lookup by token: out[r, :] = table[tok[r], :]  (decode-time x->gates table, see subgc_token_rows_f32)
__global__ __launch_bounds__(256) void token_rows_kernel(const float* __restrict__ table, int64_t ldt, const int64_t* __restrict__ tok,
                                                         int64_t tok_stride, float* __restrict__ out, int64_t ldo, int C, int rows,
                                                         int vec) {
    const int r = blockIdx.x;
    int64_t w = tok[(int64_t)r * tok_stride];
    w = w < 0 ? 0 : (w >= rows ? rows - 1 : w);
    const float* src = table + w * ldt;
    float* dst = out + (int64_t)r * ldo;
    if (vec) {
        for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4)
            *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = src[c];
    }
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                        int64_t tok_stride, const uint8_t* __restrict__ keep, float scale,
                                                        const float* __restrict__ dout, float* __restrict__ dtable, int n, int E,
                                                        int rows) {
    const int r = blockIdx.x;
    int64_t w = tok[(int64_t)r * tok_stride];
    w = w < 0 ? 0 : (w >= rows ? rows - 1 : w);
    for (int c = threadIdx.x; c < E; c += blockDim.x) {
        if (table[w * E + c] <= 0.f) continue;
        float g = dout[(int64_t)r * E + c];
        if (keep) g = keep[(int64_t)r * E + c] ? g * scale : 0.f;
        if (g != 0.f) unsafeAtomicAdd(dtable + w * E + c, g);
    }
}

// ------------------------------------------------------------------ LSTM gate math
// VW = 4: a thread owns four consecutive hidden units (float4 loads / stores; every pointer 16-byte aligned, R and the
// leading dimensions multiples of 4); VW = 1 is the unaligned fallback.  These kernels move 12-14 floats per hidden unit
// and do ~10 transcendental ops: pure HBM streaming, so bytes in flight per thread is what sets their speed.
template <int VW>
__device__ __forceinline__ void ldv(const float* p, float (&o)[VW]) {
    if (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1 % VW] = t.y; o[2 % VW] = t.z; o[3 % VW] = t.w; }
    else o[0] = *p;
}
template <int VW>
__device__ __forceinline__ void stv(float* p, const float (&o)[VW]) {
    if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % VW], o[2 % VW], o[3 % VW]);
    else *p = o[0];
}
template <int VW>
__device__ __forceinline__ void ldk(const uint8_t* p, bool (&o)[VW]) {
    if (VW == 4) { const uint32_t t = *reinterpret_cast<const uint32_t*>(p); o[0] = t & 0xffu; o[1 % VW] = t & 0xff00u; o[2 % VW] = t & 0xff0000u; o[3 % VW] = t & 0xff000000u; }
    else o[0] = *p != 0;
}

// PARTS > 0: the number of pre-activation planes is a compile-time constant, so every plane's load is issued before any is added (a
// runtime trip count made the sum a chain of dependent round trips: three planes = three memory latencies per thread); the additions
// keep their order, the result is bit for bit that of the generic form (PARTS = 0)
template <int VW, int PARTS = 0>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ g0, int64_t ld0, const float* __restrict__ g1,
                                                       int64_t ld1, const float* __restrict__ g2, int64_t ld2,
                                                       const float* __restrict__ b0, const float* __restrict__ b1,
                                                       const float* __restrict__ c_prev, float* __restrict__ c,
                                                       void* __restrict__ h, int64_t ldh, void* __restrict__ h2, int64_t ldh2,
                                                       const uint8_t* __restrict__ keep, float scale, void* __restrict__ hdrop,
                                                       int64_t ldhd, float* __restrict__ gates, int S, int R, int rows_h, int rows_h2,
                                                       int parts, int64_t plane, int hb16) {
    // hb16: the three h destinations are bf16 (they are GEMM operands only: the next step's / the logit product's A rows)
    // parts > 1: g0 is a stack of split-K partial planes g0[p * plane + ...] (subgc_lstm_fwd_gemm): summed here, no reduce pass
    const int RV = R / VW;
    const int64_t qv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qv >= (int64_t)S * RV) return;
    const int s = (int)(qv / RV), j = (int)(qv % RV) * VW;
    const int64_t q = (int64_t)s * R + j;
    float pre[4][VW];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = k * R + j;
        float t[VW];
        ldv<VW>(g0 + (int64_t)s * ld0 + col, pre[k]);
        if (PARTS > 0) {
            float tp[PARTS > 1 ? PARTS - 1 : 1][VW];
#pragma unroll
            for (int pt = 1; pt < PARTS; ++pt) ldv<VW>(g0 + pt * plane + (int64_t)s * ld0 + col, tp[pt - 1]);
#pragma unroll
            for (int pt = 1; pt < PARTS; ++pt)
#pragma unroll
                for (int e = 0; e < VW; ++e) pre[k][e] += tp[pt - 1][e];
        } else {
            for (int pt = 1; pt < parts; ++pt) {
                ldv<VW>(g0 + pt * plane + (int64_t)s * ld0 + col, t);
#pragma unroll
                for (int e = 0; e < VW; ++e) pre[k][e] += t[e];
            }
        }
        if (g1) { ldv<VW>(g1 + (int64_t)s * ld1 + col, t);
#pragma unroll
            for (int e = 0; e < VW; ++e) pre[k][e] += t[e]; }
        if (g2) { ldv<VW>(g2 + (int64_t)s * ld2 + col, t);
#pragma unroll
            for (int e = 0; e < VW; ++e) pre[k][e] += t[e]; }
        if (b0) { ldv<VW>(b0 + col, t);
#pragma unroll
            for (int e = 0; e < VW; ++e) pre[k][e] += t[e]; }
        if (b1) { ldv<VW>(b1 + col, t);
#pragma unroll
            for (int e = 0; e < VW; ++e) pre[k][e] += t[e]; }
    }
    float cp[VW], cn[VW], hn[VW], ig[VW], fg[VW], gg[VW], og[VW];
    if (c_prev) ldv<VW>(c_prev + q, cp);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        ig[e] = sigmoidf_(pre[0][e]); fg[e] = sigmoidf_(pre[1][e]); gg[e] = tanhf(pre[2][e]); og[e] = sigmoidf_(pre[3][e]);
        cn[e] = fg[e] * (c_prev ? cp[e] : 0.f) + ig[e] * gg[e];
        hn[e] = og[e] * tanhf(cn[e]);
    }
    stv<VW>(c + q, cn);
    if (s < rows_h) subgc_store_act<VW>(h, (int64_t)s * ldh + j, hn, hb16);
    if (h2 && s < rows_h2) subgc_store_act<VW>(h2, (int64_t)s * ldh2 + j, hn, hb16);
    if (hdrop) {
        float hd[VW];
        bool kp[VW];
        if (keep) ldk<VW>(keep + q, kp);
#pragma unroll
        for (int e = 0; e < VW; ++e) hd[e] = keep ? (kp[e] ? hn[e] * scale : 0.f) : hn[e];
        subgc_store_act<VW>(hdrop, (int64_t)s * ldhd + j, hd, hb16);
    }
    if (gates) {
        float* gp = gates + (int64_t)s * 4 * R + j;
        stv<VW>(gp, ig); stv<VW>(gp + R, fg); stv<VW>(gp + 2 * R, gg); stv<VW>(gp + 3 * R, og);
    }
}
// one additive source of d(h): sum over its `n` split-K partial planes p[q * stride + s * ld + j] for rows s < rows (rows past it: 0).
// n = 1 is a plain [S, R] view; n > 1 lets this kernel consume the partial planes of the data-gradient GEMM directly (no reduce pass)
struct PlaneSrc { const float* p; int64_t ld; int n; int64_t stride; int rows; };
template <int VW>
__device__ __forceinline__ void add_src(const PlaneSrc& a, int s, int j, float (&dh)[VW]) {
    if (!a.p || s >= a.rows) return;
    float t[VW];
    for (int q = 0; q < a.n; ++q) {
        ldv<VW>(a.p + q * a.stride + (int64_t)s * a.ld + j, t);
#pragma unroll
        for (int e = 0; e < VW; ++e) dh[e] += t[e];
    }
}
template <int VW>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                       const float* __restrict__ c, PlaneSrc sa, PlaneSrc sb, PlaneSrc sc,
                                                       const float* __restrict__ dh_d, int64_t ldd, const uint8_t* __restrict__ keep, float scale,
                                                       const float* __restrict__ dc, void* __restrict__ dpre,
                                                       float* __restrict__ dc_prev, int S, int R, int pb16) {
    const int RV = R / VW;
    const int64_t qv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qv >= (int64_t)S * RV) return;
    const int s = (int)(qv / RV), j = (int)(qv % RV) * VW;
    const int64_t q = (int64_t)s * R + j;
    const float* gp = gates + (int64_t)s * 4 * R + j;
    float ig[VW], fg[VW], gg[VW], og[VW], dh[VW], t[VW], cc[VW], cp[VW], dcv[VW];
    ldv<VW>(gp, ig); ldv<VW>(gp + R, fg); ldv<VW>(gp + 2 * R, gg); ldv<VW>(gp + 3 * R, og);
#pragma unroll
    for (int e = 0; e < VW; ++e) dh[e] = 0.f;
    add_src<VW>(sa, s, j, dh);
    add_src<VW>(sb, s, j, dh);
    add_src<VW>(sc, s, j, dh);
    if (dh_d) {
        bool kp[VW];
        ldv<VW>(dh_d + (int64_t)s * ldd + j, t);
        if (keep) ldk<VW>(keep + q, kp);
#pragma unroll
        for (int e = 0; e < VW; ++e) dh[e] += keep ? (kp[e] ? t[e] * scale : 0.f) : t[e];
    }
    ldv<VW>(c + q, cc);
    if (dc) ldv<VW>(dc + q, dcv);
    if (c_prev) ldv<VW>(c_prev + q, cp);
    float d0[VW], d1[VW], d2[VW], d3[VW], dcp[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        const float tc = tanhf(cc[e]);
        float dct = dh[e] * og[e] * (1.f - tc * tc);
        if (dc) dct += dcv[e];
        const float cpe = c_prev ? cp[e] : 0.f;
        d0[e] = dct * gg[e] * ig[e] * (1.f - ig[e]);
        d1[e] = dct * cpe * fg[e] * (1.f - fg[e]);
        d2[e] = dct * ig[e] * (1.f - gg[e] * gg[e]);
        d3[e] = dh[e] * tc * og[e] * (1.f - og[e]);
        dcp[e] = dct * fg[e];
    }
    const int64_t dp = (int64_t)s * 4 * R + j;                  // gate gradients: GEMM operands only -> may be stored bf16
    subgc_store_act<VW>(dpre, dp, d0, pb16); subgc_store_act<VW>(dpre, dp + R, d1, pb16);
    subgc_store_act<VW>(dpre, dp + 2 * R, d2, pb16); subgc_store_act<VW>(dpre, dp + 3 * R, d3, pb16);
    stv<VW>(dc_prev + q, dcp);
}

// ------------------------------------------------------------------ attention step
// the per-step attention kernels live in attention_vec.hip (one workgroup per sentence) and attention_group.hip (shared sets)
constexpr int MAXLEN = 512;

// ------------------------------------------------------------------ log-softmax rows / NLL
// the row is held in registers between its single read and its single write (PER x 256 >= V)
template <int PER>
__global__ __launch_bounds__(256) void log_softmax_kernel(float* __restrict__ x, int64_t ldx, int rows, int V,
                                                          const int32_t* __restrict__ active, float* __restrict__ lse_out) {
    __shared__ float sm[16];
    const int r = blockIdx.x;
    float* p = x + (int64_t)r * ldx;
    if (!lse_out && active && !active[r]) {
        for (int c = threadIdx.x; c < V; c += blockDim.x) p[c] = 0.f;
        return;
    }
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = threadIdx.x + j * 256;
        v[j] = c < V ? p[c] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) mx = fmaxf(mx, v[j]);
    mx = block_max(mx, sm);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) sum += (threadIdx.x + j * 256 < V) ? expf(v[j] - mx) : 0.f;
    sum = block_sum(sum, sm);
    const float lse = mx + logf(sum);
    if (lse_out) {                                                        // the row stays as it is: consumers subtract lse themselves
        if (threadIdx.x == 0) lse_out[r] = lse;
        return;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < V) p[c] = v[j] - lse;
    }
}
__device__ __forceinline__ void store_one(void* base, int64_t i, float v, int b16) {
    if (b16) static_cast<uint16_t*>(base)[i] = (uint16_t)subgc_f2bf(v);
    else static_cast<float*>(base)[i] = v;
}
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ logp, const float* dout, void* dlogits,
                                                              int64_t ld, int rows, int V, const int32_t* __restrict__ active, int b16) {
    __shared__ float sm[16];
    const int r = blockIdx.x;
    const float* lp = logp + (int64_t)r * ld;
    const float* d = dout + (int64_t)r * ld;
    const int64_t o = (int64_t)r * ld;
    if (active && !active[r]) {
        for (int c = threadIdx.x; c < V; c += blockDim.x) store_one(dlogits, o + c, 0.f, b16);
        return;
    }
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += blockDim.x) sum += d[c];
    sum = block_sum(sum, sm);
    for (int c = threadIdx.x; c < V; c += blockDim.x) store_one(dlogits, o + c, d[c] - expf(lp[c]) * sum, b16);
}
__global__ __launch_bounds__(1024) void nll_fwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target,
                                                       int64_t t_stride, const float* __restrict__ mask, int64_t m_stride,
                                                       float* __restrict__ loss, float* __restrict__ scratch2, int S, int T, int V,
                                                       const float* __restrict__ den_override, const float* __restrict__ lse) {
    __shared__ float sm[16];
    float num = 0.f, den = 0.f;
    for (int q = threadIdx.x; q < S * T; q += blockDim.x) {
        const int s = q / T, t = q % T;
        const float m = mask[(int64_t)s * m_stride + t];
        int64_t w = target[(int64_t)s * t_stride + t];
        w = w < 0 ? 0 : (w >= V ? V - 1 : w);
        num += -(logp[(int64_t)q * V + w] - (lse ? lse[q] : 0.f)) * m;
        den += m;
    }
    num = block_sum(num, sm);
    den = block_sum(den, sm);
    if (den_override) den = den_override[0];           // the criterion's denominator over rows this call does not see (packed decoder)
    if (threadIdx.x == 0) { scratch2[0] = num; scratch2[1] = den; loss[0] = num / den; }
}
__global__ __launch_bounds__(256) void nll_bwd_kernel(const int64_t* __restrict__ target, int64_t t_stride,
                                                      const float* __restrict__ mask, int64_t m_stride,
                                                      const float* __restrict__ scratch2, const float* __restrict__ dloss,
                                                      float* __restrict__ dlogp, int S, int T, int V) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= S * T) return;
    const int s = q / T, t = q % T;
    int64_t w = target[(int64_t)s * t_stride + t];
    w = w < 0 ? 0 : (w >= V ? V - 1 : w);
    dlogp[(int64_t)q * V + w] = -mask[(int64_t)s * m_stride + t] / scratch2[1] * dloss[0];
}
// fused backward of loss = -sum(mask * logp[target]) / sum(mask) through the log-softmax:
// dlogits[r, c] = dloss * mask_r / den * (exp(logp[r, c]) - [c == target_r]); rows that are masked or inactive get zeros.
__global__ __launch_bounds__(256) void nll_logsoftmax_bwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target,
                                                                 int64_t t_stride, const float* __restrict__ mask, int64_t m_stride,
                                                                 const float* __restrict__ scratch2, const float* __restrict__ dloss,
                                                                 void* __restrict__ dlogits, int64_t ldo, int T, int V,
                                                                 const int32_t* __restrict__ active, int b16, const float* __restrict__ lse, int vec) {
    // b16: d(logits) is a GEMM operand only (weight gradient and d(hidden)): written bf16, half the bytes of the largest tensor
    // lse != NULL: `logp` holds RAW logits and lse[r] their row log-sum-exp (subgc_row_lse_f32): exp(x - lse) is the same number
    // vec: V % 4 == 0 and 16-byte aligned rows on both sides -> four columns per lane (16-byte loads, 8/16-byte stores)
    const int r = blockIdx.x, s = r / T, t = r % T;
    const float m = mask[(int64_t)s * m_stride + t];
    const int64_t d = (int64_t)r * ldo;
    if (m == 0.f || (active && !active[r])) {
        if (vec) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; for (int c = threadIdx.x * 4; c < V; c += blockDim.x * 4) subgc_store_act<4>(dlogits, d + c, z, b16); }
        else for (int c = threadIdx.x; c < V; c += blockDim.x) store_one(dlogits, d + c, 0.f, b16);
        return;
    }
    int64_t w = target[(int64_t)s * t_stride + t];
    w = w < 0 ? 0 : (w >= V ? V - 1 : w);
    const float g = dloss[0] * m / scratch2[1];
    const float* lp = logp + (int64_t)r * V;
    const float sub = lse ? lse[r] : 0.f;
    if (vec) {
        for (int c = threadIdx.x * 4; c < V; c += blockDim.x * 4) {
            const float4 x = *reinterpret_cast<const float4*>(lp + c);
            const int wi = (int)w - c;
            const float o[4] = {g * (expf(x.x - sub) - (wi == 0 ? 1.f : 0.f)), g * (expf(x.y - sub) - (wi == 1 ? 1.f : 0.f)),
                                g * (expf(x.z - sub) - (wi == 2 ? 1.f : 0.f)), g * (expf(x.w - sub) - (wi == 3 ? 1.f : 0.f))};
            subgc_store_act<4>(dlogits, d + c, o, b16);
        }
        return;
    }
    for (int c = threadIdx.x; c < V; c += blockDim.x) store_one(dlogits, d + c, g * (expf(lp[c] - sub) - (c == (int)w ? 1.f : 0.f)), b16);
}
__global__ __launch_bounds__(256) void step_active_kernel(const int64_t* __restrict__ labels, int64_t l_stride, int S, int T,
                                                          int32_t* __restrict__ active) {
    __shared__ int any_s[512];
    __shared__ int sm_i[16];
    for (int t = 0; t < T; ++t) {
        int any = 0;
        for (int s = threadIdx.x; s < S; s += blockDim.x) any |= labels[(int64_t)s * l_stride + t] != 0;
        any = __any(any) ? 1 : 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm_i[threadIdx.x >> 6] = any;
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a |= sm_i[w];
            // the reference tests seq[:, i].sum() == 0 (labels are >= 0): "all zero" == "none non-zero"
            any_s[t] = (t == 0) ? 1 : (a && any_s[t - 1]);
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < S * T; q += blockDim.x) active[q] = any_s[q % T];
}

// ------------------------------------------------------------------ token choice (decode)
__device__ __forceinline__ void block_argmax(float& v, int& i, float* sv, int* si) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[w] = v; si[w] = i; }
    __syncthreads();
    v = sv[0]; i = si[0];
    for (int k = 1; k < nw; ++k)
        if (sv[k] > v || (sv[k] == v && si[k] < i)) { v = sv[k]; i = si[k]; }
}
constexpr int MAXK = 8;
// One workgroup per row.  The row (<= 256*PER logits) is loaded into registers ONCE, with every load in flight together:
// the first version walked the row two to k+3 times with dependent strided loads and cost 20 us for ten 38 KB rows --
// the largest single kernel of the one-image decode step.
template <int PER>
__global__ __launch_bounds__(256) void decode_pick_kernel(const float* __restrict__ logp, int64_t ld, int n, int V, int k, float temp,
                                                          const float* __restrict__ u, int t, int64_t* __restrict__ seq,
                                                          float* __restrict__ seqlp, int T, int64_t* __restrict__ next_tok,
                                                          int32_t* __restrict__ unfinished, int32_t* __restrict__ n_unfinished,
                                                          const int32_t* __restrict__ prev_count, int raw) {
    if (prev_count && *prev_count == 0) return;   // the reference has left its loop (AttModel.py:318-319)
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ float smf[16];
    const int r = blockIdx.x;
    const float* p = logp + (int64_t)r * ld;
    float x[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = threadIdx.x + j * 256;
        x[j] = c < V ? p[c] : -INFINITY;
    }
    int it; float lp;
    if (k <= 0) {
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c = threadIdx.x + j * 256;
            if (c < V && (x[j] > bv || bi == 0x7fffffff)) { bv = x[j]; bi = c; }
        }
        block_argmax(bv, bi, sv, si);
        it = bi; lp = bv;
        if (raw) {                                // raw logits: log_softmax(x)[argmax] = -log sum exp(x - max)
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < PER; ++j) sum += (threadIdx.x + j * 256 < V) ? expf(x[j] - bv) : 0.f;
            sum = block_sum(sum, smf);
            lp = -logf(sum);
        }
    } else {
        // lp' = log_softmax(logp / temp)
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < PER; ++j) { x[j] = x[j] / temp; mx = fmaxf(mx, x[j]); }
        mx = block_max(mx, smf);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) sum += (threadIdx.x + j * 256 < V) ? expf(x[j] - mx) : 0.f;
        sum = block_sum(sum, smf);
        const float lse = mx + logf(sum);
        int top_i[MAXK]; float top_v[MAXK];
        if (k <= 3) {
            // k <= 3 (the_k = 3, opts.py): ONE pass keeps every thread's own three best in registers (the global top-3 is a subset of
            // their union), then three block arg-max rounds over one candidate per thread -- the k full passes over the 38 registers
            // with their "after the previous pick" tests were 2.4x the greedy kernel (35 vs 15 us at 900 rows)
            float l0 = -INFINITY, l1 = -INFINITY, l2 = -INFINITY;
            int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int c = threadIdx.x + j * 256;        // ascending per thread: an equal value never displaces an earlier index
                if (c < V) {
                    const float v = x[j] - lse;
                    if (v > l0) { l2 = l1; i2 = i1; l1 = l0; i1 = i0; l0 = v; i0 = c; }
                    else if (v > l1) { l2 = l1; i2 = i1; l1 = v; i1 = c; }
                    else if (v > l2) { l2 = v; i2 = c; }
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q < k) {                                  // k is uniform over the workgroup
                    float bv = l0; int bi = i0;
                    block_argmax(bv, bi, sv, si);
                    top_i[q] = bi; top_v[q] = bv;
                    if (bi == i0) { l0 = l1; i0 = i1; l1 = l2; i1 = i2; l2 = -INFINITY; i2 = 0x7fffffff; }    // the owner moves on to its next best
                }
            }
        } else {
        float pv = INFINITY; int pi = -1;         // (value desc, index asc) is a total order: "after the previous pick" is one test
        for (int q = 0; q < k; ++q) {
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int c = threadIdx.x + j * 256;
                const float v = x[j] - lse;
                const bool after = v < pv || (v == pv && c > pi);
                if (c < V && after && (v > bv || bi == 0x7fffffff)) { bv = v; bi = c; }
            }
            block_argmax(bv, bi, sv, si);
            top_i[q] = bi; top_v[q] = bv;
            pv = bv; pi = bi;
        }
        }
        // Categorical(logits=top): renormalise over the k, inverse CDF in top-k order
        float m2 = top_v[0];
        for (int j = 1; j < k; ++j) m2 = fmaxf(m2, top_v[j]);
        float z = 0.f;
        for (int j = 0; j < k; ++j) z += expf(top_v[j] - m2);
        const float lz = m2 + logf(z);
        const float uu = u ? u[r] : 0.f;
        float cdf = 0.f; int pick = 0;
        for (int j = 0; j < k; ++j) {
            cdf += expf(top_v[j] - lz);
            pick += uu >= cdf;
        }
        pick = min(pick, k - 1);
        it = top_i[pick]; lp = top_v[pick];
    }
    if (threadIdx.x == 0) {
        int unf = (t == 0) ? (it > 0) : (unfinished[r] && it > 0);
        unfinished[r] = unf;
        const int64_t w = unf ? it : 0;
        seq[(int64_t)r * T + t] = w;
        seqlp[(int64_t)r * T + t] = lp;
        next_tok[r] = w;
        // a FLAG, not a count: every consumer only tests it against zero, and one same-address device atomic per row is what this
        // kernel used to spend its time on (2560 rows x ~15 ns of memory-side serialisation = 38 of its 43 us)
        if (unf && n_unfinished) *n_unfinished = 1;
    }
}

// ------------------------------------------------------------------ dropout masks (Philox-4x32-10)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed,
                                                           uint64_t offset) {
    // one Philox counter = 4 x u32 -> 4 uniforms (top 24 bits of each word) -> 4 mask bytes
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset / 4 + (uint64_t)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < n) keep[i] = ((c[j] >> 8) * (1.0f / 16777216.0f)) >= p ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------ utilities
__global__ void fill_kernel(float* x, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void copy2d_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int rows, int cols,
                              int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        const float v = x[r * ldx + c];
        float* d = y + r * ldy + c;
        *d = accumulate ? *d + v : v;
    }
}
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ rows,
                                                               float* __restrict__ dX, int64_t ldx, int M, int L,
                                                               const int32_t* __restrict__ m_dev) {
    if (m_dev) M = min(M, *m_dev);
    const int m = blockIdx.x;
    if (m >= M) return;
    const int r = rows[m];
    if (r < 0) return;
    for (int c = threadIdx.x; c < L; c += blockDim.x) unsafeAtomicAdd(dX + (int64_t)r * ldx + c, src[(int64_t)m * lds_ + c]);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const void* __restrict__ y, float scale, void* __restrict__ dz, int64_t n, int b16,
                                int y16) {
    // y16: the activation survives only as the bf16 tensor the GEMM wrote (same exponent range: y > 0 is unchanged by the rounding)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool pos = y16 ? (static_cast<const uint16_t*>(y)[i] & 0x7fffu) != 0 && !(static_cast<const uint16_t*>(y)[i] & 0x8000u)
                             : static_cast<const float*>(y)[i] > 0.f;
        store_one(dz, i, pos ? dy[i] * scale : 0.f, b16);
    }
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ rows,
                                                          void* __restrict__ dst, int64_t ldd, int M, int L,
                                                          const int32_t* __restrict__ m_dev, int b16) {
    if (m_dev) M = min(M, *m_dev);
    const int m = blockIdx.x;
    if (m >= M) return;
    const int r = rows[m];
    for (int c = threadIdx.x; c < L; c += blockDim.x) store_one(dst, (int64_t)m * ldd + c, r >= 0 ? src[(int64_t)r * lds_ + c] : 0.f, b16);
}
// dst[m, :] = src[rows[m], :] * keep[m, :] * scale: nn.Dropout on gathered copies of shared rows -- the replicated node rows of the
// Full-GC attention sets (gcn_backbone.py:50-51 x5 copies, each with its OWN keep-mask, AttModel.py:113-119) from ONE att_embed
// product over the unique rows.  src / dst fp32 or bf16 (bit 0 of b16: dst, bit 1: src); four columns per lane.
__global__ __launch_bounds__(256) void gather_rows_keep_kernel(const void* __restrict__ src, int64_t lds_, const int32_t* __restrict__ rows,
                                                               const uint8_t* __restrict__ keep, int64_t ldk, float scale,
                                                               void* __restrict__ dst, int64_t ldd, int M, int L,
                                                               const int32_t* __restrict__ m_dev, int b16) {
    if (m_dev) M = min(M, *m_dev);
    const int m = blockIdx.x;
    if (m >= M) return;
    const int r = rows[m];
    const int src16 = (b16 >> 1) & 1, dst16 = b16 & 1;
    for (int c = threadIdx.x; c < L; c += blockDim.x) {
        float x = 0.f;
        if (r >= 0) x = src16 ? subgc_bf2f(static_cast<const uint16_t*>(src)[(int64_t)r * lds_ + c]) : static_cast<const float*>(src)[(int64_t)r * lds_ + c];
        if (keep) x = keep[(int64_t)m * ldk + c] ? x * scale : 0.f;
        store_one(dst, (int64_t)m * ldd + c, x, dst16);
    }
}
// the same row gather for up to four tensors in ONE launch (blockIdx.y picks the tensor): the beam-search state fork
struct GatherSet { const float* src[4]; float* dst[4]; int64_t lds[4], ldd[4]; int cols[4]; };
template <typename IndexT>
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherSet g, const IndexT* __restrict__ rows, int M) {
    const int m = blockIdx.x, k = blockIdx.y;
    if (m >= M) return;
    const int64_t r = (int64_t)rows[m];
    const float* __restrict__ src = g.src[k] + (int64_t)(r >= 0 ? r : 0) * g.lds[k];
    float* __restrict__ dst = g.dst[k] + (int64_t)m * g.ldd[k];
    for (int c = threadIdx.x; c < g.cols[k]; c += blockDim.x) dst[c] = r >= 0 ? src[c] : 0.f;
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += g[i] * g[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, acc);
}
template <bool ZERO>       // ZERO: the gradient is left ZEROED (optimizer.zero_grad() fused into the sweep) instead of scaled and clipped
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, const float* __restrict__ sumsq,
                                                        float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                        float bc2, float gscale, uint16_t* __restrict__ p16) {
    // misc/utils.py:193: coef = clip / max(total_norm, clip); gscale (1/world after a SUM all-reduce) is applied first
    const float coef = gscale * (max_norm > 0.f ? max_norm / fmaxf(sqrtf(sumsq[0]) * gscale, max_norm) : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        g[i] = ZERO ? 0.f : gi;
        if (wd != 0.f) gi += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        const float pn = p[i] - (lr / bc1) * (mi / denom);
        p[i] = pn;
        if (p16) p16[i] = (uint16_t)subgc_f2bf(pn);              // the bf16 weight snapshot the bf16 GEMMs read, refreshed in the same sweep
    }
}

// float4 forms (n % 4 == 0, 16-byte aligned buffers: the flat parameter bucket always is): 1 KB per wave instruction
__global__ __launch_bounds__(256) void sumsq_vec_kernel(const float4* __restrict__ g, int64_t n4, float* __restrict__ out) {
    __shared__ float sm[16];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {              // four loads in flight per thread: a quarter of the workgroups keeps the bytes in flight
        const float4 x0 = g[i], x1 = g[i + stride], x2 = g[i + 2 * stride], x3 = g[i + 3 * stride];
        acc += x0.x * x0.x + x0.y * x0.y + x0.z * x0.z + x0.w * x0.w;
        acc += x1.x * x1.x + x1.y * x1.y + x1.z * x1.z + x1.w * x1.w;
        acc += x2.x * x2.x + x2.y * x2.y + x2.z * x2.z + x2.w * x2.w;
        acc += x3.x * x3.x + x3.y * x3.y + x3.z * x3.z + x3.w * x3.w;
    }
    for (; i < n4; i += stride) {
        const float4 x = g[i];
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, acc);
}
template <bool ZERO>
__global__ __launch_bounds__(256) void clip_adam_vec_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                            float4* __restrict__ v, int64_t n4, const float* __restrict__ sumsq,
                                                            float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                            float bc2, float gscale, uint16_t* __restrict__ p16) {
    const float coef = gscale * (max_norm > 0.f ? max_norm / fmaxf(sqrtf(sumsq[0]) * gscale, max_norm) : 1.f);
    const float rs2 = sqrtf(bc2), step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                          // same arithmetic, element by element, as clip_adam_kernel
            float gi = gg[e] * coef;
            gg[e] = gi;
            if (wd != 0.f) gi += wd * pp[e];
            const float mi = b1 * mm[e] + (1.f - b1) * gi;
            const float vi = b2 * vv[e] + (1.f - b2) * gi * gi;
            mm[e] = mi; vv[e] = vi;
            const float denom = sqrtf(vi) / rs2 + eps;
            pp[e] = pp[e] - step * (mi / denom);
        }
        g[i] = ZERO ? make_float4(0.f, 0.f, 0.f, 0.f) : G; m[i] = M; v[i] = V; p[i] = P;
        if (p16) *reinterpret_cast<uint2*>(p16 + 4 * i) = subgc_pack4(P.x, P.y, P.z, P.w);
    }
}

inline int ew_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }

}  // namespace

SUBGC_API int subgc_pack_rows(const int32_t* len, const int64_t* idx, int64_t idx_stride, const int32_t* img, int S, int N,
                              int32_t* off, int32_t* total, int32_t* src_row, int32_t* sent_of, void* stream) {
    SUBGC_REQUIRE(S >= 0 && N > 0, "pack_rows: bad sizes");
    SUBGC_REQUIRE(off && total && src_row && sent_of && (S == 0 || (len && idx && img)), "pack_rows: null pointer");
    SUBGC_DEBUG_RANGE(idx, 8, S, N, idx_stride, 0, N - 1, -1, "pack_rows: idx (node lists of the selected sub-graphs)", stream);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, len, idx, idx_stride, img, S, N, off, total,
                       src_row, sent_of);
    return subgc::check_launch("subgc_pack_rows");
}

SUBGC_API int subgc_embed_fwd(const float* table, const int64_t* tok, int64_t tok_stride, const uint8_t* keep, float keep_scale,
                              void* out, int n, int E, int vocab_rows, int out_bf16, void* stream) {
    SUBGC_REQUIRE(n >= 0 && E > 0 && vocab_rows > 0, "embed_fwd: bad sizes");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(table && tok && out, "embed_fwd: null pointer");
    SUBGC_DEBUG_RANGE(tok, 8, n, 1, tok_stride, 0, vocab_rows - 1, -1, "embed_fwd: tok (word ids)", stream);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table, tok, tok_stride, keep, keep_scale, out, n,
                       E, vocab_rows, out_bf16);
    return subgc::check_launch("subgc_embed_fwd");
}
SUBGC_API int subgc_token_rows_f32(const float* table, int64_t ldt, const int64_t* tok, int64_t tok_stride, float* out, int64_t ldo,
                                   int n, int C, int vocab_rows, void* stream) {
    SUBGC_REQUIRE(n >= 0 && C > 0 && vocab_rows > 0 && ldt >= C && ldo >= C, "token_rows: bad sizes");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(table && tok && out, "token_rows: null pointer");
    SUBGC_DEBUG_RANGE(tok, 8, n, 1, tok_stride, 0, vocab_rows - 1, -1, "token_rows: tok (word ids)", stream);
    const int vec = (C % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
    hipLaunchKernelGGL(token_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table, ldt, tok, tok_stride, out, ldo, C,
                       vocab_rows, vec);
    return subgc::check_launch("subgc_token_rows_f32");
}
SUBGC_API int subgc_embed_bwd(const float* table, const int64_t* tok, int64_t tok_stride, const uint8_t* keep, float keep_scale,
                              const float* dout, float* dtable, int n, int E, int vocab_rows, void* stream) {
    SUBGC_REQUIRE(n >= 0 && E > 0 && vocab_rows > 0, "embed_bwd: bad sizes");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(table && tok && dout && dtable, "embed_bwd: null pointer");
    SUBGC_DEBUG_RANGE(tok, 8, n, 1, tok_stride, 0, vocab_rows - 1, -1, "embed_bwd: tok (word ids)", stream);
    // (a four-columns-per-lane form was measured: 114 -> 215 us on Full_GC_Kar -- a lane's four atomics land on one line back to back)
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table, tok, tok_stride, keep, keep_scale, dout,
                       dtable, n, E, vocab_rows);
    return subgc::check_launch("subgc_embed_bwd");
}

SUBGC_API int subgc_lstm_fwd(const float* g0, int64_t ld0, const float* g1, int64_t ld1, const float* g2, int64_t ld2, const float* b0,
                             const float* b1, const float* c_prev, float* c, void* h, int64_t ldh, void* h2, int64_t ldh2,
                             const uint8_t* keep, float keep_scale, void* hdrop, int64_t ldhd, float* gates, int S, int R,
                             int rows_h, int rows_h2, int h_bf16, void* stream) {
    SUBGC_REQUIRE(S >= 0 && R > 0, "lstm_fwd: bad sizes");
    if (rows_h <= 0 || rows_h > S) rows_h = S;
    if (rows_h2 <= 0 || rows_h2 > S) rows_h2 = S;
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(g0 && c && h, "lstm_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)S * R;
    subgc::ProfScope prof(SUBGC_FAM_LSTM, s, 4.0 * n * 12, subgc::lstm_fwd_moved_bytes(n, 1, g1 != nullptr, g2 != nullptr, c_prev != nullptr, h2 != nullptr, hdrop != nullptr, gates != nullptr, h_bf16));
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = R % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && ldh % 4 == 0 && ldh2 % 4 == 0 && ldhd % 4 == 0 && al(g0) &&
                     al(g1) && al(g2) && al(b0) && al(b1) && al(c_prev) && al(c) && al(h) && al(h2) && al(hdrop) && al(gates) &&
                     (reinterpret_cast<uintptr_t>(keep) & 3) == 0;
    if (vec)
        hipLaunchKernelGGL(lstm_fwd_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, g0, ld0, g1, ld1, g2, ld2, b0, b1, c_prev,
                           c, h, ldh, h2, ldh2, keep, keep_scale, hdrop, ldhd, gates, S, R, rows_h, rows_h2, 1, 0, h_bf16);
    else
        hipLaunchKernelGGL(lstm_fwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g0, ld0, g1, ld1, g2, ld2, b0, b1, c_prev, c,
                           h, ldh, h2, ldh2, keep, keep_scale, hdrop, ldhd, gates, S, R, rows_h, rows_h2, 1, 0, h_bf16);
    return subgc::check_launch("subgc_lstm_fwd");
}
// gate GEMM + cell update with the split-K reduce folded into the cell kernel (training steps, S in the hundreds)
SUBGC_API int subgc_lstm_fwd_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw, int K, float* pre, int64_t ldpre, const float* g1,
                                  int64_t ld1, const float* g2, int64_t ld2, const float* b0, const float* b1, const float* c_prev, float* c,
                                  void* h, int64_t ldh, void* h2, int64_t ldh2, const uint8_t* keep, float keep_scale, void* hdrop,
                                  int64_t ldhd, float* gates, int S, int R, int rows_h, int rows_h2, int bf16_bits, int gemm_flags,
                                  void* workspace, size_t ws_bytes, void* stream) {
    // bf16_bits: bit 0 = x and w are bf16 (subgc_gemm_bf16 arithmetic), bit 1 = the h destinations are bf16
    SUBGC_REQUIRE(S >= 0 && R > 0 && K > 0, "lstm_fwd_gemm: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && w && pre && c && h && ldpre >= 4 * R, "lstm_fwd_gemm: null pointer / scratch too narrow");
    hipStream_t s = (hipStream_t)stream;
    float* const ws = static_cast<float*>(workspace);
    const int xb16 = bf16_bits & 1, h_bf16 = (bf16_bits >> 1) & 1;
    int parts = 0;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = R % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && ldh % 4 == 0 && ldh2 % 4 == 0 && ldhd % 4 == 0 && al(g1) && al(g2) && al(b0) &&
                     al(b1) && al(c_prev) && al(c) && al(h) && al(h2) && al(hdrop) && al(gates) && (reinterpret_cast<uintptr_t>(keep) & 3) == 0;
    int rc = !vec ? -100
             : xb16 ? subgc::gemm_bf16_nt_partials(static_cast<const uint16_t*>(x), ldx, static_cast<const uint16_t*>(w), ldw, S, 4 * R, K, ws, ws_bytes, s, &parts)
                    : subgc::gemm_nt_partials(static_cast<const float*>(x), ldx, static_cast<const float*>(w), ldw, S, 4 * R, K, gemm_flags, ws, ws_bytes, s, &parts);
    if (rc == -100) {                                                             // not the split-K shape: plain product, then the cell kernel
        rc = xb16 ? subgc_gemm_bf16(0, 1, S, 4 * R, K, static_cast<const uint16_t*>(x), ldx, static_cast<const uint16_t*>(w), ldw, pre, ldpre, nullptr, 0,
                                    nullptr, nullptr, 0, nullptr, 1.f, 0, nullptr, workspace, ws_bytes, stream)
                  : subgc_gemm_f32(0, 1, S, 4 * R, K, static_cast<const float*>(x), ldx, static_cast<const float*>(w), ldw, pre, ldpre, nullptr, nullptr, 0,
                                   nullptr, 1.f, gemm_flags & ~15, nullptr, nullptr, nullptr, workspace, ws_bytes, stream);
        if (rc != SUBGC_OK) return rc;
        return subgc_lstm_fwd(pre, ldpre, g1, ld1, g2, ld2, b0, b1, c_prev, c, h, ldh, h2, ldh2, keep, keep_scale, hdrop, ldhd, gates, S, R, rows_h, rows_h2,
                              h_bf16, stream);
    }
    if (rc != SUBGC_OK) return rc;
    if (rows_h <= 0 || rows_h > S) rows_h = S;
    if (rows_h2 <= 0 || rows_h2 > S) rows_h2 = S;
    const int64_t n = (int64_t)S * R;
    subgc::ProfScope prof(SUBGC_FAM_LSTM, s, 4.0 * n * 12, subgc::lstm_fwd_moved_bytes(n, parts, g1 != nullptr, g2 != nullptr, c_prev != nullptr, h2 != nullptr, hdrop != nullptr, gates != nullptr, h_bf16));
#define SUBGC_LSTM_FWD_PARTS(P_)                                                                                                                         \
    hipLaunchKernelGGL((lstm_fwd_kernel<4, P_>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, (int64_t)4 * R, g1, ld1, g2, ld2, b0, \
                       b1, c_prev, c, h, ldh, h2, ldh2, keep, keep_scale, hdrop, ldhd, gates, S, R, rows_h, rows_h2, parts, (int64_t)S * 4 * R, h_bf16)
    switch (parts) {
        case 2: SUBGC_LSTM_FWD_PARTS(2); break;
        case 3: SUBGC_LSTM_FWD_PARTS(3); break;
        case 4: SUBGC_LSTM_FWD_PARTS(4); break;
        case 5: SUBGC_LSTM_FWD_PARTS(5); break;
        case 6: SUBGC_LSTM_FWD_PARTS(6); break;
        case 8: SUBGC_LSTM_FWD_PARTS(8); break;
        default: SUBGC_LSTM_FWD_PARTS(0); break;
    }
#undef SUBGC_LSTM_FWD_PARTS
    return subgc::check_launch("subgc_lstm_fwd_gemm");
}
namespace {
int lstm_bwd_launch(const float* gates, const float* c_prev, const float* c, PlaneSrc sa, PlaneSrc sb, PlaneSrc sc, const float* dh_drop, int64_t ldd,
                    const uint8_t* keep, float keep_scale, const float* dc, void* dpre, float* dc_prev, int S, int R, int dpre_bf16, hipStream_t s,
                    const char* what) {
    const int64_t n = (int64_t)S * R;
    // moved: the four saved gates, c, c_prev, dc (read) + every d(h) source plane + d(gates) (fp32 or bf16) and dc_prev (written)
    const double planes_in = (sa.p ? sa.n : 0) + (sb.p ? sb.n : 0) + (sc.p ? sc.n : 0) + (dh_drop ? 1 + (keep ? 0.25 : 0.0) : 0.0);
    subgc::ProfScope prof(SUBGC_FAM_LSTM, s, 4.0 * n * 14, 4.0 * n * (4 + 1 + (c_prev ? 1 : 0) + (dc ? 1 : 0) + planes_in + (dpre_bf16 ? 2.0 : 4.0) + 1));
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    auto src_ok = [&](const PlaneSrc& a) { return !a.p || (al(a.p) && a.ld % 4 == 0 && a.stride % 4 == 0); };
    const bool vec = R % 4 == 0 && ldd % 4 == 0 && al(gates) && al(c_prev) && al(c) && src_ok(sa) && src_ok(sb) && src_ok(sc) &&
                     al(dh_drop) && al(dc) && al(dpre) && al(dc_prev) && (reinterpret_cast<uintptr_t>(keep) & 3) == 0;
    if (vec)
        hipLaunchKernelGGL(lstm_bwd_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, gates, c_prev, c, sa, sb, sc,
                           dh_drop, ldd, keep, keep_scale, dc, dpre, dc_prev, S, R, dpre_bf16);
    else
        hipLaunchKernelGGL(lstm_bwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gates, c_prev, c, sa, sb, sc,
                           dh_drop, ldd, keep, keep_scale, dc, dpre, dc_prev, S, R, dpre_bf16);
    return subgc::check_launch(what);
}
}  // namespace

SUBGC_API int subgc_lstm_bwd(const float* gates, const float* c_prev, const float* c, const float* dh_a, int64_t lda, const float* dh_b,
                             int64_t ldb, const float* dh_drop, int64_t ldd, const uint8_t* keep, float keep_scale, const float* dc,
                             void* dpre, float* dc_prev, int S, int R, int dpre_bf16, void* stream) {
    SUBGC_REQUIRE(S >= 0 && R > 0, "lstm_bwd: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(gates && c && dpre && dc_prev, "lstm_bwd: null pointer");
    return lstm_bwd_launch(gates, c_prev, c, PlaneSrc{dh_a, lda, 1, 0, S}, PlaneSrc{dh_b, ldb, 1, 0, S}, PlaneSrc{nullptr, 0, 0, 0, 0}, dh_drop, ldd, keep,
                           keep_scale, dc, dpre, dc_prev, S, R, dpre_bf16, (hipStream_t)stream, "subgc_lstm_bwd");
}

// subgc_lstm_bwd with up to three d(h) sources that are stacks of split-K partial planes (subgc_gemm_*_planes): source i contributes
// sum_{q < n_i} p_i[q * stride_i + s * ld_i + j] to rows s < rows_i (a NULL p_i or n_i = 0: nothing).
SUBGC_API int subgc_lstm_bwd_planes(const float* gates, const float* c_prev, const float* c, const float* p0, int64_t ld0, int n0, int64_t stride0,
                                    int rows0, const float* p1, int64_t ld1, int n1, int64_t stride1, int rows1, const float* p2, int64_t ld2, int n2,
                                    int64_t stride2, int rows2, const float* dh_drop, int64_t ldd, const uint8_t* keep, float keep_scale,
                                    const float* dc, void* dpre, float* dc_prev, int S, int R, int dpre_bf16, void* stream) {
    SUBGC_REQUIRE(S >= 0 && R > 0 && n0 >= 0 && n1 >= 0 && n2 >= 0, "lstm_bwd_planes: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(gates && c && dpre && dc_prev, "lstm_bwd_planes: null pointer");
    return lstm_bwd_launch(gates, c_prev, c, PlaneSrc{n0 ? p0 : nullptr, ld0, n0, stride0, rows0}, PlaneSrc{n1 ? p1 : nullptr, ld1, n1, stride1, rows1},
                           PlaneSrc{n2 ? p2 : nullptr, ld2, n2, stride2, rows2}, dh_drop, ldd, keep, keep_scale, dc, dpre, dc_prev, S, R, dpre_bf16,
                           (hipStream_t)stream, "subgc_lstm_bwd_planes");
}

SUBGC_API int subgc_attn_fwd(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* off,
                             const int32_t* len, void* ctx, int64_t ldctx, float* alpha, int n_stride, int S, int A, int R,
                             int bf16_bits, void* stream) {
    const int ctx_bf16 = bf16_bits & 1, uv_bf16 = (bf16_bits >> 1) & 1;      // bit 0: ctx destination, bit 1: u and v are bf16
    SUBGC_REQUIRE(S >= 0 && A > 0 && R > 0 && n_stride >= 0 && n_stride <= MAXLEN, "attn_fwd: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && v && ah && w_a && b_a && off && len && ctx, "attn_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    if (const int rc = subgc::attn_fwd_vec(u, v, ah, w_a, b_a, off, len, ctx, ldctx, alpha, n_stride, S, A, R, ctx_bf16, uv_bf16, s); rc != -100)
        return rc;
    subgc::set_error("attn_fwd: needs att_hid_size, rnn_size %% 4 == 0 (<= 512 / <= 2048) and 16-byte aligned rows (A=%d R=%d)", A, R);
    return SUBGC_EINVAL;
}
// subgc_attn_fwd whose query rows arrive as the split-K partial planes of the h2att product: row s = q_bias + sum_p (q_planes + p * plane_stride)[s, :];
// the summed rows are written to q_out [S, A] (the backward's `ah`)
SUBGC_API int subgc_attn_fwd_q(const void* u, const void* v, const float* q_planes, int n_planes, int64_t plane_stride, const float* q_bias, float* q_out,
                               const float* w_a, const float* b_a, const int32_t* off, const int32_t* len, void* ctx, int64_t ldctx, float* alpha,
                               int n_stride, int S, int A, int R, int bf16_bits, void* stream) {
    const int ctx_bf16 = bf16_bits & 1, uv_bf16 = (bf16_bits >> 1) & 1;
    SUBGC_REQUIRE(S >= 0 && A > 0 && R > 0 && n_stride >= 0 && n_stride <= MAXLEN && n_planes >= 1 && n_planes <= 16, "attn_fwd_q: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && v && q_planes && q_out && w_a && b_a && off && len && ctx, "attn_fwd_q: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    if (const int rc = subgc::attn_fwd_vec(u, v, q_planes, w_a, b_a, off, len, ctx, ldctx, alpha, n_stride, S, A, R, ctx_bf16, uv_bf16, s,
                                           subgc::QSrc{q_bias, q_out, n_planes, plane_stride}); rc != -100)
        return rc;
    subgc::set_error("attn_fwd_q: needs att_hid_size, rnn_size %% 4 == 0 (<= 512 / <= 2048) and 16-byte aligned rows (A=%d R=%d)", A, R);
    return SUBGC_EINVAL;
}
namespace {
int attn_bwd_any(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len, const float* alpha,
                 int n_stride, const float* dctx, int64_t lddctx, int n_planes, int64_t plane_stride, void* dah, float* du, float* dv, float* dw_a,
                 float* db_a, int S, int A, int R, int bf16_bits, float* dctx_keep, int64_t ldkeep, void* stream, float* de_keep = nullptr) {
    const int dah_bf16 = bf16_bits & 1, uv_bf16 = (bf16_bits >> 1) & 1;      // bit 0: dah destination, bit 1: u and v are bf16
    SUBGC_REQUIRE(S >= 0 && A > 0 && R > 0 && n_stride > 0 && n_stride <= MAXLEN && n_planes >= 1, "attn_bwd: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && v && ah && w_a && off && len && alpha && dctx && dah && dw_a, "attn_bwd: null pointer");
    SUBGC_REQUIRE((du != nullptr) != (de_keep != nullptr), "attn_bwd: exactly one of du (accumulated per step) and de_keep (deferred: subgc_attn_du_accum)");
    SUBGC_REQUIRE(!dctx_keep || ldkeep >= R, "attn_bwd: dctx_keep rows too short");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    if (const int rc = subgc::attn_bwd_vec(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dah, du, dv, dw_a, db_a, S, A, R, dah_bf16,
                                           uv_bf16, dctx_keep, ldkeep, s, n_planes, plane_stride, de_keep);
        rc != -100)
        return rc;
    subgc::set_error("attn_bwd: needs att_hid_size, rnn_size %% 4 == 0 (<= 1024 / <= 2048) and 16-byte aligned rows (A=%d R=%d)", A, R);
    return SUBGC_EINVAL;
}
}  // namespace
SUBGC_API int subgc_attn_bwd(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                             const float* alpha, int n_stride, const float* dctx, int64_t lddctx, void* dah, float* du, float* dv,
                             float* dw_a, float* db_a, int S, int A, int R, int bf16_bits, float* dctx_keep, int64_t ldkeep, void* stream) {
    return attn_bwd_any(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, 1, 0, dah, du, dv, dw_a, db_a, S, A, R, bf16_bits, dctx_keep, ldkeep, stream);
}
// subgc_attn_bwd whose d(ctx) is the sum of `dctx_planes` split-K partial planes dctx + q * plane_stride (subgc_gemm_*_planes)
SUBGC_API int subgc_attn_bwd_planes(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                                    const float* alpha, int n_stride, const float* dctx, int64_t lddctx, int dctx_planes, int64_t plane_stride,
                                    void* dah, float* du, float* dv, float* dw_a, float* db_a, int S, int A, int R, int bf16_bits,
                                    float* dctx_keep, int64_t ldkeep, void* stream) {
    return attn_bwd_any(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dctx_planes, plane_stride, dah, du, dv, dw_a, db_a, S, A, R, bf16_bits,
                        dctx_keep, ldkeep, stream);
}

// subgc_attn_bwd_planes with d(u) DEFERRED: the step files its d(e) row in de_keep [S, n_stride] instead of read-modify-writing d(u)
SUBGC_API int subgc_attn_bwd_planes_de(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                                       const float* alpha, int n_stride, const float* dctx, int64_t lddctx, int dctx_planes, int64_t plane_stride,
                                       void* dah, float* de_keep, float* dv, float* dw_a, float* db_a, int S, int A, int R, int bf16_bits,
                                       float* dctx_keep, int64_t ldkeep, void* stream) {
    return attn_bwd_any(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dctx_planes, plane_stride, dah, nullptr, dv, dw_a, db_a, S, A, R, bf16_bits,
                        dctx_keep, ldkeep, stream, de_keep);
}

SUBGC_API int subgc_attn_du_accum(const void* u, int uv_bf16, const float* ah, const float* de, int n_stride, const int32_t* step_off, int T,
                                  const int32_t* off, const int32_t* len, const float* w_a, float* du, int S, int A, void* stream) {
    SUBGC_REQUIRE(S >= 0 && A > 0 && T >= 1 && n_stride > 0 && n_stride <= MAXLEN, "attn_du_accum: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && ah && de && step_off && off && len && w_a && du, "attn_du_accum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    const int rc = subgc::attn_du_accum_vec(u, uv_bf16, ah, de, n_stride, step_off, T, off, len, w_a, du, S, A, s);
    SUBGC_REQUIRE(rc != -100, "attn_du_accum: needs A %% 4 == 0, A <= 1024 and 16-byte aligned rows");
    return rc;
}

SUBGC_API int subgc_attn_dv_accum(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off, int T,
                                  const int32_t* off, const int32_t* len, float* dv, int S, int R, void* stream) {
    SUBGC_REQUIRE(S >= 0 && R > 0 && T >= 1 && n_stride > 0 && n_stride <= MAXLEN && lddctx >= R, "attn_dv_accum: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(alpha && dctx && step_off && off && len && dv, "attn_dv_accum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    const int rc = subgc::attn_dv_accum_vec(alpha, n_stride, dctx, lddctx, step_off, T, off, len, dv, S, R, s);
    SUBGC_REQUIRE(rc != -100, "attn_dv_accum: needs R %% 4 == 0, R <= 2048 and 16-byte aligned rows");
    return rc;
}

SUBGC_API int subgc_log_softmax_rows(float* x, int64_t ldx, int rows, int V, const int32_t* active, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && V > 0 && ldx >= V, "log_softmax_rows: bad sizes");
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x, "log_softmax_rows: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_SOFTMAX, s, 4.0 * rows * (double)V * 2);
    SUBGC_REQUIRE(V <= 256 * 64, "log_softmax_rows: at most %d columns", 256 * 64);
    const int per = (V + 255) / 256;
    if (per <= 4) hipLaunchKernelGGL(log_softmax_kernel<4>, dim3(rows), dim3(256), 0, s, x, ldx, rows, V, active, (float*)nullptr);
    else if (per <= 16) hipLaunchKernelGGL(log_softmax_kernel<16>, dim3(rows), dim3(256), 0, s, x, ldx, rows, V, active, (float*)nullptr);
    else if (per <= 40) hipLaunchKernelGGL(log_softmax_kernel<40>, dim3(rows), dim3(256), 0, s, x, ldx, rows, V, active, (float*)nullptr);
    else hipLaunchKernelGGL(log_softmax_kernel<64>, dim3(rows), dim3(256), 0, s, x, ldx, rows, V, active, (float*)nullptr);
    return subgc::check_launch("subgc_log_softmax_rows");
}
SUBGC_API int subgc_row_lse_f32(const float* x, int64_t ldx, int rows, int V, float* lse, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && V > 0 && ldx >= V, "row_lse: bad sizes");
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && lse, "row_lse: null pointer");
    SUBGC_REQUIRE(V <= 256 * 64, "row_lse: at most %d columns", 256 * 64);
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_SOFTMAX, s, 4.0 * rows * (double)V);
    float* xm = const_cast<float*>(x);                                    // the kernel does not write the row when lse_out is given
    const int per = (V + 255) / 256;
    if (per <= 4) hipLaunchKernelGGL(log_softmax_kernel<4>, dim3(rows), dim3(256), 0, s, xm, ldx, rows, V, (const int32_t*)nullptr, lse);
    else if (per <= 16) hipLaunchKernelGGL(log_softmax_kernel<16>, dim3(rows), dim3(256), 0, s, xm, ldx, rows, V, (const int32_t*)nullptr, lse);
    else if (per <= 40) hipLaunchKernelGGL(log_softmax_kernel<40>, dim3(rows), dim3(256), 0, s, xm, ldx, rows, V, (const int32_t*)nullptr, lse);
    else hipLaunchKernelGGL(log_softmax_kernel<64>, dim3(rows), dim3(256), 0, s, xm, ldx, rows, V, (const int32_t*)nullptr, lse);
    return subgc::check_launch("subgc_row_lse_f32");
}
SUBGC_API int subgc_log_softmax_rows_bwd(const float* logp, const float* dout, void* dlogits, int64_t ld, int rows, int V,
                                         const int32_t* active, int out_bf16, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && V > 0 && ld >= V, "log_softmax_rows_bwd: bad sizes");
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(logp && dout && dlogits, "log_softmax_rows_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_SOFTMAX, s, 4.0 * rows * (double)V * 3);
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(rows), dim3(256), 0, s, logp, dout, dlogits, ld, rows, V, active, out_bf16);
    return subgc::check_launch("subgc_log_softmax_rows_bwd");
}
SUBGC_API int subgc_masked_nll_fwd(const float* logp, const int64_t* target, int64_t t_stride, const float* mask, int64_t m_stride,
                                   float* loss, float* scratch2, int S, int T, int V, const float* den_override, const float* lse,
                                   void* stream) {
    SUBGC_REQUIRE(S > 0 && T > 0 && V > 0, "masked_nll_fwd: bad sizes");
    SUBGC_REQUIRE(logp && target && mask && loss && scratch2, "masked_nll_fwd: null pointer");
    SUBGC_DEBUG_RANGE(target, 8, S, T, t_stride, 0, V - 1, -1, "masked_nll_fwd: target (word ids)", stream);
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, target, t_stride, mask, m_stride, loss,
                       scratch2, S, T, V, den_override, lse);
    return subgc::check_launch("subgc_masked_nll_fwd");
}
SUBGC_API int subgc_masked_nll_bwd(const int64_t* target, int64_t t_stride, const float* mask, int64_t m_stride, const float* scratch2,
                                   const float* dloss, float* dlogp, int S, int T, int V, void* stream) {
    SUBGC_REQUIRE(S > 0 && T > 0 && V > 0, "masked_nll_bwd: bad sizes");
    SUBGC_REQUIRE(target && mask && scratch2 && dloss && dlogp, "masked_nll_bwd: null pointer");
    SUBGC_DEBUG_RANGE(target, 8, S, T, t_stride, 0, V - 1, -1, "masked_nll_bwd: target (word ids)", stream);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)S * T * V;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, s, dlogp, n, 0.f);
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((S * T + 255) / 256), dim3(256), 0, s, target, t_stride, mask, m_stride, scratch2, dloss,
                       dlogp, S, T, V);
    return subgc::check_launch("subgc_masked_nll_bwd");
}
SUBGC_API int subgc_nll_logsoftmax_bwd(const float* logp, const int64_t* target, int64_t t_stride, const float* mask, int64_t m_stride,
                                       const float* scratch2, const float* dloss, void* dlogits, int64_t ld_out, int S, int T, int V,
                                       const int32_t* active, int out_bf16, const float* lse, void* stream) {
    SUBGC_REQUIRE(S > 0 && T > 0 && V > 0 && ld_out >= V, "nll_logsoftmax_bwd: bad sizes");
    SUBGC_REQUIRE(logp && target && mask && scratch2 && dloss && dlogits, "nll_logsoftmax_bwd: null pointer");
    SUBGC_DEBUG_RANGE(target, 8, S, T, t_stride, 0, V - 1, -1, "nll_logsoftmax_bwd: target (word ids)", stream);
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_SOFTMAX, s, 4.0 * S * T * (double)V * 2);
    const int vec = V % 4 == 0 && ld_out % 4 == 0 && (reinterpret_cast<uintptr_t>(logp) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(dlogits) & (out_bf16 ? 7 : 15)) == 0;
    hipLaunchKernelGGL(nll_logsoftmax_bwd_kernel, dim3(S * T), dim3(256), 0, s, logp, target, t_stride, mask, m_stride, scratch2, dloss,
                       dlogits, ld_out, T, V, active, out_bf16, lse, vec);
    return subgc::check_launch("subgc_nll_logsoftmax_bwd");
}
SUBGC_API int subgc_step_active(const int64_t* labels, int64_t l_stride, int S, int T, int32_t* active, void* stream) {
    SUBGC_REQUIRE(S > 0 && T > 0 && T <= 512, "step_active: bad sizes");
    SUBGC_REQUIRE(labels && active, "step_active: null pointer");
    hipLaunchKernelGGL(step_active_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, labels, l_stride, S, T, active);
    return subgc::check_launch("subgc_step_active");
}

SUBGC_API int subgc_decode_pick(const float* logp, int64_t ld, int n, int V, int k, float temp, const float* u, int t, int64_t* seq,
                                float* seqlp, int T, int64_t* next_tok, int32_t* unfinished, int32_t* n_unfinished,
                                const int32_t* prev_count, int raw_logits, void* stream) {
    SUBGC_REQUIRE(n >= 0 && V > 0 && k >= 0 && k <= MAXK && k <= V && t >= 0 && t < T, "decode_pick: bad sizes");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(logp && seq && seqlp && next_tok && unfinished, "decode_pick: null pointer");
    SUBGC_REQUIRE(k == 0 || temp > 0.f, "decode_pick: temperature must be positive");
    SUBGC_REQUIRE(V <= 256 * 64, "decode_pick: at most %d columns", 256 * 64);
    const int per = (V + 255) / 256;
#define LAUNCH(P) hipLaunchKernelGGL(decode_pick_kernel<P>, dim3(n), dim3(256), 0, (hipStream_t)stream, logp, ld, n, V, k, temp, u, t, seq, \
                                     seqlp, T, next_tok, unfinished, n_unfinished, prev_count, raw_logits)
    if (per <= 4) LAUNCH(4);
    else if (per <= 16) LAUNCH(16);
    else if (per <= 40) LAUNCH(40);
    else LAUNCH(64);
#undef LAUNCH
    return subgc::check_launch("subgc_decode_pick");
}

SUBGC_API int subgc_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    SUBGC_REQUIRE(n >= 0 && p >= 0.f && p < 1.f && offset % 4 == 0, "dropout_mask: bad arguments");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(keep, "dropout_mask: null pointer");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed, offset);
    return subgc::check_launch("subgc_dropout_mask");
}

SUBGC_API int subgc_fill_f32(float* x, int64_t n, float value, void* stream) {
    SUBGC_REQUIRE(n >= 0, "fill: bad size");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x, "fill: null pointer");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, value);
    return subgc::check_launch("subgc_fill_f32");
}
SUBGC_API int subgc_copy2d_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, int accumulate, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= cols, "copy2d: bad sizes");
    if (rows == 0 || cols == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && y, "copy2d: null pointer");
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols,
                       accumulate);
    return subgc::check_launch("subgc_copy2d_f32");
}
SUBGC_API int subgc_scatter_add_rows(const float* src, int64_t lds, const int32_t* rows, float* dX, int64_t ldx, int M, int L,
                                     const int32_t* m_dev, void* stream) {
    SUBGC_REQUIRE(M >= 0 && L > 0 && lds >= L && ldx >= L, "scatter_add_rows: bad sizes");
    if (M == 0) return SUBGC_OK;
    SUBGC_REQUIRE(src && rows && dX, "scatter_add_rows: null pointer");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, src, lds, rows, dX, ldx, M, L, m_dev);
    return subgc::check_launch("subgc_scatter_add_rows");
}
SUBGC_API int subgc_relu_bwd(const float* dy, const void* y, float scale, void* dz, int64_t n, int bf16_bits, void* stream) {
    SUBGC_REQUIRE(n >= 0, "relu_bwd: bad size");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(dy && y && dz, "relu_bwd: null pointer");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, scale, dz, n, bf16_bits & 1, (bf16_bits >> 1) & 1);
    return subgc::check_launch("subgc_relu_bwd");
}
SUBGC_API int subgc_gather_rows(const float* src, int64_t lds, const int32_t* rows, void* dst, int64_t ldd, int M, int L,
                                const int32_t* m_dev, int out_bf16, void* stream) {
    SUBGC_REQUIRE(M >= 0 && L > 0 && lds >= L && ldd >= L, "gather_rows: bad sizes");
    if (M == 0) return SUBGC_OK;
    SUBGC_REQUIRE(src && rows && dst, "gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, src, lds, rows, dst, ldd, M, L, m_dev, out_bf16);
    return subgc::check_launch("subgc_gather_rows");
}
SUBGC_API int subgc_gather_rows_keep(const void* src, int64_t lds, const int32_t* rows, const uint8_t* keep, int64_t ldk, float scale, void* dst,
                                     int64_t ldd, int M, int L, const int32_t* m_dev, int bf16_bits, void* stream) {
    SUBGC_REQUIRE(M >= 0 && L > 0 && lds >= L && ldd >= L && (!keep || ldk >= L), "gather_rows_keep: bad sizes");
    if (M == 0) return SUBGC_OK;
    SUBGC_REQUIRE(src && rows && dst, "gather_rows_keep: null pointer");
    hipLaunchKernelGGL(gather_rows_keep_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, src, lds, rows, keep, ldk, scale, dst, ldd, M, L, m_dev,
                       bf16_bits);
    return subgc::check_launch("subgc_gather_rows_keep");
}
SUBGC_API int subgc_gather_rows_multi(int count, const float* s0, int64_t lds0, float* d0, int64_t ldd0, int c0, const float* s1, int64_t lds1,
                                      float* d1, int64_t ldd1, int c1, const float* s2, int64_t lds2, float* d2, int64_t ldd2, int c2,
                                      const float* s3, int64_t lds3, float* d3, int64_t ldd3, int c3, const int32_t* rows, int M, void* stream) {
    SUBGC_REQUIRE(count >= 1 && count <= 4 && M >= 0, "gather_rows_multi: 1..4 tensors");
    if (M == 0) return SUBGC_OK;
    GatherSet g{{s0, s1, s2, s3}, {d0, d1, d2, d3}, {lds0, lds1, lds2, lds3}, {ldd0, ldd1, ldd2, ldd3}, {c0, c1, c2, c3}};
    for (int k = 0; k < count; ++k)
        SUBGC_REQUIRE(g.src[k] && g.dst[k] && g.cols[k] > 0 && g.lds[k] >= g.cols[k] && g.ldd[k] >= g.cols[k], "gather_rows_multi: bad tensor %d", k);
    SUBGC_REQUIRE(rows, "gather_rows_multi: null rows");
    hipLaunchKernelGGL(gather_rows_multi_kernel<int32_t>, dim3(M, count), dim3(256), 0, (hipStream_t)stream, g, rows, M);
    return subgc::check_launch("subgc_gather_rows_multi");
}
// the same gather with int64 row ids (the NMS survivor list, torch's index dtype): decode-time selection of the kept sub-graphs' read-out
// rows / node lists / node counts / scores in one launch.  Columns are 4-byte words: int64 / int32 tensors pass as their float32 views.
SUBGC_API int subgc_gather_rows_multi_i64(int count, const float* s0, int64_t lds0, float* d0, int64_t ldd0, int c0, const float* s1, int64_t lds1,
                                          float* d1, int64_t ldd1, int c1, const float* s2, int64_t lds2, float* d2, int64_t ldd2, int c2,
                                          const float* s3, int64_t lds3, float* d3, int64_t ldd3, int c3, const int64_t* rows, int M, void* stream) {
    SUBGC_REQUIRE(count >= 1 && count <= 4 && M >= 0, "gather_rows_multi_i64: 1..4 tensors");
    if (M == 0) return SUBGC_OK;
    GatherSet g{{s0, s1, s2, s3}, {d0, d1, d2, d3}, {lds0, lds1, lds2, lds3}, {ldd0, ldd1, ldd2, ldd3}, {c0, c1, c2, c3}};
    for (int k = 0; k < count; ++k)
        SUBGC_REQUIRE(g.src[k] && g.dst[k] && g.cols[k] > 0 && g.lds[k] >= g.cols[k] && g.ldd[k] >= g.cols[k], "gather_rows_multi_i64: bad tensor %d", k);
    SUBGC_REQUIRE(rows, "gather_rows_multi_i64: null rows");
    hipLaunchKernelGGL(gather_rows_multi_kernel<int64_t>, dim3(M, count), dim3(256), 0, (hipStream_t)stream, g, rows, M);
    return subgc::check_launch("subgc_gather_rows_multi_i64");
}
SUBGC_API int subgc_sumsq_f32(const float* g, int64_t n, float* sumsq, void* stream) {
    SUBGC_REQUIRE(n >= 0, "sumsq: bad size");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(g && sumsq, "sumsq: null pointer");
    if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        // one same-address float atomic per workgroup: 2048 of them queue memory-side for ~20 us (a 13 MB slice took 35 us), 512 do not
        hipLaunchKernelGGL(sumsq_vec_kernel, dim3(std::min(ew_grid(n / 4), 512)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(g), n / 4, sumsq);
        return subgc::check_launch("subgc_sumsq_f32");
    }
    hipLaunchKernelGGL(sumsq_kernel, dim3(std::min(ew_grid(n), 1024)), dim3(256), 0, (hipStream_t)stream, g, n, sumsq);
    return subgc::check_launch("subgc_sumsq_f32");
}
namespace {
template <bool ZERO>
int clip_adam_launch(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, uint16_t* p_bf16, void* stream) {
    SUBGC_REQUIRE(n >= 0 && step >= 1 && grad_scale > 0.f, "clip_adam_step: bad arguments");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(p && g && m && v && sumsq, "clip_adam_step: null pointer");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (n % 4 == 0 && al(p) && al(g) && al(m) && al(v) && (reinterpret_cast<uintptr_t>(p_bf16) & 7) == 0) {
        hipLaunchKernelGGL(clip_adam_vec_kernel<ZERO>, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(p),
                           reinterpret_cast<float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n / 4, sumsq, max_norm, lr,
                           beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, p_bf16);
        return subgc::check_launch("subgc_clip_adam_step");
    }
    hipLaunchKernelGGL(clip_adam_kernel<ZERO>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, grad_scale, p_bf16);
    return subgc::check_launch("subgc_clip_adam_step");
}
}  // namespace

SUBGC_API int subgc_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, uint16_t* p_bf16,
                                   void* stream) {
    return clip_adam_launch<false>(p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, step, grad_scale, p_bf16, stream);
}
// the same sweep with optimizer.zero_grad() (train.py: called once per iteration) folded in: g is left ZEROED instead of scaled and clipped,
// so the next step needs no fill pass over the gradient buffer
SUBGC_API int subgc_clip_adam_step_zero(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, uint16_t* p_bf16,
                                        void* stream) {
    return clip_adam_launch<true>(p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, step, grad_scale, p_bf16, stream);
}

// ---------------------------------------------------------------------------------------------------
// Packed decoder: dst[s, :] = sum over the time steps t at which sentence s is live of src[ot[t] + s, :]
// (the gradient of the loop-invariant fc->gates term; replaces T accumulate-copies).  M_t = ot[t+1] - ot[t] is
// non-increasing, so a row's live steps are a prefix of 0..T-1.
namespace {
__global__ __launch_bounds__(256) void packed_time_sum_kernel(const void* __restrict__ src, const int32_t* __restrict__ ot, int T, int S,
                                                              int C4, float* __restrict__ dst, int b16) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)S * C4) return;
    const int s = (int)(q / C4), c = (int)(q % C4) * 4;
    const int64_t C = (int64_t)C4 * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
        const int o = ot[t];
        if (s >= ot[t + 1] - o) break;
        const int64_t at = (int64_t)(o + s) * C + c;
        const float4 v = b16 ? subgc_load4_bf(static_cast<const uint16_t*>(src) + at) : *reinterpret_cast<const float4*>(static_cast<const float*>(src) + at);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dst + (int64_t)s * C + c) = a;
}
}  // namespace

SUBGC_API int subgc_packed_time_sum(const void* src, const int32_t* offsets, int T, int S, int C, float* dst, int src_bf16, void* stream) {
    SUBGC_REQUIRE(T >= 0 && S >= 0 && C > 0 && C % 4 == 0, "packed_time_sum: bad sizes (C % 4 == 0)");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(src && offsets && dst, "packed_time_sum: null pointer");
    SUBGC_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & (src_bf16 ? 7 : 15)) == 0,
                  "packed_time_sum: 16-byte alignment (8 for a bf16 source)");
    const int64_t n = (int64_t)S * (C / 4);
    hipLaunchKernelGGL(packed_time_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, offsets, T, S, C / 4,
                       dst, src_bf16);
    return subgc::check_launch("subgc_packed_time_sum");
}
