// Beam-search support (reference CaptionModel.py:28-175, AttModel.py:179-234).
//
// The reference sorts the whole [beam, V+1] log-prob matrix on every step and then reads only its first
// `beam` columns (CaptionModel.py:60-72).  Here one workgroup per beam row selects the leading k entries
// directly from the raw logits (the log-softmax of AttModel.py:340 is folded in), so a step hands the
// host 2*k numbers per row instead of a normalised and sorted vocabulary row.
#include "common.h"

namespace {

constexpr int TOPK_MAX = 32;

__device__ __forceinline__ void argmax_merge(float& v, int& i, float* sv, int* si) {
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

// Order: value descending, then index ascending (a total order, so "everything after the previous
// pick" is a single comparison and the k passes need no list of used indices).
template <int PER>
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ x, int64_t ld, int cols, int k, int normalise,
                                                       float* __restrict__ vals, int32_t* __restrict__ idx) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ float smf[16];
    const int r = blockIdx.x;
    const float* p = x + (int64_t)r * ld;
    float reg[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = threadIdx.x + j * 256;
        reg[j] = c < cols ? p[c] : -INFINITY;
    }
    float lse = 0.f;
    if (normalise) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < PER; ++j) mx = fmaxf(mx, reg[j]);
        mx = block_max(mx, smf);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) sum += (threadIdx.x + j * 256 < cols) ? expf(reg[j] - mx) : 0.f;
        sum = block_sum(sum, smf);
        lse = mx + logf(sum);
    }
    float pv = INFINITY; int pi = -1;
    for (int q = 0; q < k; ++q) {
        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int c = threadIdx.x + j * 256;
            const float v = reg[j];
            const bool after = v < pv || (v == pv && c > pi);
            if (c < cols && after && (v > bv || bi == 0x7fffffff)) { bv = v; bi = c; }
        }
        argmax_merge(bv, bi, sv, si);
        if (threadIdx.x == 0) {
            vals[(int64_t)r * k + q] = bv - lse;
            idx[(int64_t)r * k + q] = bi;
        }
        pv = bv; pi = bi;
    }
}


// k <= K = 4: ONE selection instead of k dependent workgroup arg-max passes.  Every thread keeps the sorted leading K of its own
// PER columns; two sorted lists a, b merge into the leading K of their union with static indexing -- c[j] = max(a[j], b[K-1-j]) is a
// bitonic sequence holding exactly those K (bitonic split), log2 K compare-exchange stages sort it -- so six butterfly rounds give every
// lane its wave's list and one trip through LDS the workgroup's.  The row maximum of the log-softmax is the first winner.  Same total
// order as the pass kernel (value descending, index ascending), hence the same output; 17 -> ~11 us on 20 rows x 9488 columns.
__device__ __forceinline__ bool kv_gt(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }
template <int K>
__device__ __forceinline__ void kv_merge(float (&av)[K], int (&ai)[K], const float (&bv)[K], const int (&bi)[K]) {
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (kv_gt(bv[K - 1 - j], bi[K - 1 - j], av[j], ai[j])) { av[j] = bv[K - 1 - j]; ai[j] = bi[K - 1 - j]; }
#pragma unroll
    for (int s = K / 2; s >= 1; s >>= 1)
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((j & s) == 0 && kv_gt(av[j + s], ai[j + s], av[j], ai[j])) {
                const float tv = av[j]; const int ti = ai[j];
                av[j] = av[j + s]; ai[j] = ai[j + s]; av[j + s] = tv; ai[j + s] = ti;
            }
}
template <int PER, int K>
__global__ __launch_bounds__(256) void row_topk_merge_kernel(const float* __restrict__ x, int64_t ld, int cols, int k, int normalise,
                                                             float* __restrict__ vals, int32_t* __restrict__ idx) {
    __shared__ float lv[4][K];
    __shared__ int li[4][K];
    __shared__ float smf[16];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* p = x + (int64_t)r * ld;
    float reg[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = threadIdx.x + j * 256;
        reg[j] = c < cols ? p[c] : -INFINITY;
    }
    float tv[K]; int ti[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
#pragma unroll
    for (int j = 0; j < PER; ++j) {                                        // this thread's own leading K: insertion (columns ascend with j, so a strict
        const int c = threadIdx.x + j * 256;                               // comparison keeps the earlier of two equal values in front)
        float v = reg[j]; int ci = c < cols ? c : 0x7fffffff;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (kv_gt(v, ci, tv[q], ti[q])) { const float t0 = tv[q]; const int i0 = ti[q]; tv[q] = v; ti[q] = ci; v = t0; ci = i0; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float ov[K]; int oi[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { ov[j] = __shfl_xor(tv[j], o, 64); oi[j] = __shfl_xor(ti[j], o, 64); }
        kv_merge<K>(tv, ti, ov, oi);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) { lv[wave][j] = tv[j]; li[wave][j] = ti[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j) { tv[j] = lv[0][j]; ti[j] = li[0][j]; }
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        float ov[K]; int oi[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { ov[j] = lv[w][j]; oi[j] = li[w][j]; }
        kv_merge<K>(tv, ti, ov, oi);
    }
    float lse = 0.f;
    if (normalise) {
        const float mx = tv[0];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) sum += (threadIdx.x + j * 256 < cols) ? expf(reg[j] - mx) : 0.f;
        sum = block_sum(sum, smf);
        lse = mx + logf(sum);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q < k) { vals[(int64_t)r * k + q] = tv[q] - lse; idx[(int64_t)r * k + q] = ti[q]; }
    }
}

}  // namespace

SUBGC_API int subgc_row_topk_f32(const float* x, int64_t ld, int rows, int cols, int k, int log_softmax, float* vals, int32_t* idx,
                                 void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && k >= 1 && k <= TOPK_MAX && k <= cols, "row_topk: bad sizes (k <= %d)", TOPK_MAX);
    SUBGC_REQUIRE(cols <= 256 * 64, "row_topk: at most %d columns", 256 * 64);
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && vals && idx, "row_topk: null pointer");
    const int per = (cols + 255) / 256;
    if (k <= 4 && per <= 40) {                                             // beam 2 (test.sh, Sub_GC_Kar): beam + 2 = 4 leading columns
#define LAUNCHM(P) hipLaunchKernelGGL((row_topk_merge_kernel<P, 4>), dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, cols, k, log_softmax, vals, idx)
        if (per <= 4) LAUNCHM(4);
        else if (per <= 16) LAUNCHM(16);
        else LAUNCHM(40);
#undef LAUNCHM
        return subgc::check_launch("subgc_row_topk_f32");
    }
#define LAUNCH(P) hipLaunchKernelGGL(row_topk_kernel<P>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, cols, k, log_softmax, vals, idx)
    if (per <= 4) LAUNCH(4);
    else if (per <= 16) LAUNCH(16);
    else if (per <= 40) LAUNCH(40);
    else LAUNCH(64);
#undef LAUNCH
    return subgc::check_launch("subgc_row_topk_f32");
}

// ---------------------------------------------------------------------------------------------------
// Caption ranking of the eval loop (misc/eval_utils.py:106-108 `torch.sort(subgraph_score, descending=True)`):
// order[r] = index of the r-th best score; equal scores keep their input order (stable).  n <= 8192: one
// workgroup counts, for every element, how many elements precede it -- O(n^2) on a few hundred scores.
namespace {
__global__ __launch_bounds__(256) void rank_desc_kernel(const float* __restrict__ s, int n, int64_t* __restrict__ order,
                                                        float* __restrict__ sorted) {
    extern __shared__ float sh[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sh[i] = s[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = sh[i];
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const float u = sh[j];
            r += (u > v) || (u == v && j < i);
        }
        order[r] = i;
        if (sorted) sorted[r] = v;
    }
}
}  // namespace

SUBGC_API int subgc_rank_desc_f32(const float* score, int n, int64_t* order, float* sorted, void* stream) {
    SUBGC_REQUIRE(n >= 0 && n <= 8192, "rank_desc: 0 <= n <= 8192");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(score && order, "rank_desc: null pointer");
    hipLaunchKernelGGL(rank_desc_kernel, dim3(1), dim3(256), (size_t)n * sizeof(float), (hipStream_t)stream, score, n, order, sorted);
    return subgc::check_launch("subgc_rank_desc_f32");
}

// ------------------------------------------------------------------ beam step (CaptionModel.py:28-94,126-166) on the device
// One workgroup (one wave) per sub-graph walks its beam groups in order, exactly as the host bookkeeping in subgc/beam.py
// (`_beam_step`, itself the restatement of the reference's beam_step) does: same fp32 sums, same stable orders.  With this
// the search loop has no host round trip: top-k -> beam step -> state gather -> decode step, all queued on the stream.
namespace {
constexpr int BS_MAXB = 16, BS_MAXK = 18, BS_MAXT = 64;

struct BeamStepArgs {
    const float* tv; const int32_t* ti;                  // [n][G][bd][kk] leading log-probs / word ids of every beam row (desc)
    int32_t* seq; float* lps; float* sums;               // [n][G][T][bd], [n][G][T][bd], [n][G][bd]
    int32_t* done_cnt; int32_t* done_seq; float* done_lps; float* done_p; int32_t* done_len;   // finished beams, in finishing order
    int64_t* tok; int32_t* src;                          // [n][G][bd]: next input word, source row of every slot
    int t, T, G, bd, kk, unk, constraint, cap;
    float lam;
};

__global__ __launch_bounds__(64) void beam_step_kernel(BeamStepArgs a) {
    __shared__ float v[BS_MAXB * BS_MAXK], un[BS_MAXB * BS_MAXK];
    __shared__ int32_t id[BS_MAXB * BS_MAXK], ord[BS_MAXB * BS_MAXK];
    __shared__ float cp[BS_MAXB * BS_MAXB], cr[BS_MAXB * BS_MAXB];
    __shared__ int32_t cq[BS_MAXB * BS_MAXB], ctok[BS_MAXB * BS_MAXB];
    __shared__ int32_t sel_q[BS_MAXB], sel_tok[BS_MAXB];
    __shared__ float sel_p[BS_MAXB], sel_r[BS_MAXB];
    __shared__ int32_t tmp_seq[BS_MAXT * BS_MAXB];
    __shared__ float tmp_lps[BS_MAXT * BS_MAXB];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int G = a.G, bd = a.bd, kk = a.kk, T = a.T;
    for (int g = 0; g < G; ++g) {
        const int64_t sg = (int64_t)s * G + g;
        const int tau = a.t - g;
        const bool live = g <= a.t && a.t <= T + g - 1;
        if (!live) {                                                          // rows of a sleeping group: fed <bos>/0, left in place
            if (lane < bd) { a.tok[sg * bd + lane] = 0; a.src[sg * bd + lane] = (int32_t)(sg * bd + lane); }
            continue;
        }
        int32_t* seq = a.seq + sg * T * bd;
        float* lps = a.lps + sg * T * bd;
        float* sums = a.sums + sg * bd;
        for (int e = lane; e < bd * kk; e += 64) {                            // augment (:134-137, add_diversity :33-40)
            const int q = e / kk;
            float val = a.tv[sg * bd * kk + e];
            const int32_t w = a.ti[sg * bd * kk + e];
            if (a.constraint && tau > 0 && w == seq[(tau - 1) * bd + q]) val = -INFINITY;
            if (w == a.unk) val -= 1000.f;
            const float u0 = val;
            for (int g2 = 0; g2 < g; ++g2) {
                const int32_t* pseq = a.seq + ((int64_t)s * G + g2) * T * bd + tau * bd;
                for (int b = 0; b < bd; ++b)
                    if (w == pseq[b]) val -= a.lam;
            }
            v[e] = val; un[e] = u0; id[e] = w;
        }
        for (int e = lane; e < tau * bd; e += 64) { tmp_seq[e] = seq[e]; tmp_lps[e] = lps[e]; }
        __syncthreads();
        if (lane < bd) {                                                      // stable descending order of the row (np.argsort(-vals, stable))
            const int q = lane;
            for (int j = 0; j < kk; ++j) ord[q * kk + j] = j;
            for (int j = 1; j < kk; ++j) {
                const int cur = ord[q * kk + j];
                const float key = v[q * kk + cur];
                int i = j - 1;
                while (i >= 0 && v[q * kk + ord[q * kk + i]] < key) { ord[q * kk + i + 1] = ord[q * kk + i]; --i; }
                ord[q * kk + i + 1] = cur;
            }
        }
        __syncthreads();
        if (lane == 0) {
            const int rows = tau == 0 ? 1 : bd, cols = bd < kk ? bd : kk;
            int nc = 0;
            for (int c = 0; c < cols; ++c)
                for (int q = 0; q < rows; ++q) {                              // candidate list in the reference's (column, beam) order
                    const int j = ord[q * kk + c];
                    cp[nc] = sums[q] + v[q * kk + j]; cq[nc] = q; ctok[nc] = id[q * kk + j]; cr[nc] = un[q * kk + j];
                    ++nc;
                }
            for (int vix = 0; vix < bd; ++vix) {                              // leading bd of the stable sort by -p
                int best = -1;
                for (int c = 0; c < nc; ++c)
                    if (cq[c] >= 0 && (best < 0 || cp[c] > cp[best])) best = c;
                if (best < 0) best = 0;                                        // fewer candidates than slots cannot happen (cols*rows >= bd only when tau > 0 or bd == 1) -- keep memory safe
                sel_q[vix] = cq[best] < 0 ? 0 : cq[best]; sel_tok[vix] = ctok[best]; sel_p[vix] = cp[best]; sel_r[vix] = cr[best];
                cq[best] = -1 - cq[best];                                      // taken (q recoverable, never needed again)
            }
        }
        __syncthreads();
        for (int e = lane; e < tau * bd; e += 64) {                           // fork the history: column vix <- old column sel_q[vix]
            const int r = e / bd, vix = e % bd;
            seq[e] = tmp_seq[r * bd + sel_q[vix]];
            lps[e] = tmp_lps[r * bd + sel_q[vix]];
        }
        if (lane < bd) {
            seq[tau * bd + lane] = sel_tok[lane];
            lps[tau * bd + lane] = sel_r[lane];
            sums[lane] = sel_p[lane];
            a.src[sg * bd + lane] = (int32_t)(sg * bd + sel_q[lane]);
            a.tok[sg * bd + lane] = sel_tok[lane];
        }
        __syncthreads();
        if (lane == 0) {                                                      // :150-166 finished beams, in slot order
            for (int vix = 0; vix < bd; ++vix)
                if (sel_tok[vix] == 0 || a.t == T + g - 1) {
                    const int slot = a.done_cnt[sg];
                    if (slot < a.cap) {
                        for (int r = 0; r < T; ++r) {
                            a.done_seq[(sg * a.cap + slot) * T + r] = r <= tau ? seq[r * bd + vix] : 0;
                            a.done_lps[(sg * a.cap + slot) * T + r] = r <= tau ? lps[r * bd + vix] : 0.f;
                        }
                        a.done_p[sg * a.cap + slot] = sums[vix];
                        a.done_len[sg * a.cap + slot] = tau + 1;
                        a.done_cnt[sg] = slot + 1;
                    }
                    sums[vix] = -1000.f;
                }
        }
        __syncthreads();
    }
}
}  // namespace

SUBGC_API int subgc_beam_step(const float* tv, const int32_t* ti, int32_t* seq, float* lps, float* sums, int32_t* done_cnt,
                              int32_t* done_seq, float* done_lps, float* done_p, int32_t* done_len, int64_t* tok, int32_t* src, int n,
                              int t, int T, int G, int bd, int kk, int unk, int constraint, float lam, int cap, void* stream) {
    SUBGC_REQUIRE(n >= 0 && T > 0 && T <= BS_MAXT && G > 0 && bd > 0 && bd <= BS_MAXB && kk >= 1 && kk <= BS_MAXK && cap > 0 && t >= 0,
                  "beam_step: need T <= 64, beams per group <= 16, kk <= 18");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(tv && ti && seq && lps && sums && done_cnt && done_seq && done_lps && done_p && done_len && tok && src, "beam_step: null pointer");
    BeamStepArgs a{tv, ti, seq, lps, sums, done_cnt, done_seq, done_lps, done_p, done_len, tok, src, t, T, G, bd, kk, unk, constraint, cap, lam};
    hipLaunchKernelGGL(beam_step_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, a);
    return subgc::check_launch("subgc_beam_step");
}
