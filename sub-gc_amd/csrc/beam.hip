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

}  // namespace

SUBGC_API int subgc_row_topk_f32(const float* x, int64_t ld, int rows, int cols, int k, int log_softmax, float* vals, int32_t* idx,
                                 void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && k >= 1 && k <= TOPK_MAX && k <= cols, "row_topk: bad sizes (k <= %d)", TOPK_MAX);
    SUBGC_REQUIRE(cols <= 256 * 64, "row_topk: at most %d columns", 256 * 64);
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && vals && idx, "row_topk: null pointer");
    const int per = (cols + 255) / 256;
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
