// Eval glue over MANY images: what misc/eval_utils.py:105-121 and misc/grd_utils.py:36-47 do per image between the
// model call and the metric scripts -- rank an image's captions by sGPN score, reorder its token rows / kept indices, and (grounding
// experiments) find for every word of the chosen caption the graph node with the largest attention weight.  One launch each over a
// whole decode batch, results in one arena the host reads with a single copy.  Index work: everything is exact.
#include "common.h"

namespace {

// One workgroup per image.  order[seg+r] = image-local index of the r-th best row: score descending, equal scores keep their input
// order (the same total order as rank_desc_kernel, beam.hip); identity: no sorting (Full-GC, eval_utils.py:112-115).  A NaN score ranks
// as -inf (with the index as tie-break the order stays TOTAL: every rank is taken exactly once, ord[] holds no unwritten slot).
__global__ __launch_bounds__(256) void eval_rank_rows_kernel(const float* __restrict__ score, const int64_t* __restrict__ keep,
                                                             const int64_t* __restrict__ seq, int T, const int32_t* __restrict__ seg,
                                                             int identity, int32_t* __restrict__ order, float* __restrict__ score_sorted,
                                                             int32_t* __restrict__ keep_sorted, int32_t* __restrict__ seq_sorted) {
    extern __shared__ float sh[];
    int32_t* ord = reinterpret_cast<int32_t*>(sh);
    const int i = blockIdx.x;
    const int a = seg[i], n = seg[i + 1] - a;
    float* sc = sh + n;
    for (int r = threadIdx.x; r < n; r += blockDim.x) sc[r] = score[a + r];
    __syncthreads();
    auto key = [](float x) { return x != x ? -INFINITY : x; };
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        int rank = r;
        if (!identity) {
            const float v = key(sc[r]);
            rank = 0;
            for (int j = 0; j < n; ++j) {
                const float u = key(sc[j]);
                rank += (u > v) || (u == v && j < r);
            }
        }
        ord[rank] = r;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const int src = ord[r];
        order[a + r] = src;
        score_sorted[a + r] = sc[src];
        keep_sorted[a + r] = (int32_t)keep[a + src];
    }
    for (int q = threadIdx.x; q < n * T; q += blockDim.x) {
        const int r = q / T, t = q - r * T;
        seq_sorted[(int64_t)(a + r) * T + t] = (int32_t)seq[(int64_t)(a + ord[r]) * T + t];
    }
}

// One wave per (word position j, image i): first arg-max over the N attention columns of the chosen caption row at step j.
__global__ __launch_bounds__(64) void grounding_argmax_kernel(const float* __restrict__ AL, int64_t ld_t, int64_t ld_row, int N, int T1,
                                                              const int64_t* __restrict__ seq, int T, const int64_t* __restrict__ idx,
                                                              int64_t ld_idx, const int32_t* __restrict__ seg,
                                                              const int32_t* __restrict__ order, const int32_t* __restrict__ pick,
                                                              int32_t* __restrict__ att2, int32_t* __restrict__ node,
                                                              int32_t* __restrict__ n_words) {
    const int j = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
    const int a = seg[i], n = seg[i + 1] - a;
    int32_t* o_att = att2 + (int64_t)i * T1 + j;
    int32_t* o_node = node + (int64_t)i * T1 + j;
    const int p = pick ? pick[i] : 0;
    if (n <= 0 || p < 0 || p >= n) {                                       // an image that kept no sub-graph: nothing to ground
        if (lane == 0) { *o_att = -1; *o_node = -1; if (j == 0) n_words[i] = 0; }
        return;
    }
    const int g = a + (order ? order[a + p] : p);                           // att_weights[sort_ind[subg_index]] (grd_utils.py:42)
    // len(sent.split()): the words decode_sequence emits = tokens before the first 0 (misc/utils.py:66-73)
    int w = 0;
    {
        const int64_t* row = seq + (int64_t)g * T;
        const bool nz = lane < T ? row[lane] > 0 : false;                  // T <= 64 (checked by the entry point)
        const unsigned long long m = __ballot(nz);
        const unsigned long long stop = ~m;                                // first position that is NOT a word
        w = stop ? __ffsll((long long)stop) - 1 : 64;
        if (w > T) w = T;
    }
    if (lane == 0 && j == 0) n_words[i] = w;
    if (j >= w) {
        if (lane == 0) { *o_att = -1; *o_node = -1; }
        return;
    }
    const float* r = AL + (int64_t)j * ld_t + (int64_t)g * ld_row;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < N; c += 64) {
        const float v = r[c];
        if (v > best) { best = v; bi = c; }                                // strictly greater: a lane keeps its FIRST maximum
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        if (bi == 0x7fffffff) bi = 0;                                      // a row of NaNs: torch.max would return some index; pin 0
        *o_att = bi;
        *o_node = (int32_t)idx[(int64_t)g * ld_idx + bi];
    }
}

}  // namespace

SUBGC_API int subgc_eval_rank_rows(const float* score, const int64_t* keep, const int64_t* seq, int T, const int32_t* seg, int I,
                                   int max_rows, int identity, int32_t* order, float* score_sorted, int32_t* keep_sorted,
                                   int32_t* seq_sorted, void* stream) {
    SUBGC_REQUIRE(I >= 0 && T >= 0 && max_rows >= 0 && max_rows <= 8192, "eval_rank_rows: I, T >= 0, rows per image <= 8192");
    if (I == 0) return SUBGC_OK;
    SUBGC_REQUIRE(score && keep && seq && seg && order && score_sorted && keep_sorted && seq_sorted, "eval_rank_rows: null pointer");
    const size_t lds = (size_t)(2 * max_rows + 2) * sizeof(float);          // 65 544 bytes at the 8192-row limit: above the 64 KiB default
    if (int rc = subgc::raise_lds_cached((const void*)eval_rank_rows_kernel, lds, "eval_rank_rows")) return rc;
    hipLaunchKernelGGL(eval_rank_rows_kernel, dim3(I), dim3(256), lds, (hipStream_t)stream, score,
                       keep, seq, T, seg, identity, order, score_sorted, keep_sorted, seq_sorted);
    return subgc::check_launch("subgc_eval_rank_rows");
}

SUBGC_API int subgc_grounding_argmax(const float* AL, int64_t ld_t, int64_t ld_row, int N, int T1, const int64_t* seq, int T,
                                     const int64_t* idx, int64_t ld_idx, const int32_t* seg, const int32_t* order, const int32_t* pick,
                                     int I, int32_t* att2, int32_t* node, int32_t* n_words, void* stream) {
    SUBGC_REQUIRE(I >= 0 && N >= 1 && T1 >= 1 && T >= 1 && T <= 64, "grounding_argmax: I >= 0, N >= 1, 1 <= T <= 64, T1 >= 1");
    if (I == 0) return SUBGC_OK;
    SUBGC_REQUIRE(AL && seq && idx && seg && att2 && node && n_words, "grounding_argmax: null pointer");
    SUBGC_REQUIRE(ld_row >= N && ld_idx >= N, "grounding_argmax: row strides shorter than N");
    hipLaunchKernelGGL(grounding_argmax_kernel, dim3(T1, I), dim3(64), 0, (hipStream_t)stream, AL, ld_t, ld_row, N, T1, seq, T, idx, ld_idx,
                       seg, order, pick, att2, node, n_words);
    return subgc::check_launch("subgc_grounding_argmax");
}
