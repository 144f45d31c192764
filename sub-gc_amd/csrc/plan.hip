// Index plumbing of a training step, on the device: what the host side used to do with a dozen torch ops per call
// (sort / cat / index_select / permute().contiguous() / sum launches, each a few microseconds of GPU and ~15 us of host).
//   live_plan      the packed decoder's row plan (functions_packed.py): live steps per sentence from the criterion mask and the
//                  reference's early break (AttModel.py:171-172), sentences ordered by live steps (stable, descending),
//                  live rows per step and their prefix, the criterion's denominator (misc/utils.py:123);
//   packed_rows    the packed per-step prefixes of tokens / targets / mask and the sorted per-sentence inputs, in one launch;
//   gpn_prep       gpn.py:43-52 input views ([5B, 2, hb, N] -> pos half then neg half): node lists, pooling weights (the diagonal
//                  of gpn_pool_mtx), node counts, owning image;
//   gpn_select     gpn.py:63-78: per sentence the best positive sub-graph (first max), its node list, node count and read-out row;
//   add_n          the sum autograd needs where a tensor feeds several consumers.
// Everything here is integer / copy work on a few thousand elements: latency-bound single launches, threads along rows.
#include "common.h"

#include <vector>

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void live_plan_kernel(const int64_t* __restrict__ labels, int64_t ldl, const float* __restrict__ mask,
                                                        int64_t ldm, int S, int T, int32_t* __restrict__ perm32, int64_t* __restrict__ perm64,
                                                        int32_t* __restrict__ inv32, int32_t* __restrict__ counts, int32_t* __restrict__ offs,
                                                        float* __restrict__ den) {
    extern __shared__ int live[];                       // [S]
    __shared__ unsigned any_lo, any_hi;
    __shared__ int hist[64], base[64], t_break_s;
    __shared__ float sm[16];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { any_lo = 1u; any_hi = 0u; }        // step 0 always counts (labels[:, 0] is <bos> = 0 for every sentence)
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    float dsum = 0.f;
    unsigned long long anyb = 0ull;
    for (int s = tid; s < S; s += 256) {
        int lv = 0;
        for (int t = 0; t < T; ++t) {
            const float m = mask[(int64_t)s * ldm + t];
            dsum += m;
            if (m > 0.f) lv = t + 1;
            if (labels[(int64_t)s * ldl + t] != 0) anyb |= 1ull << t;
        }
        live[s] = lv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) anyb |= __shfl_xor(anyb, o, 64);
    if (lane == 0) { atomicOr(&any_lo, (unsigned)anyb); atomicOr(&any_hi, (unsigned)(anyb >> 32)); }
    dsum = block_sum(dsum, sm);                         // 0/1 entries: exact in fp32 whatever the order
    __syncthreads();
    if (tid == 0) {
        const unsigned long long a = ((unsigned long long)any_hi << 32) | any_lo;
        int tb = 0;
        while (tb < T && ((a >> tb) & 1ull)) ++tb;      // steps before the first t >= 1 whose labels are all zero
        t_break_s = tb;
        den[0] = dsum;
    }
    __syncthreads();
    const int tb = t_break_s;
    for (int s = tid; s < S; s += 256) {
        const int lv = min(live[s], tb);
        live[s] = lv;
        atomicAdd(&hist[lv], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int above = 0;
        for (int b = T; b >= 0; --b) { base[b] = above; above += hist[b]; }      // descending by live steps
        int o = 0;
        for (int t = 0; t < T; ++t) { counts[t] = base[t]; offs[t] = o; o += base[t]; }   // sentences with live > t
        offs[T] = o;
    }
    __syncthreads();
    if (tid >= 64) return;
    // stable counting sort by one wave: lane b carries the next free slot of bucket b
    int run = lane <= T ? base[lane] : 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int c0 = 0; c0 < S; c0 += 64) {
        const int s = c0 + lane, key = s < S ? live[s] : -1;
        const int run0 = run;
        int pre = 0;
        for (int b = 0; b <= T; ++b) {
            const unsigned long long m = __ballot(key == b);
            if (lane == b) run += __popcll(m);
            if (key == b) pre = __popcll(m & lt);
        }
        const int start = __shfl(run0, key < 0 ? 0 : key, 64);
        if (key >= 0) {
            const int pos = start + pre;
            perm32[pos] = s;
            if (perm64) perm64[pos] = s;
            if (inv32) inv32[s] = pos;
        }
    }
}

// packed row r = offs[t] + j  <->  (sentence perm[j], step t), j < counts[t]
__global__ __launch_bounds__(256) void packed_rows_kernel(const int64_t* __restrict__ labels, int64_t ldl, const int64_t* __restrict__ target,
                                                          int64_t ldt, const float* __restrict__ mask, int64_t ldm,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ offs, int S, int T,
                                                          int64_t* __restrict__ labels_p, int LW, int64_t* __restrict__ tok_flat,
                                                          int64_t* __restrict__ tgt_p, float* __restrict__ msk_p,
                                                          const int32_t* __restrict__ lens, const int64_t* __restrict__ idx, int64_t ldi,
                                                          const int32_t* __restrict__ img, int N, int32_t* __restrict__ lens_p,
                                                          int64_t* __restrict__ idx_p, int32_t* __restrict__ img_p) {
    __shared__ int so[65];
    if (threadIdx.x <= T) so[threadIdx.x] = offs[threadIdx.x];
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int rows = so[T];
    if (q < rows) {
        int t = 0;
        while (t + 1 < T && so[t + 1] <= q) ++t;
        const int src = perm[q - so[t]];
        tok_flat[q] = labels[(int64_t)src * ldl + t];
        tgt_p[q] = target[(int64_t)src * ldt + t];
        msk_p[q] = mask[(int64_t)src * ldm + t];
    }
    if (q < S * LW) {
        const int j = q / LW, c = q - j * LW;
        labels_p[q] = labels[(int64_t)perm[j] * ldl + c];
    }
    if (q < S * N) {
        const int j = q / N, c = q - j * N;
        idx_p[q] = idx[(int64_t)perm[j] * ldi + c];
    }
    if (q < S) { lens_p[q] = lens[perm[q]]; img_p[q] = img[perm[q]]; }
}

// sub-graph g = c * (b5 * hb) + s * hb + h  <-  slot [s, c, h] of the loader's [b5, 2, hb, ...] tensors
__global__ __launch_bounds__(64) void gpn_prep_kernel(const int64_t* __restrict__ obj_ind, const float* __restrict__ pool_mtx,
                                                      const float* __restrict__ att_masks, int b5, int hb, int N, int spi,
                                                      int64_t* __restrict__ idx, float* __restrict__ w, float* __restrict__ denom,
                                                      int32_t* __restrict__ img) {
    const int g = blockIdx.x, per = b5 * hb;
    const int c = g / per, r = g - c * per, s = r / hb, h = r - s * hb;
    const int64_t slot = ((int64_t)s * 2 + c) * hb + h;
    float cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += 64) {
        idx[(int64_t)g * N + i] = obj_ind[slot * N + i];
        w[(int64_t)g * N + i] = pool_mtx[(slot * N + i) * N + i];
        cnt += att_masks[slot * N + i];
    }
    cnt = wave_sum(cnt);
    if (threadIdx.x == 0) { denom[g] = cnt; img[g] = s / spi; }
}

// Decode-time candidate views of MANY images in one launch (gpn.py:84-96: the test branch reads counterpart 0 of the loader's 5 identical
// copies): candidate q of image b = slot q of its [2, M_b, N(, N)] counterpart-0 block (positive slots, then negative ones).
__global__ __launch_bounds__(64) void gpn_test_prep_kernel(const int64_t* __restrict__ table, int images, int N, int64_t* __restrict__ idx,
                                                           float* __restrict__ w, float* __restrict__ denom, int32_t* __restrict__ lens,
                                                           int32_t* __restrict__ img, int32_t* __restrict__ offsets32) {
    const int64_t* __restrict__ offs = table;                              // [images + 1] candidate offsets, then 4 words per image
    const int g = blockIdx.x;
    if (g <= images && threadIdx.x == 0 && offsets32) offsets32[g] = (int32_t)offs[g];
    const int total = (int)offs[images];
    if (g >= total) return;
    int lo = 0, hi = images - 1;                                           // the image that owns candidate g (uniform over the workgroup)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offs[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + (images + 1) + 4 * (int64_t)lo;
    const int64_t* __restrict__ obj = reinterpret_cast<const int64_t*>(e[0]);
    const float* __restrict__ pool = reinterpret_cast<const float*>(e[1]);
    const float* __restrict__ msk = reinterpret_cast<const float*>(e[2]);
    const int64_t q = g - offs[lo];
    float cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += 64) {
        idx[(int64_t)g * N + i] = obj[q * N + i];
        w[(int64_t)g * N + i] = pool[(q * N + i) * N + i];
        cnt += msk[q * N + i];
    }
    cnt = wave_sum(cnt);
    if (threadIdx.x == 0) { denom[g] = cnt; lens[g] = (int32_t)cnt; img[g] = (int32_t)e[3]; }
}

__global__ __launch_bounds__(256) void gpn_select_kernel(const float* __restrict__ score, const int64_t* __restrict__ obj_ind,
                                                         const float* __restrict__ att_masks, const float* __restrict__ read_out, int b5,
                                                         int hb, int N, int W, int64_t* __restrict__ sel_idx, int32_t* __restrict__ lens,
                                                         float* __restrict__ ro_sel, int32_t* __restrict__ sel, int spi,
                                                         int32_t* __restrict__ img_s) {
    __shared__ int best_s;
    __shared__ float sm[16];
    const int s = blockIdx.x;
    if (threadIdx.x == 0) {
        int best = 0;
        float bv = score[(int64_t)s * hb];
        for (int h = 1; h < hb; ++h) {
            const float v = score[(int64_t)s * hb + h];
            if (v > bv) { bv = v; best = h; }           // first max (torch.max, gpn.py:66)
        }
        best_s = best;
        if (sel) sel[s] = best;
        if (img_s) img_s[s] = s / spi;
    }
    __syncthreads();
    const int h = best_s;
    const int64_t slot = ((int64_t)s * 2 + 0) * hb + h;                // counterpart 0 = the positive sub-graphs
    float cnt = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        sel_idx[(int64_t)s * N + i] = obj_ind[slot * N + i];
        cnt += att_masks[slot * N + i];
    }
    cnt = block_sum(cnt, sm);
    if (threadIdx.x == 0) lens[s] = (int32_t)cnt;
    if (ro_sel) {
        const float* src = read_out + ((int64_t)s * hb + h) * W;       // positive half comes first
        for (int i = threadIdx.x; i < W; i += 256) ro_sel[(int64_t)s * W + i] = src[i];
    }
}

// out may alias a
__global__ __launch_bounds__(256) void add_n_kernel(float* out, const float* a, const float* __restrict__ b, const float* __restrict__ c,
                                                    const float* __restrict__ d, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 x = reinterpret_cast<const float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        if (c) { const float4 z = reinterpret_cast<const float4*>(c)[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
        if (d) { const float4 z = reinterpret_cast<const float4*>(d)[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
        reinterpret_cast<float4*>(out)[i] = x;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) out[i] = a[i] + b[i] + (c ? c[i] : 0.f) + (d ? d[i] : 0.f);
}

__global__ __launch_bounds__(256) void fill2d_kernel(float* __restrict__ x, int64_t ld, int rows, int cols, float v) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < (int64_t)rows * cols) x[(q / cols) * ld + q % cols] = v;
}

// lens[r] = (int) sum of row r (0/1 mask rows -> node counts); one wave per row
__global__ __launch_bounds__(256) void row_count_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols, int32_t* __restrict__ lens) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float c = 0.f;
    for (int i = lane; i < cols; i += 64) c += x[(int64_t)r * ld + i];
    c = wave_sum(c);
    if (lane == 0) lens[r] = (int32_t)c;
}

// part[slab][c][col] = sum over the slab's rows r with cls[r] == c of X[r, col]   (C <= 64 classes: thread = column, the per-class
// accumulators of a column live in LDS and are touched by that thread only)
__global__ __launch_bounds__(256) void class_partials_kernel(const float* __restrict__ X, int64_t ldx, const int32_t* __restrict__ cls, int M, int L,
                                                             int C, int rpb, float* __restrict__ part) {
    extern __shared__ float acc[];                     // [C][256]
    const int col = blockIdx.x * 256 + threadIdx.x;
    for (int c = 0; c < C; ++c) acc[c * 256 + threadIdx.x] = 0.f;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    if (col < L) {
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            const int c0 = cls[r], c1 = cls[r + 1], c2 = cls[r + 2], c3 = cls[r + 3];
            const float x0 = X[(int64_t)r * ldx + col], x1 = X[(int64_t)(r + 1) * ldx + col], x2 = X[(int64_t)(r + 2) * ldx + col],
                        x3 = X[(int64_t)(r + 3) * ldx + col];
            if ((unsigned)c0 < (unsigned)C) acc[c0 * 256 + threadIdx.x] += x0;
            if ((unsigned)c1 < (unsigned)C) acc[c1 * 256 + threadIdx.x] += x1;
            if ((unsigned)c2 < (unsigned)C) acc[c2 * 256 + threadIdx.x] += x2;
            if ((unsigned)c3 < (unsigned)C) acc[c3 * 256 + threadIdx.x] += x3;
        }
        for (; r < r1; ++r) {
            const int c0 = cls[r];
            if ((unsigned)c0 < (unsigned)C) acc[c0 * 256 + threadIdx.x] += X[(int64_t)r * ldx + col];
        }
        for (int c = 0; c < C; ++c) part[((int64_t)blockIdx.y * C + c) * L + col] = acc[c * 256 + threadIdx.x];
    }
}

}  // namespace

SUBGC_API int subgc_class_partials(const float* X, int64_t ldx, const int32_t* cls, int M, int L, int C, int slabs, float* part, void* stream) {
    SUBGC_REQUIRE(M > 0 && L > 0 && C > 0 && C <= 64 && slabs > 0 && ldx >= L, "class_partials: bad sizes (at most 64 classes)");
    SUBGC_REQUIRE(X && cls && part, "class_partials: null pointer");
    SUBGC_DEBUG_RANGE(cls, 4, M, 1, 1, 0, C - 1, -1, "class_partials: cls (class ids)", stream);
    const int rpb = (M + slabs - 1) / slabs;
    const size_t lds = (size_t)C * 256 * sizeof(float);
    if (int rc = subgc::raise_lds_cached((const void*)class_partials_kernel, lds, "class_partials")) return rc;
    hipLaunchKernelGGL(class_partials_kernel, dim3((L + 255) / 256, slabs), dim3(256), lds, (hipStream_t)stream, X, ldx, cls, M, L, C, rpb, part);
    return subgc::check_launch("subgc_class_partials");
}

SUBGC_API int subgc_live_plan(const int64_t* labels, int64_t ld_labels, const float* mask, int64_t ld_mask, int S, int T, int32_t* perm32,
                              int64_t* perm64, int32_t* inv32, int32_t* counts, int32_t* offs, float* den, void* stream) {
    SUBGC_REQUIRE(S > 0 && S <= 16384 && T > 0 && T <= 63, "live_plan: S in 1..16384 and T in 1..63, got S=%d T=%d", S, T);
    SUBGC_REQUIRE(labels && mask && perm32 && counts && offs && den, "live_plan: null pointer");
    hipLaunchKernelGGL(live_plan_kernel, dim3(1), dim3(256), (size_t)S * sizeof(int), (hipStream_t)stream, labels, ld_labels, mask, ld_mask, S, T,
                       perm32, perm64, inv32, counts, offs, den);
    return subgc::check_launch("subgc_live_plan");
}

SUBGC_API int subgc_packed_rows(const int64_t* labels, int64_t ld_labels, const int64_t* target, int64_t ld_target, const float* mask,
                                int64_t ld_mask, const int32_t* perm, const int32_t* offs, int S, int T, int64_t* labels_p, int label_cols,
                                int64_t* tok_flat, int64_t* tgt_p, float* msk_p, const int32_t* lens, const int64_t* idx, int64_t ld_idx,
                                const int32_t* img, int N, int32_t* lens_p, int64_t* idx_p, int32_t* img_p, void* stream) {
    SUBGC_REQUIRE(S > 0 && T > 0 && T <= 63 && N > 0 && label_cols > 0, "packed_rows: bad sizes");
    SUBGC_REQUIRE(labels && target && mask && perm && offs && labels_p && tok_flat && tgt_p && msk_p && lens && idx && img && lens_p && idx_p && img_p,
                  "packed_rows: null pointer");
    const int64_t n = std::max<int64_t>((int64_t)S * T, (int64_t)S * std::max(N, label_cols));
    hipLaunchKernelGGL(packed_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, ld_labels, target, ld_target,
                       mask, ld_mask, perm, offs, S, T, labels_p, label_cols, tok_flat, tgt_p, msk_p, lens, idx, ld_idx, img, N, lens_p, idx_p, img_p);
    return subgc::check_launch("subgc_packed_rows");
}

SUBGC_API int subgc_gpn_prep(const int64_t* gpn_obj_ind, const float* gpn_pool_mtx, const float* att_masks, int b5, int hb, int N,
                             int sentences_per_image, int64_t* idx, float* w, float* denom, int32_t* img, void* stream) {
    SUBGC_REQUIRE(b5 > 0 && hb > 0 && N > 0 && sentences_per_image > 0, "gpn_prep: bad sizes");
    SUBGC_REQUIRE(gpn_obj_ind && gpn_pool_mtx && att_masks && idx && w && denom && img, "gpn_prep: null pointer");
    SUBGC_DEBUG_RANGE(gpn_obj_ind, 8, (int64_t)b5 * 2 * hb, N, N, 0, N - 1, -1, "gpn_prep: gpn_obj_ind (node lists of the sampled sub-graphs)", stream);
    if (subgc::debug_bounds())
        if (int rc = subgc::debug_check_mask_agrees(gpn_obj_ind, att_masks, (int64_t)b5 * 2 * hb * N, N - 1, "gpn_prep: gpn_obj_ind vs att_masks", (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(gpn_prep_kernel, dim3(2 * b5 * hb), dim3(64), 0, (hipStream_t)stream, gpn_obj_ind, gpn_pool_mtx, att_masks, b5, hb, N,
                       sentences_per_image, idx, w, denom, img);
    return subgc::check_launch("subgc_gpn_prep");
}

namespace {
// Per-image early break of a BATCHED greedy / top-k decode (AttModel.py:318-319 breaks when no row of the ONE image of a call is
// unfinished): image b = rows bounds[b] .. bounds[b+1]; a row is unfinished after step t while all its tokens up to t are > 0; the image
// stops at the first step after which none of its rows is (brk, else T - 1); log-probs beyond that step are what the reference never
// wrote: zeroed.  out[b] = (brk, any step stopped).
__global__ __launch_bounds__(256) void decode_batch_finish_kernel(const int64_t* __restrict__ seq, float* __restrict__ seqlp, const int32_t* __restrict__ bounds,
                                                                  int T, int32_t* __restrict__ out) {
    __shared__ int alive_s[64];
    __shared__ int brk_s, any_s;
    const int b = blockIdx.x, r0 = bounds[b], r1 = bounds[b + 1];
    for (int t = threadIdx.x; t < T; t += blockDim.x) alive_s[t] = 0;
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x)
        for (int t = 0; t < T && seq[(int64_t)r * T + t] > 0; ++t) atomicOr(&alive_s[t], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int brk = T - 1, any = 0;
        for (int t = 0; t < T; ++t)
            if (!alive_s[t]) { brk = t; any = 1; break; }
        brk_s = brk; any_s = any;
        out[2 * b] = brk; out[2 * b + 1] = any;
    }
    __syncthreads();
    const int brk = brk_s;
    for (int i = threadIdx.x; i < (r1 - r0) * T; i += blockDim.x) {
        const int t = i % T;
        if (t > brk) seqlp[(int64_t)r0 * T + i] = 0.f;
    }
}
}  // namespace

namespace {
// out[b][0 .. words) = the first `words` 4-byte words of the tensor at address ptrs[b]: block 0 (counterpart 0) of every image's loader
// tensor, stacked into one batch array without a host-side concatenation
__global__ __launch_bounds__(256) void gather_blocks_kernel(const int64_t* __restrict__ ptrs, int64_t words, float* __restrict__ out) {
    const float* __restrict__ src = reinterpret_cast<const float*>(ptrs[blockIdx.y]);
    float* __restrict__ dst = out + (int64_t)blockIdx.y * words;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace

SUBGC_API int subgc_gather_blocks(const int64_t* ptrs, int count, int64_t words, float* out, void* stream) {
    SUBGC_REQUIRE(count >= 0 && count <= 65535 && words >= 0, "gather_blocks: bad sizes");
    if (count == 0 || words == 0) return SUBGC_OK;
    SUBGC_REQUIRE(ptrs && out, "gather_blocks: null pointer");
    const unsigned gx = (unsigned)std::min<int64_t>((words + 255) / 256, 64);
    hipLaunchKernelGGL(gather_blocks_kernel, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, ptrs, words, out);
    return subgc::check_launch("subgc_gather_blocks");
}

SUBGC_API int subgc_decode_batch_finish(const int64_t* seq, float* seqlp, const int32_t* bounds, int images, int T, int32_t* out, void* stream) {
    SUBGC_REQUIRE(images >= 0 && T > 0 && T <= 64, "decode_batch_finish: at most 64 steps");
    if (images == 0) return SUBGC_OK;
    SUBGC_REQUIRE(seq && seqlp && bounds && out, "decode_batch_finish: null pointer");
    hipLaunchKernelGGL(decode_batch_finish_kernel, dim3(images), dim3(256), 0, (hipStream_t)stream, seq, seqlp, bounds, T, out);
    return subgc::check_launch("subgc_decode_batch_finish");
}

SUBGC_API int subgc_gpn_test_prep(const int64_t* table, int images, int total, int N, int64_t* idx, float* w, float* denom, int32_t* lens,
                                  int32_t* img, int32_t* offsets32, void* stream) {
    SUBGC_REQUIRE(images > 0 && total >= 0 && N > 0, "gpn_test_prep: bad sizes");
    SUBGC_REQUIRE(table && (total == 0 || (idx && w && denom && lens && img)), "gpn_test_prep: null pointer");
    if (subgc::debug_bounds() && total > 0) {
        // the table lives on the device: bring it over and check every image's candidate node lists (counterpart 0, what the kernel reads)
        std::vector<int64_t> t((size_t)images * 5 + 1);
        if (hipMemcpyAsync(t.data(), table, t.size() * sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
            subgc::set_error("gpn_test_prep: debug check cannot read the address table");
            return SUBGC_ELAUNCH;
        }
        for (int b = 0; b < images; ++b) {
            const int64_t mb = t[b + 1] - t[b];
            SUBGC_REQUIRE(mb >= 0 && t[images] <= total, "gpn_test_prep: candidate offsets are not ascending / exceed `total` [debug bounds mode]");
            const int64_t* obj = reinterpret_cast<const int64_t*>(t[images + 1 + 4 * (size_t)b]);
            const float* msk = reinterpret_cast<const float*>(t[images + 1 + 4 * (size_t)b + 2]);
            if (int rc = subgc::debug_check_range(obj, 8, mb, N, N, 0, N - 1, -1, "gpn_test_prep: gpn_obj_ind (candidate node lists)", (hipStream_t)stream)) return rc;
            if (int rc = subgc::debug_check_mask_agrees(obj, msk, mb * N, N - 1, "gpn_test_prep: gpn_obj_ind vs att_masks", (hipStream_t)stream)) return rc;
        }
    }
    hipLaunchKernelGGL(gpn_test_prep_kernel, dim3(std::max(total, images + 1)), dim3(64), 0, (hipStream_t)stream, table, images, N, idx, w, denom,
                       lens, img, offsets32);
    return subgc::check_launch("subgc_gpn_test_prep");
}

SUBGC_API int subgc_gpn_select(const float* score, const int64_t* gpn_obj_ind, const float* att_masks, const float* read_out, int b5, int hb,
                               int N, int read_out_cols, int64_t* sel_idx, int32_t* lens, float* ro_sel, int32_t* sel,
                               int sentences_per_image, int32_t* img_s, void* stream) {
    SUBGC_REQUIRE(b5 > 0 && hb > 0 && N > 0 && read_out_cols >= 0 && sentences_per_image > 0, "gpn_select: bad sizes");
    SUBGC_REQUIRE(score && gpn_obj_ind && att_masks && sel_idx && lens && (!ro_sel || read_out), "gpn_select: null pointer");
    SUBGC_DEBUG_RANGE(gpn_obj_ind, 8, (int64_t)b5 * 2 * hb, N, N, 0, N - 1, -1, "gpn_select: gpn_obj_ind", stream);
    hipLaunchKernelGGL(gpn_select_kernel, dim3(b5), dim3(256), 0, (hipStream_t)stream, score, gpn_obj_ind, att_masks, read_out, b5, hb, N,
                       read_out_cols, sel_idx, lens, ro_sel, sel, sentences_per_image, img_s);
    return subgc::check_launch("subgc_gpn_select");
}

SUBGC_API int subgc_add_n_f32(float* out, const float* a, const float* b, const float* c, const float* d, int64_t n, void* stream) {
    SUBGC_REQUIRE(n >= 0, "add_n: negative size");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(out && a && b, "add_n: null pointer");
    const bool vec = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                       reinterpret_cast<uintptr_t>(d)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>((n4 + 255) / 256, 1), 4096);
    hipLaunchKernelGGL(add_n_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, a, b, c, d, n4, n);
    return subgc::check_launch("subgc_add_n_f32");
}

SUBGC_API int subgc_fill2d_f32(float* x, int64_t ld, int rows, int cols, float value, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, "fill2d: bad sizes");
    if (rows == 0 || cols == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x, "fill2d: null pointer");
    hipLaunchKernelGGL(fill2d_kernel, dim3((unsigned)(((int64_t)rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, value);
    return subgc::check_launch("subgc_fill2d_f32");
}

SUBGC_API int subgc_row_count_f32(const float* x, int64_t ld, int rows, int cols, int32_t* lens, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols > 0 && ld >= cols, "row_count: bad sizes");
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && lens, "row_count: null pointer");
    hipLaunchKernelGGL(row_count_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, lens);
    return subgc::check_launch("subgc_row_count_f32");
}
