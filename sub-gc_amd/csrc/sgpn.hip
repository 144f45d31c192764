// sGPN kernels: fused sub-graph gather + masked max/mean pooling (fwd + bwd), score-head tail
// (dot + sigmoid + BCE, fwd + bwd) and sub-graph NMS on 256-bit node-set masks.
//
// Reference op sites: gpn.py:152-172 (advanced-index gather of [G,N,L]), :174-185 (diagonal bmm,
// max, mean), :54-57 (score MLP tail + BCELoss), :108-150 (python-set NMS).  The gathered tensor
// (388 MB at B=128) is never materialised: each workgroup reads the node rows it needs.
#include "common.h"

namespace {

constexpr int MAXN = 256;   // node slots per sub-graph the pooling kernels keep in LDS

// grid (L/256, G): thread = one feature column of one sub-graph
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ X, const int64_t* __restrict__ idx,
                                                       int64_t idx_stride, const float* __restrict__ w, int64_t w_g,
                                                       int64_t w_i, const float* __restrict__ denom,
                                                       const int32_t* __restrict__ img, float* __restrict__ out,
                                                       int32_t* __restrict__ argmax, int G, int N, int L) {
    __shared__ int row_s[MAXN];
    __shared__ float w_s[MAXN];
    const int g = blockIdx.y;
    const int64_t base = (int64_t)img[g] * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        int64_t n = idx[(int64_t)g * idx_stride + i];
        n = n < 0 ? 0 : (n >= N ? N - 1 : n);
        row_s[i] = (int)(base + n);
        w_s[i] = w[(int64_t)g * w_g + (int64_t)i * w_i];
    }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= L) return;
    float mx = -INFINITY, sum = 0.f;
    int am = 0;
    for (int i = 0; i < N; ++i) {
        const float wi = w_s[i];
        const float v = wi != 0.f ? wi * X[(int64_t)row_s[i] * L + col] : 0.f;
        if (v > mx) { mx = v; am = i; }     // strict >: first maximum, like torch.max on CPU
        sum += v;
    }
    out[(int64_t)g * 2 * L + col] = mx;
    out[(int64_t)g * 2 * L + L + col] = sum / denom[g];
    if (argmax) argmax[(int64_t)g * L + col] = am;
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                       int64_t idx_stride, const float* __restrict__ w, int64_t w_g,
                                                       int64_t w_i, const float* __restrict__ denom,
                                                       const int32_t* __restrict__ img, const int32_t* __restrict__ argmax,
                                                       float* __restrict__ dX, int G, int N, int L) {
    __shared__ int row_s[MAXN];
    __shared__ float w_s[MAXN];
    const int g = blockIdx.y;
    const int64_t base = (int64_t)img[g] * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        int64_t n = idx[(int64_t)g * idx_stride + i];
        n = n < 0 ? 0 : (n >= N ? N - 1 : n);
        row_s[i] = (int)(base + n);
        w_s[i] = w[(int64_t)g * w_g + (int64_t)i * w_i];
    }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= L) return;
    const float dmax = dout[(int64_t)g * 2 * L + col];
    const float dmean = dout[(int64_t)g * 2 * L + L + col] / denom[g];
    const int am = argmax[(int64_t)g * L + col];
    for (int i = 0; i < N; ++i) {
        const float wi = w_s[i];
        if (wi == 0.f) continue;
        const float gsum = wi * (dmean + (i == am ? dmax : 0.f));
        unsafeAtomicAdd(dX + (int64_t)row_s[i] * L + col, gsum);
    }
}

// one wave per sub-graph: z = <hid * keep * scale, w2> + b2 ; score = sigmoid(z)
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ hid, const uint8_t* __restrict__ keep,
                                                        float scale, const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ score, int G, int H) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    float acc = 0.f;
    for (int h = lane; h < H; h += 64) {
        float v = hid[(int64_t)g * H + h];
        if (keep) v = keep[(int64_t)g * H + h] ? v * scale : 0.f;
        acc += v * w2[h];
    }
    acc = wave_sum(acc);
    if (lane == 0) score[g] = sigmoidf_(acc + b2[0]);
}
// mean BCE against target = [1]*(G/2) ++ [0]*(G/2), logs clamped at -100 (nn.BCELoss)
__global__ __launch_bounds__(256) void bce_mean_kernel(const float* __restrict__ score, float* __restrict__ loss, int G) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float s = score[g];
        acc += (g < G / 2) ? -fmaxf(logf(s), -100.f) : -fmaxf(logf(1.f - s), -100.f);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) loss[0] = acc / (float)G;
}
// One workgroup per chunk of SB_CHUNK sub-graphs, threads along H: dw2[h] is accumulated in a register over the
// chunk and leaves as ONE atomic per (chunk, h) -- G x H same-address atomics (one wave per sub-graph) cost 105 us
// at G = 2560, H = 512, ten times the rest of the kernel.
constexpr int SB_CHUNK = 16;
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ hid, const uint8_t* __restrict__ keep, float scale,
                                                        const float* __restrict__ w2, const float* __restrict__ score,
                                                        const float* __restrict__ dloss, float* __restrict__ dhid,
                                                        float* __restrict__ dw2, float* __restrict__ db2, int G, int H) {
    __shared__ float dz_s[SB_CHUNK];
    const int g0 = blockIdx.x * SB_CHUNK, n = min(SB_CHUNK, G - g0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int g = g0 + i;
        const float s = score[g], t = g < G / 2 ? 1.f : 0.f;
        // BCE backward of torch: (s - t) / max((1 - s) s, 1e-12) / G, then sigmoid': s (1 - s)
        dz_s[i] = dloss[0] * (s - t) / fmaxf((1.f - s) * s, 1e-12f) / (float)G * (s * (1.f - s));
    }
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        const float w = w2[h];
        float acc = 0.f;
        for (int i = 0; i < n; ++i) {
            const int64_t q = (int64_t)(g0 + i) * H + h;
            const float k = keep ? (keep[q] ? scale : 0.f) : 1.f;
            dhid[q] = dz_s[i] * w * k;
            acc += dz_s[i] * hid[q] * k;
        }
        unsafeAtomicAdd(dw2 + h, acc);
    }
    if (threadIdx.x == 0) {
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += dz_s[i];
        unsafeAtomicAdd(db2, sum);
    }
}
__global__ void zero_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ------------------------------------------------------------------ NMS
constexpr int W = SUBGC_NMS_WORDS;
__device__ __forceinline__ bool iou_gt(const uint64_t* a, const uint64_t* b, double thres) {
    int inter = 0, uni = 0, na = 0, nb = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        inter += __popcll(a[w] & b[w]); uni += __popcll(a[w] | b[w]);
        na += __popcll(a[w]); nb += __popcll(b[w]);
    }
    if (na == 0 || nb == 0) return false;       // gpn.py:145-146: an empty set never overlaps
    return (double)inter / (double)uni > thres;
}
// single workgroup of 1024 threads.  scratch: masks [M][W] u64, order [M] i32, flag [M] i32
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ score, const int64_t* __restrict__ idx,
                                                   int64_t idx_stride, const int32_t* __restrict__ len, int M, int N,
                                                   double thres, int max_keep, int64_t* __restrict__ keep,
                                                   int32_t* __restrict__ n_keep, uint64_t* __restrict__ masks,
                                                   int32_t* __restrict__ order, int32_t* __restrict__ flag,
                                                   const int32_t* __restrict__ offsets) {
    extern __shared__ unsigned char removed[];   // [M]
    const int t = threadIdx.x, nt = blockDim.x;
    if (offsets) {                               // batched form: workgroup b owns candidates offsets[b] .. offsets[b+1] of image b
        const int g0 = offsets[blockIdx.x];
        M = offsets[blockIdx.x + 1] - g0;
        score += g0; idx += (int64_t)g0 * idx_stride; len += g0; keep += g0; n_keep += blockIdx.x;
        masks += (int64_t)g0 * W; order += g0; flag += g0;
    }
    for (int m = t; m < M; m += nt) {
        uint64_t bits[W] = {0};
        const int l = min(len[m], N);
        for (int i = 0; i < l; ++i) {
            const int64_t n = idx[(int64_t)m * idx_stride + i];
            if (n >= 0 && n < 64 * W) bits[n >> 6] |= 1ull << (n & 63);
        }
        for (int w = 0; w < W; ++w) masks[(int64_t)m * W + w] = bits[w];
        // rank by (score desc, index desc): == np.argsort(score, kind='stable')[::-1]
        const float s = score[m];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const float sj = score[j];
            rank += (sj > s) || (sj == s && j > m);
        }
        order[rank] = m;
        flag[m] = 0;
        removed[m] = 0;
    }
    __syncthreads();
    int kept = 0;                                // every thread counts the kept candidates itself: the value is uniform
    for (int i = 0; i < M; ++i) {
        if (removed[i]) continue;                // uniform: last writer of removed[] was followed by a barrier
        const int cur = order[i];
        if (t == 0) flag[cur] = 1;
        ++kept;
        uint64_t a[W];
        for (int w = 0; w < W; ++w) a[w] = masks[(int64_t)cur * W + w];
        for (int j = i + 1 + t; j < M; j += nt) {
            if (removed[j]) continue;
            if (iou_gt(a, masks + (int64_t)order[j] * W, thres)) removed[j] = 1;
        }
        __syncthreads();
        if (kept >= max_keep) break;             // later survivors cannot enter the first max_keep
    }
    __syncthreads();
    if (t == 0) {
        int c = 0;
        for (int m = 0; m < M; ++m)
            if (flag[m]) keep[c++] = m;
        n_keep[0] = c;
    }
}

}  // namespace

SUBGC_API int subgc_subgraph_pool_fwd(const float* X, const int64_t* idx, int64_t idx_stride, const float* w, int64_t w_gstride,
                                      int64_t w_istride, const float* denom, const int32_t* img, float* out, int32_t* argmax,
                                      int G, int N, int L, void* stream) {
    SUBGC_REQUIRE(G >= 0 && N > 0 && N <= MAXN && L > 0, "subgraph_pool_fwd: bad sizes G=%d N=%d L=%d", G, N, L);
    if (G == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && idx && w && denom && img && out, "subgraph_pool_fwd: null pointer");
    SUBGC_DEBUG_RANGE(idx, 8, G, N, idx_stride, 0, N - 1, -1, "subgraph_pool_fwd: idx (node lists of the sub-graphs)", stream);
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_POOL, s, 4.0 * G * L * 3.0);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((L + 255) / 256, G), dim3(256), 0, s, X, idx, idx_stride, w, w_gstride, w_istride,
                       denom, img, out, argmax, G, N, L);
    return subgc::check_launch("subgc_subgraph_pool_fwd");
}

SUBGC_API int subgc_subgraph_pool_bwd(const float* dout, const int64_t* idx, int64_t idx_stride, const float* w, int64_t w_gstride,
                                      int64_t w_istride, const float* denom, const int32_t* img, const int32_t* argmax, float* dX,
                                      int G, int N, int L, void* stream) {
    SUBGC_REQUIRE(G >= 0 && N > 0 && N <= MAXN && L > 0, "subgraph_pool_bwd: bad sizes");
    if (G == 0) return SUBGC_OK;
    SUBGC_REQUIRE(dout && idx && w && denom && img && argmax && dX, "subgraph_pool_bwd: null pointer");
    SUBGC_DEBUG_RANGE(idx, 8, G, N, idx_stride, 0, N - 1, -1, "subgraph_pool_bwd: idx", stream);
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_POOL, s, 4.0 * G * L * 3.0);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((L + 255) / 256, G), dim3(256), 0, s, dout, idx, idx_stride, w, w_gstride, w_istride,
                       denom, img, argmax, dX, G, N, L);
    return subgc::check_launch("subgc_subgraph_pool_bwd");
}

SUBGC_API int subgc_gpn_score_fwd(const float* hid, const uint8_t* keep, float keep_scale, const float* w2, const float* b2,
                                  float* score, float* loss, int G, int H, void* stream) {
    SUBGC_REQUIRE(G >= 0 && H > 0, "gpn_score_fwd: bad sizes");
    if (G == 0) return SUBGC_OK;
    SUBGC_REQUIRE(hid && w2 && b2 && score, "gpn_score_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(score_fwd_kernel, dim3((G + 3) / 4), dim3(256), 0, s, hid, keep, keep_scale, w2, b2, score, G, H);
    if (loss) hipLaunchKernelGGL(bce_mean_kernel, dim3(1), dim3(256), 0, s, (const float*)score, loss, G);
    return subgc::check_launch("subgc_gpn_score_fwd");
}

SUBGC_API int subgc_gpn_score_bwd(const float* hid, const uint8_t* keep, float keep_scale, const float* w2, const float* score,
                                  const float* dloss, float* dhid, float* dw2, float* db2, int G, int H, void* stream) {
    SUBGC_REQUIRE(G > 0 && H > 0, "gpn_score_bwd: bad sizes");
    SUBGC_REQUIRE(hid && w2 && score && dloss && dhid && dw2 && db2, "gpn_score_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_kernel, dim3((H + 255) / 256), dim3(256), 0, s, dw2, H);
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, s, db2, 1);
    hipLaunchKernelGGL(score_bwd_kernel, dim3((G + SB_CHUNK - 1) / SB_CHUNK), dim3(256), 0, s, hid, keep, keep_scale, w2, score, dloss, dhid, dw2, db2,
                       G, H);
    return subgc::check_launch("subgc_gpn_score_bwd");
}

SUBGC_API int subgc_subgraph_nms(const float* score, const int64_t* idx, int64_t idx_stride, const int32_t* len, int M, int N,
                                 double thres, int max_keep, int64_t* keep, int32_t* n_keep, void* scratch, size_t scratch_bytes,
                                 void* stream) {
    SUBGC_REQUIRE(M >= 0 && N > 0 && max_keep > 0, "subgraph_nms: bad sizes");
    SUBGC_REQUIRE(M <= 65536, "subgraph_nms: at most 65536 candidates (got %d)", M);
    // idx rows are node lists over the image's N nodes (gpn.py:108-150: ids index the [N, .] node table), and a sub-graph is a bit mask
    // of 64 * SUBGC_NMS_WORDS bits: more nodes than that would silently drop ids from the masks and change the kept set
    SUBGC_REQUIRE(N <= 64 * W, "subgraph_nms: node ids must be < %d (64 * SUBGC_NMS_WORDS), got %d nodes per image", 64 * W, N);
    SUBGC_REQUIRE(keep && n_keep, "subgraph_nms: null output");
    SUBGC_DEBUG_RANGE(idx, 8, M, N, idx_stride, 0, N - 1, -1, "subgraph_nms: idx (candidate node lists)", stream);
    hipStream_t s = (hipStream_t)stream;
    SUBGC_REQUIRE(M == 0 || (score && idx && len && scratch), "subgraph_nms: null pointer");
    const size_t need = (size_t)M * (W * 8 + 8);
    SUBGC_REQUIRE(scratch_bytes >= need, "subgraph_nms: scratch too small (%zu < %zu)", scratch_bytes, need);
    uint64_t* masks = (uint64_t*)scratch;
    int32_t* order = (int32_t*)(masks + (size_t)M * W);
    int32_t* flag = order + M;
    const size_t lds = (size_t)((M + 15) / 16 * 16);
    if (int rc = subgc::raise_lds_cached((const void*)nms_kernel, lds, "subgraph_nms")) return rc;
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(1024), lds, s, score, idx, idx_stride, len, M, N, thres, max_keep, keep, n_keep, masks,
                       order, flag, (const int32_t*)nullptr);
    return subgc::check_launch("subgc_subgraph_nms");
}

namespace {
// survivors of the batched NMS in image order: pos = (kept of the images before b) + j  ->  keep[pos] = keep_all[offsets[b] + j] (index inside
// the image), glob[pos] = offsets[b] + keep[pos] (row of the concatenated candidate arrays).  One workgroup; image counts are small.
__global__ __launch_bounds__(256) void nms_compact_kernel(const int64_t* __restrict__ keep_all, const int32_t* __restrict__ n_keep,
                                                          const int32_t* __restrict__ offsets, int images, int total, int64_t* __restrict__ keep,
                                                          int64_t* __restrict__ glob) {
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int b = 0; b < images; ++b) {
        const int base = base_s, nk = n_keep[b], g0 = offsets[b];
        for (int j = threadIdx.x; j < nk && base + j < total; j += blockDim.x) {
            const int64_t k = keep_all[g0 + j];
            keep[base + j] = k;
            glob[base + j] = k + g0;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + nk;
        __syncthreads();
    }
}
}  // namespace

SUBGC_API int subgc_nms_compact(const int64_t* keep_all, const int32_t* n_keep, const int32_t* offsets, int images, int total, int64_t* keep,
                                int64_t* glob, void* stream) {
    SUBGC_REQUIRE(images >= 0 && total >= 0, "nms_compact: bad sizes");
    if (images == 0 || total == 0) return SUBGC_OK;
    SUBGC_REQUIRE(keep_all && n_keep && offsets && keep && glob, "nms_compact: null pointer");
    hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, keep_all, n_keep, offsets, images, total, keep, glob);
    return subgc::check_launch("subgc_nms_compact");
}

SUBGC_API int subgc_subgraph_nms_batched(const float* score, const int64_t* idx, int64_t idx_stride, const int32_t* len,
                                         const int32_t* offsets, int images, int total, int max_m, int N, double thres, int max_keep,
                                         int64_t* keep, int32_t* n_keep, void* scratch, size_t scratch_bytes, void* stream) {
    SUBGC_REQUIRE(images >= 0 && total >= 0 && max_m >= 0 && max_m <= 65536 && N > 0 && max_keep > 0, "subgraph_nms_batched: bad sizes");
    SUBGC_REQUIRE(N <= 64 * W, "subgraph_nms_batched: node ids must be < %d (64 * SUBGC_NMS_WORDS), got %d nodes per image", 64 * W, N);
    if (images == 0) return SUBGC_OK;
    SUBGC_REQUIRE(offsets && keep && n_keep, "subgraph_nms_batched: null pointer");
    SUBGC_REQUIRE(total == 0 || (score && idx && len && scratch), "subgraph_nms_batched: null pointer");
    SUBGC_DEBUG_RANGE(idx, 8, total, N, idx_stride, 0, N - 1, -1, "subgraph_nms_batched: idx (candidate node lists)", stream);
    const size_t need = (size_t)total * (W * 8 + 8);
    SUBGC_REQUIRE(scratch_bytes >= need, "subgraph_nms_batched: scratch too small (%zu < %zu)", scratch_bytes, need);
    uint64_t* masks = (uint64_t*)scratch;
    int32_t* order = (int32_t*)(masks + (size_t)total * W);
    int32_t* flag = order + total;
    const size_t lds = (size_t)((max_m + 15) / 16 * 16);
    if (int rc = subgc::raise_lds_cached((const void*)nms_kernel, lds, "subgraph_nms_batched")) return rc;
    hipLaunchKernelGGL(nms_kernel, dim3(images), dim3(1024), lds, (hipStream_t)stream, score, idx, idx_stride, len, 0, N, thres, max_keep,
                       keep, n_keep, masks, order, flag, offsets);
    return subgc::check_launch("subgc_subgraph_nms_batched");
}
