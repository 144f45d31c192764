// On-device batch assembly: the tensor-building part of the reference loader's __getitem__
// (dataloaders/dataloader.py:269-367) -- what the loader does with numpy on 6 worker processes per batch:
//   :276-308  sub-graph node / predicate masks -> compacted index rows padded with the dummy index,
//             prefix att-mask, diagonal 0/1 pooling matrix;
//   :336-354  scene graph padded to obj_num / rel_num rows with the dummy node, one-hot(0) class rows,
//             dummy relation endpoints;
//   :356-363  captions -> labels with <bos>/<eos> slots and the "start + sentence + end" mask.
// The random choice of WHICH sub-graphs to use (:232-267, numpy RNG) stays on the host: it is a few integers.
// All kernels are pure HBM-bound scatter/copy work: threads run along the contiguous dimension.
#include "common.h"

namespace {

// one wave-sized workgroup per mask row: ballot-free ordered compaction through an LDS prefix
__global__ __launch_bounds__(256) void mask_compact_kernel(const uint8_t* __restrict__ mask, int64_t ld, int W, int N, int64_t pad,
                                                           int64_t* __restrict__ ind, float* __restrict__ att_mask,
                                                           float* __restrict__ pool_mtx) {
    __shared__ int pos[1024];
    __shared__ int cnt_s;
    const int g = blockIdx.x;
    const uint8_t* row = mask + (int64_t)g * ld;
    // W <= 1024: thread t owns columns t, t+256, ...; sequential prefix by thread 0 over the (tiny) row
    for (int i = threadIdx.x; i < W; i += blockDim.x) pos[i] = row[i] != 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int i = 0; i < W; ++i) { const int on = pos[i]; pos[i] = on ? c : -1; c += on; }
        cnt_s = c;
    }
    __syncthreads();
    const int cnt = min(cnt_s, N);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        ind[(int64_t)g * N + i] = pad;
        if (att_mask) att_mask[(int64_t)g * N + i] = i < cnt ? 1.f : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += blockDim.x)
        if (pos[i] >= 0 && pos[i] < N) ind[(int64_t)g * N + pos[i]] = i;
    if (pool_mtx) {
        float* pm = pool_mtx + (int64_t)g * N * N;
        for (int q = threadIdx.x; q < N * N; q += blockDim.x) {
            const int r = q / N, c = q - r * N;
            pm[q] = (r == c && r < cnt) ? 1.f : 0.f;
        }
    }
}

// dst[b, r, :] = src[off[b] + r, :] for r < min(off[b+1]-off[b], limit); other rows: one-hot(0) when onehot0 else 0
__global__ __launch_bounds__(256) void pad_rows_f32_kernel(const float* __restrict__ src, const int64_t* __restrict__ off, int R, int C,
                                                           int limit, int onehot0, float* __restrict__ dst) {
    const int b = blockIdx.y, r = blockIdx.x;
    const int64_t o = off[b];
    const int n = min((int)(off[b + 1] - o), limit);
    float* d = dst + ((int64_t)b * R + r) * C;
    if (r < n) {
        const float* s = src + (o + r) * C;
        for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = s[c];
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = (onehot0 && c == 0) ? 1.f : 0.f;
    }
}
__global__ __launch_bounds__(256) void pad_rows_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ off, int R, int C,
                                                           int limit, int64_t pad, int64_t* __restrict__ dst, int B) {
    const int64_t total = (int64_t)B * R * C;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(q % C);
        const int r = (int)((q / C) % R);
        const int b = (int)(q / ((int64_t)R * C));
        const int64_t o = off[b];
        const int n = min((int)(off[b + 1] - o), limit);
        dst[q] = r < n ? src[(o + r) * C + c] : pad;
    }
}

// rows start[g] .. start[g] + count[g] of the packed src -> dst[g, :R, :C], the rest `pad` (segments need not be contiguous or ordered)
__global__ __launch_bounds__(256) void pad_segments_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ start,
                                                               const int64_t* __restrict__ count, int R, int C, int64_t pad,
                                                               int64_t* __restrict__ dst, int G) {
    const int64_t total = (int64_t)G * R * C;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(q % C);
        const int r = (int)((q / C) % R);
        const int g = (int)(q / ((int64_t)R * C));
        const int64_t n = min(count[g], (int64_t)R);
        dst[q] = r < n ? src[(start[g] + r) * C + c] : pad;
    }
}

__global__ __launch_bounds__(64) void caption_labels_kernel(const int64_t* __restrict__ cap, int64_t ld, int S, int Lq,
                                                            int64_t* __restrict__ labels, float* __restrict__ masks) {
    const int s = blockIdx.x;
    __shared__ int nz;
    if (threadIdx.x == 0) nz = 0;
    __syncthreads();
    int mine = 0;
    for (int j = threadIdx.x; j < Lq; j += blockDim.x) mine += cap[(int64_t)s * ld + j] != 0;
    if (mine) atomicAdd(&nz, mine);
    __syncthreads();
    const int keep = nz + 2;
    for (int j = threadIdx.x; j < Lq + 2; j += blockDim.x) {
        labels[(int64_t)s * (Lq + 2) + j] = (j >= 1 && j <= Lq) ? cap[(int64_t)s * ld + j - 1] : 0;
        masks[(int64_t)s * (Lq + 2) + j] = j < keep ? 1.f : 0.f;
    }
}

}  // namespace

SUBGC_API int subgc_mask_compact(const uint8_t* mask, int64_t ld, int G, int W, int N, int64_t pad, int64_t* ind, float* att_mask,
                                 float* pool_mtx, void* stream) {
    SUBGC_REQUIRE(G >= 0 && W > 0 && W <= 1024 && N > 0 && ld >= W, "mask_compact: bad sizes (W <= 1024)");
    if (G == 0) return SUBGC_OK;
    SUBGC_REQUIRE(mask && ind, "mask_compact: null pointer");
    hipLaunchKernelGGL(mask_compact_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, mask, ld, W, N, pad, ind, att_mask, pool_mtx);
    return subgc::check_launch("subgc_mask_compact");
}

SUBGC_API int subgc_pad_rows_f32(const float* src, const int64_t* off, int B, int R, int C, int limit, int onehot0, float* dst,
                                 void* stream) {
    SUBGC_REQUIRE(B >= 0 && R > 0 && C > 0 && limit >= 0 && limit <= R, "pad_rows_f32: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(off && dst && (src || limit == 0), "pad_rows_f32: null pointer");
    hipLaunchKernelGGL(pad_rows_f32_kernel, dim3(R, B), dim3(256), 0, (hipStream_t)stream, src, off, R, C, limit, onehot0, dst);
    return subgc::check_launch("subgc_pad_rows_f32");
}

SUBGC_API int subgc_pad_rows_i64(const int64_t* src, const int64_t* off, int B, int R, int C, int limit, int64_t pad, int64_t* dst,
                                 void* stream) {
    SUBGC_REQUIRE(B >= 0 && R > 0 && C > 0 && limit >= 0 && limit <= R, "pad_rows_i64: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(off && dst && (src || limit == 0), "pad_rows_i64: null pointer");
    const int64_t total = (int64_t)B * R * C;
    hipLaunchKernelGGL(pad_rows_i64_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, src,
                       off, R, C, limit, pad, dst, B);
    return subgc::check_launch("subgc_pad_rows_i64");
}

SUBGC_API int subgc_pad_segments_i64(const int64_t* src, const int64_t* start, const int64_t* count, int G, int R, int C, int64_t pad,
                                     int64_t* dst, void* stream) {
    SUBGC_REQUIRE(G >= 0 && R > 0 && C > 0, "pad_segments_i64: bad sizes");
    if (G == 0) return SUBGC_OK;
    SUBGC_REQUIRE(start && count && dst, "pad_segments_i64: null pointer");
    const int64_t total = (int64_t)G * R * C;
    hipLaunchKernelGGL(pad_segments_i64_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       src, start, count, R, C, pad, dst, G);
    return subgc::check_launch("subgc_pad_segments_i64");
}

SUBGC_API int subgc_caption_labels(const int64_t* captions, int64_t ld, int S, int seq_length, int64_t* labels, float* masks, void* stream) {
    SUBGC_REQUIRE(S >= 0 && seq_length > 0 && ld >= seq_length, "caption_labels: bad sizes");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(captions && labels && masks, "caption_labels: null pointer");
    hipLaunchKernelGGL(caption_labels_kernel, dim3(S), dim3(64), 0, (hipStream_t)stream, captions, ld, S, seq_length, labels, masks);
    return subgc::check_launch("subgc_caption_labels");
}
