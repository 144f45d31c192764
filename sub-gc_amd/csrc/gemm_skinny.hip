// Skinny (weight-streaming) GEMM for decode: C[M,N] = act(A[M,K] W[N,K]^T + bias), M <= 16.
//
// One caption = one row, so a decode step multiplies <= 16 activation rows by 152 MB of fp32 weights:
// the contraction is HBM-bound (reference: the per-step nn.Linear / nn.LSTMCell calls of
// AttModel.py:332-340,411-423,453 at batch = kept sub-graphs of ONE image).  MFMA tiles would waste
// >= 84 % of their rows, so this kernel runs on the VALU at streaming rate instead:
//   * the A rows (M x K fp32, <= 150 KB) are staged once per workgroup in LDS;
//   * the K dimension is spread over the THREADS of a workgroup (KPT4 float4 per thread per W row), so a
//     W row is read with fully coalesced 16-byte loads, RB rows in flight per thread;
//   * each thread FMAs its W slice against the matching A slice (ds_read_b128, conflict-free: consecutive
//     lanes read consecutive 16 B), then the workgroup reduces with wave shuffles + one small LDS pass;
//   * one workgroup per CU streams a contiguous block of W rows (grid = 256).
#include "common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int THREADS, int KPT4, int RB>
__global__ __launch_bounds__(THREADS) void gemm_skinny_nt_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                                 int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                                 const float* __restrict__ bias, int M, int N, int K, int relu,
                                                                 int rows_per_wg) {
    constexpr int NW = THREADS / 64, MMAX = 16, KS = THREADS * KPT4 * 4;   // KS = padded K covered by the workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];            // A image [M][KS] + reduction scratch
    float* As = smem;
    float* red = smem + (size_t)M * KS;                                     // [NW][RB][MMAX]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int k0 = t * KPT4 * 4;
    const int n_begin = blockIdx.x * rows_per_wg, n_end = min(N, n_begin + rows_per_wg);
    if (n_begin >= n_end) return;

    float4 w[RB][KPT4];
    auto load_w = [&](int n) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int q = 0; q < KPT4; ++q) {
                const int k = k0 + q * 4;
                const bool ok = n + r < n_end && k < K;
                const float4 v = ld4(W + (int64_t)(ok ? n + r : n_begin) * ldb + (ok ? k : 0));   // clamped, branch-free
                w[r][q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    };
    load_w(n_begin);                                   // the HBM stream starts before the (L2-resident) A rows are staged
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int q = 0; q < KPT4; ++q) {
            const int k = k0 + q * 4;
            *reinterpret_cast<float4*>(As + (size_t)m * KS + k) = k < K ? ld4(A + (int64_t)m * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    // each thread only ever reads back the A slice it wrote itself: no barrier needed for As

    for (int n = n_begin; n < n_end; n += RB) {
        float acc[RB][MMAX];
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            if (m < M) {                               // M is uniform: no divergence
                float s[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) s[r] = 0.f;
#pragma unroll
                for (int q = 0; q < KPT4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(As + (size_t)m * KS + k0 + q * 4);
#pragma unroll
                    for (int r = 0; r < RB; ++r) s[r] += a.x * w[r][q].x + a.y * w[r][q].y + a.z * w[r][q].z + a.w * w[r][q].w;
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r][m] = s[r];
            } else {
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r][m] = 0.f;
            }
        }
        if (n + RB < n_end) load_w(n + RB);            // next W rows in flight while this block is reduced
#pragma unroll
        for (int m = 0; m < MMAX; ++m)
            if (m < M)
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r][m] = wave_sum(acc[r][m]);
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int m = 0; m < MMAX; ++m) red[(wave * RB + r) * MMAX + m] = acc[r][m];
        __syncthreads();
        if (t < RB * MMAX) {
            const int r = t / MMAX, m = t % MMAX;
            if (m < M && n + r < n_end) {
                float o = bias ? bias[n + r] : 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) o += red[(ww * RB + r) * MMAX + m];
                if (relu) o = fmaxf(o, 0.f);
                C[(int64_t)m * ldc + n + r] = o;
            }
        }
    }
}

template <int THREADS, int KPT4, int RB>
int launch(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K,
           int relu, hipStream_t s) {
    constexpr int KS = THREADS * KPT4 * 4;
    const size_t lds = sizeof(float) * ((size_t)M * KS + (THREADS / 64) * RB * 16);
    if (lds > 150 * 1024) return -100;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)gemm_skinny_nt_kernel<THREADS, KPT4, RB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) {
            subgc::set_error("gemm(skinny): cannot raise the dynamic LDS limit");
            return SUBGC_ELAUNCH;
        }
        attr_set = true;
    }
    // one workgroup per CU streams a contiguous block of W rows (a multiple of RB, at least 2*RB so the prefetch overlaps)
    int rows_per_wg = (int)subgc::cdiv(N, 256);
    rows_per_wg = std::max(2 * RB, (rows_per_wg + RB - 1) / RB * RB);
    const int wgs = (int)subgc::cdiv(N, rows_per_wg);
    hipLaunchKernelGGL((gemm_skinny_nt_kernel<THREADS, KPT4, RB>), dim3(wgs), dim3(THREADS), lds, s, A, lda, W, ldb, C, ldc, bias, M, N, K,
                       relu, rows_per_wg);
    return subgc::check_launch("subgc_gemm_f32(skinny)");
}

}  // namespace

namespace subgc {

int gemm_skinny_nt(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                   int relu, hipStream_t s) {
    if (M < 1 || M > 16) return -100;
    const int k4 = (K + 3) / 4;                       // float4 per row
    if (k4 <= 256 * 1) return launch<256, 1, 8>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 256 * 2) return launch<256, 2, 4>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 512 * 2) return launch<512, 2, 4>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 1024 * 2) return launch<1024, 2, 2>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    return -100;
}

}  // namespace subgc
