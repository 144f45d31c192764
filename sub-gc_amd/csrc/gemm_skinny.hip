// Skinny (weight-streaming) GEMM for decode: C[M,N] = act(A[M,K] W[N,K]^T + bias), M <= 16.
//
// One caption = one row, so a decode step multiplies <= 16 activation rows by 152 MB of fp32 weights:
// the contraction is HBM-bound (reference: the per-step nn.Linear / nn.LSTMCell calls of
// AttModel.py:332-340,411-423,453 at batch = kept sub-graphs of ONE image).  MFMA tiles would waste
// >= 84 % of their rows, so this kernel runs on the VALU at streaming rate instead:
// (see the kernel comment below for the data layout).
#include "common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Kernel shape (second version).  The first version spread K over the 256 threads of a workgroup: every W row then
// needed M x (6-step wave reduction) in EVERY wave plus a workgroup combine through LDS and two barriers -- at M = 10 the
// reductions, not the weight stream, set the time (50 us for the 38 MB logit matrix).  Now a WAVE owns whole W rows:
//   * lane l holds the k-slices {(q*64 + l)*4 .. +3} of a row (KPT4 coalesced float4 loads = 1 KB per wave instruction);
//   * the A rows (M x K fp32) are staged once per workgroup in LDS; a lane reads its slice with conflict-free ds_read_b128
//     and reuses it for the RB rows it has in flight;
//   * one 6-step wave reduction per (row, m), no barrier, no cross-wave traffic; lane r*16+m stores C[m, n+r].
template <int KPT4, int RB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_nt_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                                    int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                                    const float* __restrict__ bias, int M, int N, int K, int relu) {
    constexpr int MMAX = 16, KS = 64 * KPT4 * 4;
    extern __shared__ __attribute__((aligned(16))) float As[];              // [M][KS]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < M * (KS / 4); i += WAVES * 64) {
        const int m = i / (KS / 4), k = (i % (KS / 4)) * 4;
        *reinterpret_cast<float4*>(As + (size_t)m * KS + k) = k < K ? ld4(A + (int64_t)m * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int n0 = (blockIdx.x * WAVES + wave) * RB;
    if (n0 >= N) return;
    float4 w[RB][KPT4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < KPT4; ++q) {
            const int k = (q * 64 + lane) * 4;
            const bool ok = n0 + r < N && k < K;
            const float4 v = ld4(W + (int64_t)(ok ? n0 + r : n0) * ldb + (ok ? k : 0));           // clamped, branch-free
            w[r][q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float mine = 0.f;                                                        // lane r*16+m ends up owning C[m, n0+r]
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {                                                         // M is uniform: no divergence
            float s[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) s[r] = 0.f;
#pragma unroll
            for (int q = 0; q < KPT4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(As + (size_t)m * KS + (q * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < RB; ++r) s[r] += a.x * w[r][q].x + a.y * w[r][q].y + a.z * w[r][q].z + a.w * w[r][q].w;
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const float tot = wave_sum(s[r]);
                if (lane == r * MMAX + m) mine = tot;
            }
        }
    }
    const int r = lane / MMAX, m = lane % MMAX;
    if (r < RB && m < M && n0 + r < N) {
        float o = mine + (bias ? bias[n0 + r] : 0.f);
        if (relu) o = fmaxf(o, 0.f);
        C[(int64_t)m * ldc + n0 + r] = o;
    }
}

template <int KPT4, int RB, int WAVES>
int launch(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K,
           int relu, hipStream_t s) {
    constexpr int KS = 64 * KPT4 * 4;
    const size_t lds = sizeof(float) * (size_t)M * KS;
    if (lds > 150 * 1024) return -100;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)gemm_skinny_nt_kernel<KPT4, RB, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
            hipSuccess) {
            subgc::set_error("gemm(skinny): cannot raise the dynamic LDS limit");
            return SUBGC_ELAUNCH;
        }
        attr_set = true;
    }
    const int wgs = (int)subgc::cdiv(N, (int64_t)RB * WAVES);
    hipLaunchKernelGGL((gemm_skinny_nt_kernel<KPT4, RB, WAVES>), dim3(wgs), dim3(WAVES * 64), lds, s, A, lda, W, ldb, C, ldc, bias, M, N, K,
                       relu);
    return subgc::check_launch("subgc_gemm_f32(skinny)");
}

// rows per wave: as many as keep >= ~1500 waves in flight (RB x 16 <= 64 lanes own the results: RB <= 4)
template <int KPT4>
int pick_rb(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K, int relu,
            hipStream_t s) {
    if (KPT4 <= 8 && N >= 4 * 1500) return launch<KPT4, 4, 8>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (N >= 2 * 1500) return launch<KPT4, 2, 8>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    return launch<KPT4, 1, 8>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
}

}  // namespace

namespace subgc {

int gemm_skinny_nt(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                   int relu, hipStream_t s) {
    if (M < 1 || M > 16) return -100;
    const int k4 = (K + 3) / 4;                       // float4 per row; a wave covers 64 of them per KPT4 step
    if (k4 <= 64 * 4) return pick_rb<4>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 64 * 8) return pick_rb<8>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 64 * 12) return pick_rb<12>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    if (k4 <= 64 * 16) return pick_rb<16>(A, lda, W, ldb, C, ldc, bias, M, N, K, relu, s);
    return -100;
}

}  // namespace subgc
