// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fmaf chains) with the
// fused epilogues the Sub-GC path needs: bias, residual add, ReLU, dropout keep-mask, accumulate,
// gathered A rows, scattered C rows and a device-side row count (ragged row sets).
//
// Replaces every nn.Linear / nn.LSTMCell contraction of the path and their backward
// (reference: AttModel.py:363-366,376-377,386,411-413,421-423,336-340,453;
//  graph_conv_unit.py:29-30; gpn.py:54,79).
//
// Shape of one workgroup: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN x BK(32);
// each wave owns a (BM/2)x(BN/2) sub-tile as MT x NT MFMA tiles of 32x32 (16 accumulator
// VGPRs each).  Operands are staged HBM -> registers -> LDS (two LDS stages, one barrier per
// K-tile; the next tile's global loads are in flight while the current tile is multiplied).
//
// LDS images, chosen per operand by how it lies in memory so that no transposition is needed:
//   K-contiguous operand (A not transposed / B = nn.Linear weight [N,K]):  T[row][BK+4]
//       a lane reads ONE ds_read_b128 = 4 consecutive k of its row and feeds 4 MFMAs;
//       row stride 36 floats = 9 x 16 B slots -> the 16-lane groups of ds_read_b128 hit 16
//       distinct slots (conflict-free).
//   K-major operand (A transposed [K,M] / B [K,N]):  T[k][BR+4]
//       a lane reads ds_read_b32 at [k][row0 + lane&31]: 32 consecutive floats per half-wave.
// MFMA k-assignment inside a chunk of 8 k: lanes 0-31 take k = 0..3, lanes 32-63 k = 4..7, one
// per MFMA -- A and B use the same assignment, so the sum over k is complete (order differs
// from a sequential loop, which fp32 tolerates: results are within rounding, not bit-equal).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* add; const uint8_t* keep;
    const int32_t* a_rows; const int32_t* c_rows; const int32_t* m_dev;
    int64_t lda, ldb, ldc, ldadd;
    int M, N, K, flags;
    float keep_scale;
};

constexpr int BK = 32;
constexpr int KPAD = 4;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Load 4 consecutive elements starting at logical index `i0` of a line of `n` elements.
template <bool VEC>
__device__ __forceinline__ float4 load4_guard(const float* line, int i0, int n) {
    if (VEC) {
        if (i0 < n) return ld4(line + i0);   // n % 4 == 0 and i0 % 4 == 0: all in or all out
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 r;
    r.x = (i0 + 0 < n) ? line[i0 + 0] : 0.f;
    r.y = (i0 + 1 < n) ? line[i0 + 1] : 0.f;
    r.z = (i0 + 2 < n) ? line[i0 + 2] : 0.f;
    r.w = (i0 + 3 < n) ? line[i0 + 3] : 0.f;
    return r;
}

// ---- staging of one operand tile --------------------------------------------------------
// KC = K-contiguous image: ROWS x BK, LDS [ROWS][BK+KPAD]; per thread ROWS*BK/4/256 float4.
// KM = K-major image:      BK x ROWS, LDS [BK][ROWS+KPAD].
template <int ROWS, bool KMAJOR>
struct Stage {
    static constexpr int NV = ROWS * BK / 4 / 256;   // float4 per thread
    static constexpr int LDS_FLOATS = KMAJOR ? BK * (ROWS + KPAD) : ROWS * (BK + KPAD);
    float4 r[NV];

    // src: operand base; ld: leading dim; row0: first tile row; k0: first k;
    // nrows / K: logical extents; rows_idx: optional gather (K-contiguous only)
    template <bool VEC>
    __device__ __forceinline__ void load(const float* src, int64_t ld, int row0, int k0, int nrows, int K,
                                         const int32_t* rows_idx) {
        const int t = threadIdx.x;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;           // float4 id inside the tile
            if (!KMAJOR) {
                const int rr = f / (BK / 4), c4 = f % (BK / 4);
                const int row = row0 + rr;
                int64_t srow = row;
                bool ok = row < nrows;
                if (rows_idx != nullptr && ok) {
                    const int g = rows_idx[row];
                    ok = g >= 0;
                    srow = g;
                }
                r[v] = ok ? load4_guard<VEC>(src + srow * ld, k0 + c4 * 4, K) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int kk = f / (ROWS / 4), c4 = f % (ROWS / 4);
                const int k = k0 + kk;
                r[v] = (k < K) ? load4_guard<VEC>(src + (int64_t)k * ld, row0 + c4 * 4, nrows)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    __device__ __forceinline__ void store(float* lds) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;
            if (!KMAJOR) {
                const int rr = f / (BK / 4), c4 = f % (BK / 4);
                *reinterpret_cast<float4*>(lds + rr * (BK + KPAD) + c4 * 4) = r[v];
            } else {
                const int kk = f / (ROWS / 4), c4 = f % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + kk * (ROWS + KPAD) + c4 * 4) = r[v];
            }
        }
    }
};

// fragment fetch for chunk c (8 k) of a 32-row MFMA tile starting at tile row `r0`
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void frag(const float* lds, int r0, int c, int lane, float (&out)[4]) {
    const int i = lane & 31, h = lane >> 5;
    if (!KMAJOR) {
        const float4 q = *reinterpret_cast<const float4*>(lds + (r0 + i) * (BK + KPAD) + c * 8 + h * 4);
        out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
    } else {
        const float* p = lds + (c * 8 + h * 4) * (ROWS + KPAD) + r0 + i;
#pragma unroll
        for (int s = 0; s < 4; ++s) out[s] = p[s * (ROWS + KPAD)];
    }
}

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    constexpr bool A_KM = TA;        // A stored [K,M]  -> K-major image
    constexpr bool B_KM = !TB;       // B stored [K,N]  -> K-major image
    using SA = Stage<BM, A_KM>;
    using SB = Stage<BN, B_KM>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const lA0 = smem;
    float* const lB0 = smem + 2 * SA::LDS_FLOATS;

    // m_dev bounds the ROWS of the stored A: M when A is [M,K], K when A is stored transposed [K,M]
    const int M = (p.m_dev && !TA) ? min(p.M, *p.m_dev) : p.M;
    const int K = (p.m_dev && TA) ? min(p.K, *p.m_dev) : p.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    SA sa; SB sb;
    const int nk = (K + BK - 1) / BK;
    sa.template load<VEC>(p.A, p.lda, m0, 0, M, K, TA ? nullptr : p.a_rows);
    sb.template load<VEC>(p.B, p.ldb, n0, 0, p.N, K, nullptr);
    sa.store(lA0); sb.store(lB0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float* lAc = lA0 + cur * SA::LDS_FLOATS;
        const float* lBc = lB0 + cur * SB::LDS_FLOATS;
        if (kt + 1 < nk) {
            sa.template load<VEC>(p.A, p.lda, m0, (kt + 1) * BK, M, K, TA ? nullptr : p.a_rows);
            sb.template load<VEC>(p.B, p.ldb, n0, (kt + 1) * BK, p.N, K, nullptr);
        }
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float fa[MT][4], fb[NT][4];
#pragma unroll
            for (int a = 0; a < MT; ++a) frag<BM, A_KM>(lAc, wm + a * 32, c, lane, fa[a]);
#pragma unroll
            for (int b = 0; b < NT; ++b) frag<BN, B_KM>(lBc, wn + b * 32, c, lane, fb[b]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(lA0 + (cur ^ 1) * SA::LDS_FLOATS); sb.store(lB0 + (cur ^ 1) * SB::LDS_FLOATS);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_l = lane & 31, hrow = 4 * (lane >> 5);
    const bool relu = p.flags & SUBGC_GEMM_RELU, accum = p.flags & SUBGC_GEMM_ACCUM;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int col = n0 + wn + b * 32 + col_l;
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                if (m >= M) continue;
                int64_t row = m;
                if (p.c_rows) {
                    const int g = p.c_rows[m];
                    if (g < 0) continue;
                    row = g;
                }
                float v = acc[a][b][r] + bias;
                if (p.add) v += p.add[row * p.ldadd + col];
                if (relu) v = fmaxf(v, 0.f);
                if (p.keep) v *= p.keep[row * p.ldc + col] ? p.keep_scale : 0.f;
                float* dst = p.C + row * p.ldc + col;
                if (accum) v += *dst;
                *dst = v;
            }
        }
    }
}

template <int BM, int BN, bool TA, bool TB, bool VEC>
int launch(const GemmArgs& a, hipStream_t s) {
    using SA = Stage<BM, TA>;
    using SB = Stage<BN, !TB>;
    const size_t lds = sizeof(float) * 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
    dim3 grid((unsigned)subgc::cdiv(a.N, BN), (unsigned)subgc::cdiv(a.M, BM));
    static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)gemm_f32_kernel<BM, BN, TA, TB, VEC>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            subgc::set_error("gemm: cannot raise dynamic LDS limit to %zu", lds);
            return SUBGC_ELAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, VEC>), grid, dim3(256), lds, s, a);
    return subgc::check_launch("subgc_gemm_f32");
}

template <bool TA, bool TB, bool VEC>
int pick_tile(const GemmArgs& a, hipStream_t s) {
    // 256 CUs: prefer the big tile only when it still yields >= ~1.5 workgroups per CU
    const int64_t big = subgc::cdiv(a.M, 128) * subgc::cdiv(a.N, 128);
    if (big >= 384) return launch<128, 128, TA, TB, VEC>(a, s);
    return launch<64, 64, TA, TB, VEC>(a, s);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

SUBGC_API int subgc_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                             const float* add, int64_t ldadd, const uint8_t* keep, float keep_scale, int flags,
                             const int32_t* a_rows, const int32_t* c_rows, const int32_t* m_dev, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size M=%d N=%d K=%d", M, N, K);
    if (M == 0 || N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(A && B && C, "gemm: null operand");
    SUBGC_REQUIRE(!(transA && transB), "gemm: transA && transB not supported");
    SUBGC_REQUIRE(!(transA && a_rows), "gemm: a_rows needs transA == 0");
    SUBGC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "gemm: leading dimension too small");
    SUBGC_REQUIRE(!add || ldadd >= N, "gemm: ldadd too small");
    GemmArgs a{A, B, C, bias, add, keep, a_rows, c_rows, m_dev, lda, ldb, ldc, ldadd, M, N, K, flags, keep_scale};
    hipStream_t s = (hipStream_t)stream;
    // vector path: every staged line is read as aligned float4 and is all-in or all-out of range
    const bool vecA = aligned16(A) && lda % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
    const bool vecB = aligned16(B) && ldb % 4 == 0 && (transB ? K % 4 == 0 && !(transA && m_dev) : N % 4 == 0);
    const bool vec = vecA && vecB;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    if (!transA && transB) return vec ? pick_tile<false, true, true>(a, s) : pick_tile<false, true, false>(a, s);
    if (!transA && !transB) return vec ? pick_tile<false, false, true>(a, s) : pick_tile<false, false, false>(a, s);
    return vec ? pick_tile<true, false, true>(a, s) : pick_tile<true, false, false>(a, s);
}

// ---- column sums (bias gradients) -------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                     float* __restrict__ out, int accumulate,
                                                     const int32_t* m_dev, int rows_per_block) {
    // block (bx, by): 64 columns x a slab of rows; 4 waves stride the slab, LDS-reduce, one atomic per column
    __shared__ float sm[4][64];
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc = 0.f;
    if (col < N)
        for (int r = r0 + w; r < r1; r += 4) acc += X[(int64_t)r * ldx + col];
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < N) {
        const float v = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
        atomicAdd(out + col, v);
    }
}
__global__ void zero_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
}  // namespace

SUBGC_API int subgc_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out, int accumulate,
                               const int32_t* m_dev, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && ldx >= N, "colsum: bad sizes");
    if (N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && out, "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) hipLaunchKernelGGL(zero_kernel, dim3((N + 255) / 256), dim3(256), 0, s, out, N);
    if (M == 0) return subgc::check_launch("subgc_colsum_f32");
    const int rows_per_block = 256;
    dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, X, ldx, M, N, out, accumulate, m_dev, rows_per_block);
    return subgc::check_launch("subgc_colsum_f32");
}
