// Shared host-side plumbing of libsubgc_hip.so: error codes, thread-local error text,
// launch checking and the HIP-event profiling hook declared in include/subgc_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/subgc_hip.h"

#define SUBGC_API extern "C" __attribute__((visibility("default")))

namespace subgc {

void set_error(const char* fmt, ...);

// RAII bracket: records a hipEvent pair around a launch when profiling of `family` is on.
struct ProfScope {
    ProfScope(int family, hipStream_t s, double work, double moved = -1.0);      // moved: bytes the launch really moves (default: = work)
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SUBGC_ELAUNCH;
    }
    return SUBGC_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Debug bounds mode (subgc_debug_bounds(1); off by default): every entry point that consumes an index tensor it did not produce -- rel_ind,
// gpn_obj_ind, token / label / target ids, class ids -- validates it BEFORE its kernels run and returns SUBGC_EINVAL naming the tensor, the
// first bad position and its value, the way the reference fails on a bad loader tensor (an IndexError from torch, the asserts of
// gpn.py:117-118) instead of clamping or reading out of range.  One extra launch + a stream synchronisation per checked tensor: a
// debugging aid, never on in the timed paths; inside a stream capture the checks are skipped (a synchronisation is illegal there).
bool debug_bounds();
// every x[r * ld + c], r < rows, c < cols (elem = 4: int32, 8: int64) must lie in [lo, hi] or equal `also_ok` (pass lo - 1 for "nothing else")
int debug_check_range(const void* x, int elem, int64_t rows, int64_t cols, int64_t ld, int64_t lo, int64_t hi, int64_t also_ok, const char* what,
                      hipStream_t s);
// gpn.py:117-118: a sub-graph's node list holds the dummy node (N - 1) exactly where its attention mask is zero
int debug_check_mask_agrees(const int64_t* obj_ind, const float* mask, int64_t n, int64_t dummy, const char* what, hipStream_t s);
// bytes an LSTM cell forward launch over n = rows x R hidden units moves: `parts` pre-activation planes + the additive gate terms (4 floats
// per unit each), c_prev, c, the h destinations (fp32 or bf16) and the saved gates -- against the 12 floats per unit of the algorithmic count
inline double lstm_fwd_moved_bytes(int64_t n, int parts, bool g1, bool g2, bool c_prev, bool h2, bool hdrop, bool gates, int h_bf16) {
    const double hb = h_bf16 ? 2.0 : 4.0;
    return (double)n * (4.0 * 4 * ((parts < 1 ? 1 : parts) + (g1 ? 1 : 0) + (g2 ? 1 : 0)) + (c_prev ? 4.0 : 0.0) + 4.0 + hb * (1 + (h2 ? 1 : 0) + (hdrop ? 1 : 0)) +
                       (gates ? 16.0 : 0.0));
}
// Dynamic LDS above 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize raised once per kernel AND device; the largest size granted
// is remembered per (kernel, device), so a hot launch path pays one table look-up instead of a driver call (the GEMM files keep their own
// per-instantiation bit masks).  > 160 KiB (the gfx950 LDS) is SUBGC_EINVAL with the kernel's name in the error text.
int raise_lds_cached(const void* kernel, size_t bytes, const char* what);
// gridDim.z of the per-column GCN kernels: their node / relation loop is strided over z, and a workgroup's loop is a chain of gather
// latencies -- enough slices for ~2048 workgroups (4 slices left 9-25 nodes per workgroup: 2.3-3.6 TB/s)
inline int gcn_zsplit(int col_groups, int B, int items) {
    const int wg = col_groups * B > 0 ? col_groups * B : 1;
    int z = (2048 + wg - 1) / wg;
    if (z < 4) z = 4;
    if (z > items) z = items;
    return z < 1 ? 1 : z;
}


// gemm_skinny.hip: C[M,N] = act(A[M,K] W[N,K]^T + bias) for M <= 16 (weight-streaming bound); returns -100
// when the shape is not covered (caller falls back to the tiled MFMA kernels).
int gemm_nt_partials(const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int K, int gemm_flags, float* ws, size_t ws_bytes,
                     hipStream_t s, int* splits);
// a weight / bias gradient over NO rows: what is not accumulated into becomes zero (gemm_f32.hip; shared by both wgrad entry points)
int wgrad_no_rows(float* dW, int64_t lddw, float* db, int M, int N, bool accum_w, bool accum_b, hipStream_t s);
int gemm_bf16_nt_partials(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, float* ws, size_t ws_bytes,
                          hipStream_t s, int* splits);
int gemm_skinny_nt(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N,
                   int K, int relu, hipStream_t stream, const float* add = nullptr, int64_t ldadd = 0);

// The attention query of a row as the forward kernels take it: `ah` + n_planes - 1 further planes `stride` floats apart (the split-K
// partial planes of the h2att product, summed on load) + bias; the summed row is written to `out` (the backward reads it).  The
// default = one plane, no bias, nothing written: `ah` IS the query.
struct QSrc { const float* bias; float* out; int n_planes; int64_t stride; };
// attention_vec.hip: float4 forms of the per-step attention kernels; return -100 when they do not apply
int attn_fwd_vec(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* off,
                 const int32_t* len, void* ctx, int64_t ldctx, float* alpha, int n_stride, int S, int A, int R, int ctx_b16, int uv_b16,
                 hipStream_t s, QSrc qs = QSrc{nullptr, nullptr, 1, 0});
int attn_bwd_vec(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                 const float* alpha, int n_stride, const float* dctx, int64_t lddctx, void* dah, float* du, float* dv, float* dw_a,
                 float* db_a, int S, int A, int R, int dah_b16, int uv_b16, float* dctx_keep, int64_t ldkeep, hipStream_t s, int n_planes = 1,
                 int64_t plane_stride = 0, float* de_keep = nullptr);
int attn_du_accum_vec(const void* u, int uv_b16, const float* ah, const float* de, int n_stride, const int32_t* step_off, int T, const int32_t* off,
                      const int32_t* len, const float* w_a, float* du, int S, int A, hipStream_t s);
int attn_dv_accum_vec(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off, int T, const int32_t* off,
                      const int32_t* len, float* dv, int S, int R, hipStream_t s);

}  // namespace subgc

// query chunk (four columns at `idx` = row * A + column) of a QSrc: planes summed in order, bias last
__device__ __forceinline__ float4 subgc_load_q(const float* __restrict__ ah, const subgc::QSrc& qs, int64_t idx, int col) {
    // the planes are REQUESTED together (a runtime-length loop of load + add is one memory latency per plane at the head of every
    // attention launch: 4-8 planes on the 512-wide query product); planes past the count re-read the last one and are not added, the
    // additions keep the plane order
    constexpr int MAXQ = 8;
    float4 x[MAXQ];
#pragma unroll
    for (int p = 0; p < MAXQ; ++p) x[p] = *reinterpret_cast<const float4*>(ah + (p < qs.n_planes ? p : qs.n_planes - 1) * qs.stride + idx);
    float4 a = x[0];
#pragma unroll
    for (int p = 1; p < MAXQ; ++p)
        if (p < qs.n_planes) { a.x += x[p].x; a.y += x[p].y; a.z += x[p].z; a.w += x[p].w; }
    for (int p = MAXQ; p < qs.n_planes; ++p) {
        const float4 b = *reinterpret_cast<const float4*>(ah + p * qs.stride + idx);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (qs.bias) {
        const float4 b = *reinterpret_cast<const float4*>(qs.bias + col);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    return a;
}

#define SUBGC_DEBUG_RANGE(x, elem, rows, cols, ld, lo, hi, also_ok, what, s)                                                       \
    do {                                                                                                                        \
        if (subgc::debug_bounds())                                                                                             \
            if (int rc_ = subgc::debug_check_range(x, elem, rows, cols, ld, lo, hi, also_ok, what, (hipStream_t)(s))) return rc_; \
    } while (0)

#define SUBGC_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            subgc::set_error(__VA_ARGS__); \
            return SUBGC_EINVAL;          \
        }                                 \
    } while (0)

// ---- device helpers ------------------------------------------------------------------
// Wave-wide reductions on the DPP path (gfx9 family: quad_perm / row_half_mirror / row_mirror inside a 16-lane row, row_bcast:15 / :31
// across rows, the total read from lane 63): six VALU moves of a few cycles each.  The __shfl_xor butterfly they replace compiles to six
// DEPENDENT ds_bpermute exchanges through the LDS crossbar -- ~1 us per lone reduction when nothing else is ready to issue.
// PRECONDITION (unlike the butterfly, which gave every ACTIVE lane the sum over the active lanes): ALL 64 LANES ACTIVE.  The total is
// read from lane 63 and the row_bcast steps pull from lanes 15 / 31 / 47: in a partial wave (blockDim not a multiple of 64, a call under
// a lane-dependent branch or after an early return of some lanes) those registers are stale and every lane receives garbage.  Callers
// keep their reductions wave-uniform (inactive work contributes the identity: 0 for sums, -inf / the value itself for maxima); builds with
// -DSUBGC_DEBUG_EXEC trap when the rule is broken.  gfx9-family DPP controls only (this library targets gfx950 alone).
#ifdef SUBGC_DEBUG_EXEC
#define SUBGC_ASSERT_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define SUBGC_ASSERT_FULL_WAVE() do { } while (0)
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float v, float other) {      // lane <- DPP source lane of v; `other` where the row is masked off
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, other), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    SUBGC_ASSERT_FULL_WAVE();
    v += dpp_or<0xB1, 0xF>(v, 0.f);      // quad_perm [1,0,3,2]
    v += dpp_or<0x4E, 0xF>(v, 0.f);      // quad_perm [2,3,0,1]
    v += dpp_or<0x141, 0xF>(v, 0.f);     // row_half_mirror: every lane of an 8-group holds the group's sum
    v += dpp_or<0x140, 0xF>(v, 0.f);     // row_mirror: ... of a 16-lane row, the row's sum
    v += dpp_or<0x142, 0xA>(v, 0.f);     // row_bcast:15 into rows 1 and 3
    v += dpp_or<0x143, 0xC>(v, 0.f);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over the 64 lanes of N values at once: the N chains interleave
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
    SUBGC_ASSERT_FULL_WAVE();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0xB1, 0xF>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0x4E, 0xF>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0x141, 0xF>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0x140, 0xF>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0x142, 0xA>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += dpp_or<0x143, 0xC>(v[k], 0.f);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
}
// Segment sums of two gather lists at once: sa += fa(ia[j]) for j in [a0, a1), sb += fb(ib[j]) for j in [b0, b1).  Up to four rows of EACH list
// are requested before any is added (wave-uniform bounds: the skipped loads are scalar branches), additions in list order, so the result is
// bit-identical to the plain loops -- which waited for every row before asking for the next (degree 2-3 per node: a chain of memory latencies).
template <class FA, class FB>
__device__ __forceinline__ void gather_pair(const int* __restrict__ ia, int a0, int a1, FA fa, const int* __restrict__ ib, int b0, int b1, FB fb,
                                            float4& sa, float4& sb) {
    constexpr int U = 4;
    for (int ja = a0, jb = b0; ja < a1 || jb < b1; ja += U, jb += U) {
        float4 xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ja + u < a1) xa[u] = fa(ia[ja + u]);
            if (jb + u < b1) xb[u] = fb(ib[jb + u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ja + u < a1) { sa.x += xa[u].x; sa.y += xa[u].y; sa.z += xa[u].z; sa.w += xa[u].w; }
            if (jb + u < b1) { sb.x += xb[u].x; sb.y += xb[u].y; sb.z += xb[u].z; sb.w += xb[u].w; }
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
    SUBGC_ASSERT_FULL_WAVE();
    v = fmaxf(v, dpp_or<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_or<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_or<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// block-wide sum through LDS scratch (>= 16 floats); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sm[i]);
    return r;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tanh for the attention scores (AttModel.py:459, S x n x att_hid_size of them per decoder step, forward and backward: the kernels
// are VALU-bound on it): 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units -- 6 instructions instead of libm's ~40; absolute
// error <= 5e-7 (d tanh = 2e/(e+1)^2 * rel_err(e) <= 0.5 * 1e-6), exact limits +-1, NaN propagates
__device__ __forceinline__ float subgc_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, e + 1.f);
}
