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
#include "bf16_util.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* add; const uint8_t* keep;
    const int32_t* a_rows; const int32_t* c_rows; const int32_t* m_dev;
    int64_t lda, ldb, ldc, ldadd;
    int M, N, K, flags;
    float keep_scale;
    float* ws; size_t ws_bytes;      // the caller's split-K scratch for THIS call (host-side use only)
    int xmode;                       // arithmetic of the 128x128-tile forms: 0 fp32 pipe, 1 bf16x3 split, 2 bf16 operands
    int no_splitk;                   // SUBGC_GEMM_NO_SPLITK of this call (measurement scripts)
    int planes_only = 0;             // subgc_gemm_f32_planes: split-K forms leave their partial planes in `ws` (no reduce pass)
    int* splits_out = nullptr;       // ... and report how many (1 = the plain kernel wrote C)
    // subgc_gemm_f32_wgrad (transposed-A forms only): column sums of the stored A -- sum_k A[k][m], the bias gradient next to the
    // weight gradient dY^T x -- taken from the staging registers of the workgroups of tile column 0
    float* cs_out = nullptr;         // [M] destination (cs_accum: added to)
    float* cs_part = nullptr;        // [splits][M] partial sums of the split-K forms (tail of the caller's workspace)
    int cs_accum = 0;
    // subgc_gemm_f32_pair: a SECOND problem of the same shape, layout and epilogue in the same launch (the second half of the grid)
    const float* A2 = nullptr; const float* B2 = nullptr; float* C2 = nullptr; const float* bias2 = nullptr;
    int nprob = 1;
};

// Pair launches (see gemm_bf16.hip): workgroups [0, nwg/2) work on problem 0, [nwg/2, nwg) on problem 1
__device__ __forceinline__ int select_problem(GemmArgs& q, int& b, int& nwg) {
    if (q.nprob != 2) return 0;
    nwg >>= 1;
    if (b < nwg) return 0;
    b -= nwg;
    q.A = q.A2; q.B = q.B2; q.C = q.C2; q.bias = q.bias2;
    return 1;
}

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
// K-contiguous image: ROWS x BK, LDS [ROWS][BK+KPAD]; per thread NV = ROWS*BK/4/256 float4.
// K-major image:      BK x ROWS, LDS [BK][ROWS+KPAD].
// Everything that does not depend on k (row pointers, row validity) is computed ONCE by init();
// a per-tile load is then one 64-bit add + one global_load_dwordx4 per float4, so the VALU work
// between MFMA groups stays small.  Out-of-range elements are loaded from a clamped (valid)
// address and zeroed when the registers are written to LDS, so loads never wait early.
template <int ROWS, bool KMAJOR>
struct Stage {
    static constexpr int NV = ROWS * BK / 4 / 256;   // float4 per thread
    static constexpr int LDS_FLOATS = KMAJOR ? BK * (ROWS + KPAD) : ROWS * (BK + KPAD);
    float4 r[NV];
    const float* base[NV];     // element (row, k = 0) [K-contiguous]  /  (k = kk_local, col) [K-major]
    unsigned rowok;            // bit v: the row / column group of r[v] is inside the matrix
    unsigned okmask;           // rowok & (k in range), decided per tile
    int64_t ld_;
    int extent;                // nrows (logical rows of this operand)
    float4 cs;                 // K-major A only: running sum over k of this thread's four columns (every r[v] holds the same four)

    // add the tile in the staging registers to `cs`; MASKED as store() masks, raw as store_raw() does not (steady state: every k is
    // in range and a column beyond the matrix only feeds a sum that is never stored)
    template <bool MASKED>
    __device__ __forceinline__ void colsum_add() {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool on = !MASKED || ((okmask >> v) & 1u);
            cs.x += on ? r[v].x : 0.f; cs.y += on ? r[v].y : 0.f; cs.z += on ? r[v].z : 0.f; cs.w += on ? r[v].w : 0.f;
        }
    }

    template <bool VEC>
    __device__ __forceinline__ void init(const float* src, int64_t ld, int row0, int nrows, const int32_t* rows_idx) {
        const int t = threadIdx.x;
        rowok = 0; ld_ = ld; extent = nrows;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;
            if (!KMAJOR) {
                const int rr = f / (BK / 4), c4 = f % (BK / 4);
                const int row = row0 + rr;
                bool ok = row < nrows;
                int64_t srow = max(min(row, nrows - 1), 0);
                if (rows_idx != nullptr) {
                    const int g = rows_idx[srow];
                    ok = ok && g >= 0;
                    srow = max(g, 0);
                }
                base[v] = src + srow * ld + c4 * 4;
                if (ok) rowok |= 1u << v;
            } else {
                const int kk = f / (ROWS / 4), c4 = f % (ROWS / 4);
                const int col = row0 + c4 * 4;
                const bool ok = VEC ? col < nrows : true;      // scalar path guards per element
                base[v] = src + (int64_t)kk * ld + (VEC ? (col < nrows ? col : 0) : col);
                if (ok) rowok |= 1u << v;
            }
        }
    }

    template <bool VEC>
    __device__ __forceinline__ void load(int row0, int k0, int K) {
        const int t = threadIdx.x;
        if (VEC && k0 + BK <= K) {                   // interior tile (wave-uniform test): no k checks at all
            okmask = rowok;
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + (KMAJOR ? (int64_t)k0 * ld_ : (int64_t)k0));
            return;
        }
        okmask = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;
            const bool rok = (rowok >> v) & 1u;
            if (!KMAJOR) {
                const int kk = k0 + (f % (BK / 4)) * 4;
                if (VEC) {
                    r[v] = ld4(base[v] - (f % (BK / 4)) * 4 + (kk < K ? kk : 0));   // k out of range: element 0 of the row
                    if (rok && kk < K) okmask |= 1u << v;
                } else {
                    r[v] = rok ? load4_guard<false>(base[v] - (f % (BK / 4)) * 4, kk, K) : make_float4(0.f, 0.f, 0.f, 0.f);
                    okmask |= 1u << v;
                }
            } else {
                const int k = k0 + f / (ROWS / 4);
                if (VEC) {
                    r[v] = ld4(base[v] + (int64_t)(k < K ? k0 : -(f / (ROWS / 4))) * ld_);   // k out of range: row 0
                    if (rok && k < K) okmask |= 1u << v;
                } else {
                    const int col = row0 + (f % (ROWS / 4)) * 4;
                    r[v] = (k < K) ? load4_guard<false>(base[v] - col + (int64_t)k0 * ld_, col, extent) : make_float4(0.f, 0.f, 0.f, 0.f);
                    okmask |= 1u << v;
                }
            }
        }
    }

    // interior tile (all k in range), vector path only: branch-free
    __device__ __forceinline__ void load_interior(int k0) {
        okmask = rowok;
#pragma unroll
        for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + (KMAJOR ? (int64_t)k0 * ld_ : (int64_t)k0));
    }

    __device__ __forceinline__ void store_raw(float* lds) const {      // no masking (see mainloop steady state)
        const int t = threadIdx.x;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;
            if (!KMAJOR) *reinterpret_cast<float4*>(lds + (f / (BK / 4)) * (BK + KPAD) + (f % (BK / 4)) * 4) = r[v];
            else *reinterpret_cast<float4*>(lds + (f / (ROWS / 4)) * (ROWS + KPAD) + (f % (ROWS / 4)) * 4) = r[v];
        }
    }

    __device__ __forceinline__ void store(float* lds) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + v * 256;
            const float4 q = (okmask >> v) & 1u ? r[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!KMAJOR) {
                const int rr = f / (BK / 4), c4 = f % (BK / 4);
                *reinterpret_cast<float4*>(lds + rr * (BK + KPAD) + c4 * 4) = q;
            } else {
                const int kk = f / (ROWS / 4), c4 = f % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + kk * (ROWS + KPAD) + c4 * 4) = q;
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

#include "gemm_x3.h"
#include "gemm_x3_k16.h"

// ---- one (tile, k-range) unit of work: acc += A[m0.., kt0*BK .. kt1*BK) * B[.., n0..] --------------
// Software pipeline (one barrier per K-tile, no MFMA bubble around it):
//   * global loads run TWO tiles ahead: tile kt+2 is requested in the middle of iteration kt, right
//     after the staging registers were drained into the other LDS stage (so they have ~1.75
//     iterations of MFMA work to land);
//   * the staging registers of tile kt+1 are written to LDS[nxt] after the 2nd of the 4 k-chunks;
//   * MFMA fragments are double-buffered: chunk c+1 is fetched from LDS before chunk c is multiplied;
//   * the barrier sits BEFORE the last chunk's MFMAs and is immediately followed by the fetch of the
//     next tile's first fragments, so the matrix pipe has a full chunk of work queued while the
//     workgroup re-synchronises.
template <int BM, int BN, bool TA, bool TB, bool VEC, int MT, int NT, bool CS = false>
__device__ __forceinline__ void mainloop(const GemmArgs& p, float* smem, int M, int K, int m0, int n0, int kt0, int kt1,
                                         f32x16 (&acc)[MT][NT], float4* colsum = nullptr) {
    static_assert(!CS || TA, "column sums of A are taken from its K-major staging registers");
    constexpr int WM = BM / 2, WN = BN / 2, NC = BK / 8;
    constexpr bool A_KM = TA, B_KM = !TB;
    using SA = Stage<BM, A_KM>;
    using SB = Stage<BN, B_KM>;
    float* const lA0 = smem;
    float* const lB0 = smem + 2 * SA::LDS_FLOATS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int32_t* arows = TA ? nullptr : p.a_rows;
    if constexpr (CS) *colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kt1 <= kt0) return;
    SA sa; SB sb;
    sa.template init<VEC>(p.A, p.lda, m0, M, arows);
    sb.template init<VEC>(p.B, p.ldb, n0, p.N, nullptr);
    sa.template load<VEC>(m0, kt0 * BK, K);
    sb.template load<VEC>(n0, kt0 * BK, K);
    sa.store(lA0); sb.store(lB0);
    if constexpr (CS) { sa.cs = make_float4(0.f, 0.f, 0.f, 0.f); sa.template colsum_add<true>(); }
    if (kt0 + 1 < kt1) {
        sa.template load<VEC>(m0, (kt0 + 1) * BK, K);
        sb.template load<VEC>(n0, (kt0 + 1) * BK, K);
    }
    __syncthreads();
    float fa[2][MT][4], fb[2][NT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a) frag<BM, A_KM>(lA0, wm + a * 32, 0, lane, fa[0][a]);
#pragma unroll
    for (int b = 0; b < NT; ++b) frag<BN, B_KM>(lB0, wn + b * 32, 0, lane, fb[0][b]);

    // one K-tile; STEADY = tiles kt+1 and kt+2 exist and kt+2 is an interior tile: no branches, and the
    // staging / fragment traffic is interleaved one-by-one with the MFMAs (sched_group_barrier), so a
    // single wave keeps the matrix pipe busy by itself instead of alternating MFMA and memory phases.
    auto ktile = [&](int kt, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int cur = (kt - kt0) & 1;
        const float* lAc = lA0 + cur * SA::LDS_FLOATS;
        const float* lBc = lB0 + cur * SB::LDS_FLOATS;
        float* lAn = lA0 + (cur ^ 1) * SA::LDS_FLOATS;
        float* lBn = lB0 + (cur ^ 1) * SB::LDS_FLOATS;
        const bool has_next = STEADY || kt + 1 < kt1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int fc = c & 1, fn = fc ^ 1;
            if (c + 1 < NC) {                                    // fragments of the next chunk of this tile
#pragma unroll
                for (int a = 0; a < MT; ++a) frag<BM, A_KM>(lAc, wm + a * 32, c + 1, lane, fa[fn][a]);
#pragma unroll
                for (int b = 0; b < NT; ++b) frag<BN, B_KM>(lBc, wn + b * 32, c + 1, lane, fb[fn][b]);
            }
            if (c == 1 && has_next) {                            // drain staging regs -> other LDS stage, refill them
                if (STEADY) { sa.store_raw(lAn); sb.store_raw(lBn); } else { sa.store(lAn); sb.store(lBn); }
                if constexpr (CS) { if (STEADY) sa.template colsum_add<false>(); else sa.template colsum_add<true>(); }
                if (STEADY) {
                    sa.load_interior((kt + 2) * BK);
                    sb.load_interior((kt + 2) * BK);
                } else if (kt + 2 < kt1) {
                    sa.template load<VEC>(m0, (kt + 2) * BK, K);
                    sb.template load<VEC>(n0, (kt + 2) * BK, K);
                }
            }
            if (c == NC - 1) {                                   // re-synchronise, then prefetch the next tile's first chunk
                if (STEADY) {
#pragma unroll
                    for (int i = 0; i < (NC - 1) * 4 * MT * NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
                    }
                }
                __syncthreads();
                if (has_next) {
#pragma unroll
                    for (int a = 0; a < MT; ++a) frag<BM, A_KM>(lAn, wm + a * 32, 0, lane, fa[fn][a]);
#pragma unroll
                    for (int b = 0; b < NT; ++b) frag<BN, B_KM>(lBn, wn + b * 32, 0, lane, fb[fn][b]);
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[fc][b][s], fa[fc][a][s], acc[a][b], 0, 0, 0);   // operands swapped: C^T tile (epilogue)
        }
    };
    int kt = kt0;
    if (VEC && arows == nullptr) {
        // In the steady state nothing is masked: rows / columns beyond M / N are read from a clamped
        // (valid) address and only feed accumulator rows / columns the epilogue never stores.
        const int steady_end = min(kt1, K / BK) - 2;             // kt + 2 must be an interior tile of this unit
        for (; kt < steady_end; ++kt) ktile(kt, std::true_type{});
    }
    for (; kt < kt1; ++kt) ktile(kt, std::false_type{});
    __syncthreads();      // a following unit (stream-K) re-uses the LDS stages
    if constexpr (CS) *colsum = sa.cs;
}

// Column sums of the K-major A tile a workgroup staged (mainloop<.., CS = true>): thread t summed columns 4 (t % (BM/4)) .. +3 over the
// k-rows t / (BM/4) + 256/(BM/4) j of every tile; the 256/(BM/4) row groups meet in LDS (free after the main loop's last barrier) and
// thread m < BM stores column m0 + m.  Deterministic: fixed order inside a thread, fixed order over the groups.
template <int BM>
__device__ __forceinline__ void colsum_store(float* smem, const float4& cs, int m0, int M, float* dst, bool accum) {
    constexpr int G = 256 / (BM / 4);
    const int t = threadIdx.x;
    *reinterpret_cast<float4*>(smem + (t / (BM / 4)) * BM + (t % (BM / 4)) * 4) = cs;
    __syncthreads();
    if (t < BM && m0 + t < M) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) v += smem[g * BM + t];
        dst[m0 + t] = accum ? dst[m0 + t] + v : v;
    }
    __syncthreads();
}

// Epilogue.  Every main loop issues its MFMAs with the operands SWAPPED (B fragment first), so an accumulator tile holds C^T:
// lane l owns output ROW (l & 31) and register r the column (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the 32x32 tile -- four
// CONSECUTIVE columns per register quad = one 16-byte store (with the column-per-lane layout a lane issued 64 scalar stores per
// 128x128 tile and the store issue of a K = 1000 product rivalled its K loop).  f(m, n, quad) is called per whole / partial quad.
struct Quad { float v[4]; };
// WAVES_N: waves along N (the wave owns MT x NT tiles of 32 x 32 at row (wave / WAVES_N) * 32 MT, column (wave % WAVES_N) * 32 NT): 2 for the
// 2 x 2 waves of the 128 x 128 / 64 x 64 loops, 4 for the 2 x 4 waves of the eight-phase loop.  One instantiation per tile row `a` (a fold):
// the accumulator index stays a constant whatever the unroller decides.
template <int MT, int NT, int WAVES_N, int A, typename F>
__device__ __forceinline__ void quads_of_row(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, F& f) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 7;
    const int wm = (wave / WAVES_N) * (32 * MT), wn = (wave % WAVES_N) * (32 * NT);
    const int m = m0 + wm + A * 32 + (lane & 31);
    if (m >= M) return;
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn + b * 32 + 8 * q + 4 * (lane >> 5);
            if (n >= N) continue;
            Quad x{{acc[A][b][4 * q], acc[A][b][4 * q + 1], acc[A][b][4 * q + 2], acc[A][b][4 * q + 3]}};
            f(m, n, x);
        }
}
template <int MT, int NT, int WAVES_N, typename F, int... As>
__device__ __forceinline__ void for_each_quad_seq(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, F& f, std::integer_sequence<int, As...>) {
    (quads_of_row<MT, NT, WAVES_N, As>(acc, m0, n0, M, N, f), ...);
}
template <int BM, int BN, int MT, int NT, int WAVES_N = 2, typename F>
__device__ __forceinline__ void for_each_quad(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, F&& f) {
    static_assert(BM == (WAVES_N == 2 ? 2 : 2) * 32 * MT && BN == WAVES_N * 32 * NT, "two wave rows x WAVES_N wave columns cover the tile");
    for_each_quad_seq<MT, NT, WAVES_N>(acc, m0, n0, M, N, f, std::make_integer_sequence<int, MT>{});
}

// The same walk with the accumulators TRANSPOSED THROUGH LDS first (round 6; the bf16 kernels' for_each_quad_lds has the measurements): in the
// register layout one store instruction touches 32 rows and writes 32 bytes of each; here every wave writes a 32-row slab of its sub-tile
// into its own LDS region (row pitch + 4 floats), reads it back row-major and hands f() quads whose lanes are consecutive along a row --
// one store instruction = whole row segments of 128 / 256 bytes.  The stages are dead (the main loop ends with a barrier, so does the
// column-sum exchange); a wave only touches its own region, in program order.  N % 4 == 0.
template <int MT, int NT, int WAVES_N, int A, typename F>
__device__ __forceinline__ void quads_of_row_lds(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, float* mine, F& f) {
    constexpr int WC = 32 * NT, PITCH = WC + 4, LPR = WC / 4, RPI = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 7;
    const int wm = (wave / WAVES_N) * (32 * MT), wn = (wave % WAVES_N) * (32 * NT);
    float* wr = mine + (lane & 31) * PITCH + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(wr + b * 32 + 8 * q) = make_float4(acc[A][b][4 * q], acc[A][b][4 * q + 1], acc[A][b][4 * q + 2], acc[A][b][4 * q + 3]);
    const int c = 4 * (lane % LPR), n = n0 + wn + c;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int r = RPI * i + lane / LPR;
        const float4 t = *reinterpret_cast<const float4*>(mine + r * PITCH + c);
        const int m = m0 + wm + A * 32 + r;
        if (m < M && n < N) {
            Quad x{{t.x, t.y, t.z, t.w}};
            f(m, n, x);
        }
    }
}
template <int MT, int NT, int WAVES_N, typename F, int... As>
__device__ __forceinline__ void for_each_quad_lds_seq(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, float* mine, F& f, std::integer_sequence<int, As...>) {
    (quads_of_row_lds<MT, NT, WAVES_N, As>(acc, m0, n0, M, N, mine, f), ...);
}
// `smem`: the workgroup's stages (4 waves x 32 rows x (32 NT + 4) floats = 35 KB of the 74 KB of a 128 x 128 tile, 18 of 37 KB for 64 x 64)
template <int BM, int BN, int MT, int NT, typename F>
__device__ __forceinline__ void for_each_quad_lds(const f32x16 (&acc)[MT][NT], int m0, int n0, int M, int N, float* smem, F&& f) {
    static_assert(BM == 2 * 32 * MT && BN == 2 * 32 * NT, "2 x 2 waves cover the tile");
    float* mine = smem + ((threadIdx.x >> 6) & 3) * (32 * (32 * NT + 4));
    for_each_quad_lds_seq<MT, NT, 2>(acc, m0, n0, M, N, mine, f, std::make_integer_sequence<int, MT>{});
}

__device__ __forceinline__ bool dev_aligned(const void* p, unsigned mask) { return (reinterpret_cast<uintptr_t>(p) & mask) == 0; }

template <int BM, int BN, int MT, int NT, int WAVES_N = 2>
__device__ __forceinline__ void epilogue(const GemmArgs& p, int M, int m0, int n0, f32x16 (&acc)[MT][NT], float* smem = nullptr) {
    const bool relu = p.flags & SUBGC_GEMM_RELU, accum = p.flags & SUBGC_GEMM_ACCUM;
    const bool vec = p.N % 4 == 0 && p.ldc % 4 == 0 && dev_aligned(p.C, 15) && (!p.bias || dev_aligned(p.bias, 15)) &&
                     (!p.add || (p.ldadd % 4 == 0 && dev_aligned(p.add, 15))) && (!p.keep || dev_aligned(p.keep, 3));
    auto body = [&](int m, int n, Quad& x) {
        int64_t row = m;
        if (p.c_rows) {
            const int g = p.c_rows[m];
            if (g < 0) return;
            row = g;
        }
        if (vec) {
            if (p.bias) { const float4 t = ld4(p.bias + n); x.v[0] += t.x; x.v[1] += t.y; x.v[2] += t.z; x.v[3] += t.w; }
            if (p.add) { const float4 t = ld4(p.add + row * p.ldadd + n); x.v[0] += t.x; x.v[1] += t.y; x.v[2] += t.z; x.v[3] += t.w; }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = fmaxf(x.v[e], 0.f);
            }
            if (p.keep) {
                const uint32_t k4 = *reinterpret_cast<const uint32_t*>(p.keep + row * p.ldc + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = ((k4 >> (8 * e)) & 0xffu) ? x.v[e] * p.keep_scale : 0.f;
            }
            float4* d = reinterpret_cast<float4*>(p.C + row * p.ldc + n);
            if (accum) { const float4 o = *d; x.v[0] += o.x; x.v[1] += o.y; x.v[2] += o.z; x.v[3] += o.w; }
            *d = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = n + e;
            if (col >= p.N) break;
            float v = x.v[e] + (p.bias ? p.bias[col] : 0.f);
            float* dst = p.C + row * p.ldc + col;
            if (p.add) v += p.add[row * p.ldadd + col];
            if (relu) v = fmaxf(v, 0.f);
            if (p.keep) v *= p.keep[row * p.ldc + col] ? p.keep_scale : 0.f;
            if (accum) v += *dst;
            *dst = v;
        }
    };
    if constexpr (WAVES_N == 2) {
        if (smem != nullptr && vec) { for_each_quad_lds<BM, BN, MT, NT>(acc, m0, n0, M, p.N, smem, body); return; }      // (workgroup-uniform)
    }
    for_each_quad<BM, BN, MT, NT, WAVES_N>(acc, m0, n0, M, p.N, body);
}

// ---- workgroup -> tile mapping for L2 locality ------------------------------------------------------
// Workgroup b is dispatched to XCD b % 8 and every XCD has its own 4 MiB L2.  (1) give each XCD a
// CONTIGUOUS chunk of the tile sequence (bijective for any grid size), (2) order the sequence in
// groups of GROUP_M tile-rows walked column by column, so the ~64 workgroups resident on one XCD
// cover an ~8x8 patch of tiles: 8 A panels + 8 B panels, each k-slice fetched into that L2 once and
// hit 7 more times.  (rocprof r01: 35 % TCC miss rate with the naive row-major mapping.)
constexpr int GROUP_M = 8;
__device__ __forceinline__ int xcd_chunked_id(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_group = GROUP_M * tiles_n;
    const int g = t / per_group, first_m = g * GROUP_M, in_g = t - g * per_group;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// data-parallel form: one workgroup per output tile
template <int BM, int BN, bool TA, bool TB, bool VEC, int XM = 0, bool CS = false>
__global__ __launch_bounds__(XM ? 512 : 256, XM >= 3 ? 2 : 1) void gemm_f32_kernel(const GemmArgs p_in) {
    constexpr int MT = BM / 64, NT = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GemmArgs p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    select_problem(p, wg, nwg);
    // m_dev bounds the ROWS of the stored A: M when A is [M,K], K when A is stored transposed [K,M]
    const int M = (p.m_dev && !TA) ? min(p.M, *p.m_dev) : p.M;
    const int K = (p.m_dev && TA) ? min(p.K, *p.m_dev) : p.K;
    // The tile sequence is laid over the LIVE rows (device-side count): with a ragged row count the grid is sized
    // for the allocation, and mapping it over p.M would put every live tile on the first one or two XCDs.
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, live = tiles_m * tiles_n;
    if (wg >= live) return;
    int tm, tn;
    tile_of(xcd_chunked_id(wg, live), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    f32x16 acc[MT][NT];
    zero_acc(acc);
    if constexpr (XM >= 3) {
        mainloop_x16_ws<BM, BN, TA, TB, MT, NT, (XM == 3 ? 6 : 1)>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc);
        if (threadIdx.x >= 256) return;
    } else if constexpr (XM != 0) {
        mainloop_x3_ws<BM, BN, TA, TB, MT, NT, (XM == 1 ? 6 : 1)>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc);
        if (threadIdx.x >= 256) return;                          // staging waves hold no accumulators
    } else if constexpr (CS) {
        float4 cs;
        mainloop<BM, BN, TA, TB, VEC, MT, NT, true>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc, &cs);
        if (n0 == 0) colsum_store<BM>(smem, cs, m0, M, p.cs_out, p.cs_accum != 0);           // workgroup-uniform
    } else mainloop<BM, BN, TA, TB, VEC, MT, NT>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc);
    epilogue<BM, BN, MT, NT>(p, M, m0, n0, acc, XM == 0 ? smem : nullptr);       // (the staging-wave forms have left half the workgroup behind)
}

// split-K form for shapes whose tile count cannot fill 256 CUs (the per-step recurrent GEMMs, M = 640:
// 160 tiles of 128x128).  The K range is cut into `splits` equal parts, workgroup (tile, part) writes
// its raw partial tile to a caller-provided workspace ws[part][M][N] with plain stores (no atomics:
// device-scope fp32 atomics are executed memory-side on this chip and cost more than the GEMM saves),
// and splitk_reduce_kernel sums the parts and applies the (bias / accumulate) epilogue.  Both launches
// are stream-ordered; the workspace is just-written and comes back out of L2 / Infinity Cache.
template <int BM, int BN, bool TA, bool TB, bool VEC, int XM = 0, bool CS = false>
__global__ __launch_bounds__(XM ? 512 : 256, XM >= 3 ? 2 : 1) void gemm_f32_splitk_kernel(const GemmArgs p_in, float* __restrict__ ws, int splits, int kt_per_split) {
    constexpr int MT = BM / 64, NT = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GemmArgs p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    if (select_problem(p, wg, nwg)) ws += (size_t)splits * p.M * p.N;          // the second problem's planes follow the first's
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int u = xcd_chunked_id(wg, nwg);                       // parts of one tile stay on one XCD, back to back
    const int tile = u / splits, part = u - tile * splits;
    int tm, tn;
    tile_of(tile, tiles_m, tiles_n, tm, tn);
    (void)tiles;
    const int m0 = tm * BM, n0 = tn * BN;
    // a device-side row count bounds K of a transposed-A contraction (weight gradients over the ragged attention rows):
    // the parts are then cut from the LIVE K range so that they stay balanced
    const int K = (TA && p.m_dev) ? min(p.K, *p.m_dev) : p.K;
    const int kt_all = (K + BK - 1) / BK;
    if (TA && p.m_dev) kt_per_split = (kt_all + splits - 1) / splits;
    const int kt0 = min(kt_all, part * kt_per_split), kt1 = min(kt_all, kt0 + kt_per_split);
    f32x16 acc[MT][NT];
    zero_acc(acc);
    if constexpr (XM >= 3) {
        mainloop_x16_ws<BM, BN, TA, TB, MT, NT, (XM == 3 ? 6 : 1)>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
        if (threadIdx.x >= 256) return;
    } else if constexpr (XM != 0) {
        mainloop_x3_ws<BM, BN, TA, TB, MT, NT, (XM == 1 ? 6 : 1)>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
        if (threadIdx.x >= 256) return;
    } else if constexpr (CS) {
        float4 cs;
        mainloop<BM, BN, TA, TB, VEC, MT, NT, true>(p, smem, p.M, K, m0, n0, kt0, kt1, acc, &cs);
        if (n0 == 0) colsum_store<BM>(smem, cs, m0, p.M, p.cs_part + (size_t)part * p.M, false);   // an empty part stores zeros
    } else mainloop<BM, BN, TA, TB, VEC, MT, NT>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
    // raw partial tile -> ws[part][m][n] (accumulators hold C^T quads, see epilogue)
    float* out = ws + (size_t)part * p.M * p.N;
    const bool vec = p.N % 4 == 0;                               // ws planes are 16-byte aligned
    auto body = [&](int m, int n, Quad& x) {
        float* d = out + (size_t)m * p.N + n;
        if (vec) { *reinterpret_cast<float4*>(d) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]); return; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.N) d[e] = x.v[e];
    };
    if (XM == 0 && vec) for_each_quad_lds<BM, BN, MT, NT>(acc, m0, n0, p.M, p.N, smem, body);
    else for_each_quad<BM, BN, MT, NT>(acc, m0, n0, p.M, p.N, body);
}

// C = [C +] bias + sum_parts ws[part]   (float4 along N when N % 4 == 0 and C is 16-byte aligned)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C,
                                                            int64_t ldc, const float* __restrict__ bias, int accum, int vec,
                                                            const float* __restrict__ cs_part = nullptr, float* __restrict__ cs_out = nullptr, int cs_accum = 0,
                                                            float* __restrict__ C2 = nullptr, const float* __restrict__ bias2 = nullptr) {
    const size_t plane = (size_t)M * N;
    if (blockIdx.y == 1) { ws += (size_t)splits * plane; C = C2; bias = bias2; }          // pair launch: grid.y = problem
    if (cs_part != nullptr)                                      // the column sums of A that came with the weight gradient, parts added in order
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int s = 0; s < splits; ++s) v += cs_part[(size_t)s * M + m];
            cs_out[m] = cs_accum ? cs_out[m] + v : v;
        }
    if (vec) {
        const int n4 = N >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)M * n4; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t row = i / n4;
            const int c4 = (int)(i - row * n4) * 4;
            float4 v = bias ? *reinterpret_cast<const float4*>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < splits; ++s) {
                const float4 q = *reinterpret_cast<const float4*>(ws + s * plane + (size_t)row * N + c4);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            float4* d = reinterpret_cast<float4*>(C + row * ldc + c4);
            if (accum) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *d = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)M * N; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t row = i / N;
            const int col = (int)(i - row * N);
            float v = bias ? bias[col] : 0.f;
            for (int s = 0; s < splits; ++s) v += ws[s * plane + (size_t)i];
            float* d = C + row * ldc + col;
            *d = accum ? *d + v : v;
        }
    }
}

template <typename KernelT>
int raise_lds(KernelT kernel, size_t lds, uint64_t& done) {
    if (lds <= 64 * 1024) return SUBGC_OK;            // > 64 KiB of dynamic LDS needs the opt-in once per kernel AND DEVICE (bit = device index)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done & bit) return SUBGC_OK;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        subgc::set_error("gemm: cannot raise dynamic LDS limit to %zu", lds);
        return SUBGC_ELAUNCH;
    }
    done |= bit;
    return SUBGC_OK;
}

template <int BM, int BN, bool TA, bool TB, bool VEC, int XM = 0>
int launch(const GemmArgs& a, hipStream_t s) {
    using SA = Stage<BM, TA>;
    using SB = Stage<BN, !TB>;
    const size_t lds = XM >= 3 ? x16_lds_bytes(BM, BN, XM == 3 ? 3 : 1) : XM ? x3_lds_bytes(BM, BN, XM == 1 ? 3 : 1) : sizeof(float) * 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
    dim3 grid((unsigned)(a.nprob * subgc::cdiv(a.N, BN) * subgc::cdiv(a.M, BM)));          // pair launches: both problems' tiles
    if (a.splits_out) *a.splits_out = 1;
    static uint64_t attr_set = 0;
    if constexpr (TA && XM == 0) {
        if (a.cs_out) {
            static uint64_t attr_set_cs = 0;
            if (int rc = raise_lds(gemm_f32_kernel<BM, BN, TA, TB, VEC, 0, true>, lds, attr_set_cs)) return rc;
            hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, VEC, 0, true>), grid, dim3(256), lds, s, a);
            return subgc::check_launch("subgc_gemm_f32_wgrad");
        }
    }
    if (int rc = raise_lds(gemm_f32_kernel<BM, BN, TA, TB, VEC, XM>, lds, attr_set)) return rc;
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, VEC, XM>), grid, dim3(XM ? 512 : 256), lds, s, a);
    return subgc::check_launch("subgc_gemm_f32");
}

// Dispatch constants (measured).  The library reads NO environment variable and keeps no tunable state: what a
// measurement script wants to switch off is a bit of the call's `flags` (SUBGC_GEMM_NO_SPLITK / SUBGC_GEMM_NO_SKINNY).
constexpr int g_two_parts_min_kt = 32;  // two K parts for a tile count between one and two rounds of slots: K-tiles the product needs at least (round 4: 64 -> 32, see pick_tile)
constexpr int g_smallm = 256;          // M <= this prefers the 64x64 split-K form: one 128x128 workgroup per CU is latency-bound
constexpr int g_ragged64 = 1;          // ragged launches use 64x64 tiles
constexpr int g_x3 = 0;                // arithmetic when a call names none: the fp32 matrix pipe

// pick the number of K parts for 128x128 tiles so that tiles x parts fills the 512 workgroup slots
// (2 per CU) in whole rounds; returns 1 when splitting does not pay
inline int choose_splits(int tiles, int kt) {
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 8; ++s) {
        const int per = (kt + s - 1) / s;
        if (s > 1 && per < 12) break;                            // keep >= 384 of K per part
        const int rounds = (tiles * s + 511) / 512;
        const double cost = rounds * (per + 3.0) + 0.6 * (s > 1 ? s + 1 : 0);   // +3: prologue/epilogue, reduce pass
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

template <int BM, int BN, bool TA, bool TB, bool VEC, int XM = 0>
int launch_splitk(const GemmArgs& a, hipStream_t s, int splits, bool reduce = true) {
    using SA = Stage<BM, TA>;
    using SB = Stage<BN, !TB>;
    const size_t lds = XM >= 3 ? x16_lds_bytes(BM, BN, XM == 3 ? 3 : 1) : XM ? x3_lds_bytes(BM, BN, XM == 1 ? 3 : 1) : sizeof(float) * 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
    const int tiles = (int)(a.nprob * subgc::cdiv(a.N, BN) * subgc::cdiv(a.M, BM));
    const int kt = (a.K + BK - 1) / BK, per = (kt + splits - 1) / splits;
    static uint64_t attr_set = 0;
    bool with_cs = false;
    if constexpr (TA && XM == 0) {
        if (a.cs_out) {
            static uint64_t attr_set_cs = 0;
            if (int rc = raise_lds(gemm_f32_splitk_kernel<BM, BN, TA, TB, VEC, 0, true>, lds, attr_set_cs)) return rc;
            hipLaunchKernelGGL((gemm_f32_splitk_kernel<BM, BN, TA, TB, VEC, 0, true>), dim3(tiles * splits), dim3(256), lds, s, a, a.ws, splits, per);
            with_cs = true;
        }
    }
    if (!with_cs) {
        if (int rc = raise_lds(gemm_f32_splitk_kernel<BM, BN, TA, TB, VEC, XM>, lds, attr_set)) return rc;
        hipLaunchKernelGGL((gemm_f32_splitk_kernel<BM, BN, TA, TB, VEC, XM>), dim3(tiles * splits), dim3(XM ? 512 : 256), lds, s, a, a.ws, splits, per);
    }
    if (a.splits_out) *a.splits_out = splits;
    if (!reduce || a.planes_only) return subgc::check_launch("subgc_gemm_f32(split-K, partials)");   // the consumer sums the planes itself
    const int vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0) &&
                    (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                    (a.nprob == 1 || (((reinterpret_cast<uintptr_t>(a.C2) & 15) == 0) && (!a.bias2 || (reinterpret_cast<uintptr_t>(a.bias2) & 15) == 0)));
    const int64_t n = (int64_t)a.M * a.N / (vec ? 4 : 1);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048), (unsigned)a.nprob), dim3(256), 0, s, (const float*)a.ws,
                       splits, a.M, a.N, a.C, a.ldc, a.bias, (a.flags & SUBGC_GEMM_ACCUM) ? 1 : 0, vec,
                       with_cs ? (const float*)a.cs_part : nullptr, a.cs_out, a.cs_accum, a.C2, a.bias2);
    return subgc::check_launch("subgc_gemm_f32(split-K)");
}

template <bool TA, bool TB, bool VEC>
int pick_tile(const GemmArgs& a, hipStream_t s) {
    // 256 CUs: the big tile when it yields >= ~1.5 workgroups per CU; otherwise split-K over big tiles
    // when the epilogue is a plain (bias / accumulate) one and a workspace was registered; else small tiles.
    const int64_t big = a.nprob * subgc::cdiv(a.M, 128) * subgc::cdiv(a.N, 128);                  // pair launches: the tiles of both problems
    // bf16x3-split form (gemm_x3.h): vector-addressable operands, no gathered A rows, 128x128 tiles
    const int xm = (VEC && !a.a_rows) ? a.xmode : 0;
    float* const g_ws = a.ws;
    const size_t g_ws_bytes = a.ws_bytes;
    const int g_splitk = a.no_splitk ? 0 : 1;
    // a ragged launch (device-side row count) is sized for the allocation; its live tiles are usually few, and one 128x128
    // workgroup alone on a CU cannot hide its own load latency: small tiles put several workgroups on every CU
    if (a.m_dev && !TA && xm == 0 && g_ragged64) return launch<64, 64, TA, TB, VEC>(a, s);
    // Between one and two rounds of 512 workgroup slots the second round is nearly empty and its tiles run alone on their CUs
    // (the 600-tile logit weight gradient: 98 -> 111 TFLOP/s with two K parts); the same cost model decides.
    if (big >= 384 && big < 1024 && g_splitk && xm == 0 && !a.add && !a.keep && !(a.flags & SUBGC_GEMM_RELU) && !a.a_rows && !a.c_rows &&
        (!a.m_dev || TA) && g_ws && 2 * (size_t)a.nprob * a.M * a.N * sizeof(float) <= g_ws_bytes) {
        const int kt = (a.K + BK - 1) / BK;
        const double one = (double)((big + 511) / 512) * (kt + 3.0), two = (double)((2 * big + 511) / 512) * ((kt + 1) / 2 + 3.0) + 1.8;
        if (kt >= g_two_parts_min_kt && two < 0.9 * one) return launch_splitk<128, 128, TA, TB, VEC>(a, s, 2);
    }
    // kernel variants: 3 = three-plane split with 16-deep stages (two workgroups per CU), 2 = single bf16 plane with 32-deep stages
    if (big >= 384) return xm == 1 ? launch<128, 128, TA, TB, VEC, VEC ? 3 : 0>(a, s) : xm == 2 ? launch<128, 128, TA, TB, VEC, VEC ? 2 : 0>(a, s)
                                                                                                 : launch<128, 128, TA, TB, VEC>(a, s);
    const bool plain = !a.add && !a.keep && !(a.flags & SUBGC_GEMM_RELU) && !a.a_rows && !a.c_rows && (!a.m_dev || TA);
    if (plain && g_splitk && g_ws && big >= 16 && a.M > g_smallm) {
        // more than half a round of tiles and a short K: the K loop of a part does not get faster with twice the workgroups resident (the
        // operand feed is shared), the planes and the reduce pass come on top -- 4736 x 512 x 1024 (148 tiles): 83 us in two parts, 62 us whole
        const int kt_all = (a.K + BK - 1) / BK;
        const int splits = (big >= 128 && kt_all <= 48) ? 1 : choose_splits((int)big, kt_all);
        if (splits > 1 && big * splits >= 200 && (size_t)splits * a.nprob * a.M * a.N * sizeof(float) <= g_ws_bytes)
            return xm == 1 ? launch_splitk<128, 128, TA, TB, VEC, VEC ? 3 : 0>(a, s, splits)
                           : xm == 2 ? launch_splitk<128, 128, TA, TB, VEC, VEC ? 2 : 0>(a, s, splits) : launch_splitk<128, 128, TA, TB, VEC>(a, s, splits);
    }
    if (plain && g_splitk && g_ws) {
        // small contractions (the per-step h2att projection and its data gradient): split K over 64x64 tiles
        // until ~2 workgroups per CU exist; each part keeps >= 4 K-tiles
        const int small = (int)(a.nprob * subgc::cdiv(a.M, 64) * subgc::cdiv(a.N, 64)), kt = (a.K + BK - 1) / BK;
        int splits = 1;
        while (small * (splits + 1) <= 768 && kt / (splits + 1) >= 4 && splits < 8) ++splits;
        if (splits > 1 && small * splits >= 128 && (size_t)splits * a.nprob * a.M * a.N * sizeof(float) <= g_ws_bytes)
            return launch_splitk<64, 64, TA, TB, VEC>(a, s, splits);
    }
    return launch<64, 64, TA, TB, VEC>(a, s);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile / split choice for one call -- with the row cut in front of it (round 4, as in gemm_bf16.hip): a tile count a few tiles above whole rounds
// of the 512 workgroup slots costs a whole extra round -- Sub_GC_Kar's 8320 relation rows x 1024 columns are 65 x 8 = 520 tiles of
// 128 x 128, 1.016 rounds (106 us against 70 for 8192 rows).  Few rows beyond the last whole round: the whole rounds go out as one launch,
// the remaining rows as a second one with its own tile choice.  A pair's two problems share the rounds.
int dispatch(const GemmArgs& a, int transA, int transB, bool vec, int flags, hipStream_t s) {
    if (!transA && !a.a_rows && !a.c_rows && !a.m_dev && !(flags & SUBGC_GEMM_NO_ROW_CUT)) {
        const int64_t tn = subgc::cdiv(a.N, 128) * a.nprob, tm = subgc::cdiv(a.M, 128);
        if (tn <= 512 && 512 % tn == 0) {
            const int64_t per_round = 512 / tn, whole = tm / per_round * per_round, M1 = whole * 128, rest = a.M - M1;
            if (whole >= per_round && rest > 0 && rest <= 256 && rest * 16 <= M1) {
                GemmArgs head = a, tail = a;
                head.M = (int)M1;
                tail.M = (int)rest;
                tail.A = a.A + M1 * a.lda;
                tail.C = a.C + M1 * a.ldc;
                if (a.add) tail.add = a.add + M1 * a.ldadd;
                if (a.keep) tail.keep = a.keep + M1 * a.ldc;          // the mask shares the destination's leading dimension (epilogue)
                if (a.nprob == 2) { tail.A2 = a.A2 + M1 * a.lda; tail.C2 = a.C2 + M1 * a.ldc; }
                int rc;
                if (transB) rc = vec ? pick_tile<false, true, true>(head, s) : pick_tile<false, true, false>(head, s);
                else rc = vec ? pick_tile<false, false, true>(head, s) : pick_tile<false, false, false>(head, s);
                if (rc) return rc;
                if (transB) return vec ? pick_tile<false, true, true>(tail, s) : pick_tile<false, true, false>(tail, s);
                return vec ? pick_tile<false, false, true>(tail, s) : pick_tile<false, false, false>(tail, s);
            }
        }
    }
    if (!transA && transB) return vec ? pick_tile<false, true, true>(a, s) : pick_tile<false, true, false>(a, s);
    if (!transA && !transB) return vec ? pick_tile<false, false, true>(a, s) : pick_tile<false, false, false>(a, s);
    return vec ? pick_tile<true, false, true>(a, s) : pick_tile<true, false, false>(a, s);
}

}  // namespace

SUBGC_API int subgc_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                             const float* add, int64_t ldadd, const uint8_t* keep, float keep_scale, int flags,
                             const int32_t* a_rows, const int32_t* c_rows, const int32_t* m_dev, void* workspace, size_t ws_bytes,
                             void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size M=%d N=%d K=%d", M, N, K);
    SUBGC_REQUIRE((workspace != nullptr || ws_bytes == 0) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                  "gemm: workspace must be 16-byte aligned (NULL with 0 bytes = none)");
    if (M == 0 || N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(A && B && C, "gemm: null operand");
    SUBGC_REQUIRE(!(transA && transB), "gemm: transA && transB not supported");
    SUBGC_REQUIRE(!(transA && a_rows), "gemm: a_rows needs transA == 0");
    SUBGC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "gemm: leading dimension too small");
    SUBGC_REQUIRE(!add || ldadd >= N, "gemm: ldadd too small");
    const int mode_bits = (flags >> 4) & 3;                  // SUBGC_GEMM_MODE_*: 0 = the process default
    GemmArgs a{A, B, C, bias, add, keep, a_rows, c_rows, m_dev, lda, ldb, ldc, ldadd, M, N, K, flags & 15, keep_scale,
               static_cast<float*>(workspace), ws_bytes, mode_bits ? mode_bits - 1 : g_x3, (flags & SUBGC_GEMM_NO_SPLITK) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    // vector path: every staged line is read as aligned float4 and is all-in or all-out of range
    const bool vecA = aligned16(A) && lda % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
    const bool vecB = aligned16(B) && ldb % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0);
    const bool vec = vecA && vecB;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    if (!transA && transB && M <= 80 && vec && N >= 64 && !keep && !(flags & SUBGC_GEMM_NO_SKINNY) && !(flags & SUBGC_GEMM_ACCUM) && !a_rows && !c_rows && !m_dev &&
        (!add || add != C)) {
        const int rc = subgc::gemm_skinny_nt(A, lda, B, ldb, C, ldc, bias, M, N, K, (flags & SUBGC_GEMM_RELU) ? 1 : 0, s, add, ldadd);
        if (rc != -100) return rc;      // -100: shape not covered by the weight-streaming form
    }
    return dispatch(a, transA, transB, vec, flags, s);
}

// Two products of the SAME shape, layout and epilogue in one launch (see subgc_gemm_bf16_pair): the grid's first half works on
// (A1, B1 -> C1), the second on (A2, B2 -> C2); tile choice, K parts and the row cut are made for both together.  Epilogue: bias, ACCUM.
SUBGC_API int subgc_gemm_f32_pair(int transA, int transB, int M, int N, int K, const float* A1, const float* A2, int64_t lda, const float* B1,
                                  const float* B2, int64_t ldb, float* C1, float* C2, int64_t ldc, const float* bias1, const float* bias2, int flags,
                                  void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32_pair: negative size M=%d N=%d K=%d", M, N, K);
    SUBGC_REQUIRE((workspace != nullptr || ws_bytes == 0) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                  "gemm_f32_pair: workspace must be 16-byte aligned (NULL with 0 bytes = none)");
    if (M == 0 || N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(A1 && A2 && B1 && B2 && C1 && C2, "gemm_f32_pair: null operand");
    SUBGC_REQUIRE(!(transA && transB), "gemm_f32_pair: transA && transB not supported");
    SUBGC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "gemm_f32_pair: leading dimension too small");
    SUBGC_REQUIRE((bias1 != nullptr) == (bias2 != nullptr), "gemm_f32_pair: bias for both problems or for none");
    const int mode_bits = (flags >> 4) & 3;
    GemmArgs a{A1, B1, C1, bias1, nullptr, nullptr, nullptr, nullptr, nullptr, lda, ldb, ldc, 0, M, N, K, flags & 15, 1.f,
               static_cast<float*>(workspace), ws_bytes, mode_bits ? mode_bits - 1 : g_x3, (flags & SUBGC_GEMM_NO_SPLITK) ? 1 : 0};
    a.A2 = A2; a.B2 = B2; a.C2 = C2; a.bias2 = bias2; a.nprob = 2;
    hipStream_t s = (hipStream_t)stream;
    const bool vecA = aligned16(A1) && aligned16(A2) && lda % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
    const bool vecB = aligned16(B1) && aligned16(B2) && ldb % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0);
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 4.0 * M * (double)N * K);
    return dispatch(a, transA, transB, vecA && vecB, flags, s);
}

namespace subgc {
int wgrad_no_rows(float* dW, int64_t lddw, float* db, int M, int N, bool accum_w, bool accum_b, hipStream_t s) {
    if (!accum_b && M > 0 && hipMemsetAsync(db, 0, sizeof(float) * (size_t)M, s) != hipSuccess) { set_error("wgrad: clearing the bias gradient failed"); return SUBGC_ELAUNCH; }
    if (!accum_w && M > 0 && N > 0 &&
        hipMemset2DAsync(dW, sizeof(float) * (size_t)lddw, 0, sizeof(float) * (size_t)N, (size_t)M, s) != hipSuccess) {
        set_error("wgrad: clearing the weight gradient failed");
        return SUBGC_ELAUNCH;
    }
    return SUBGC_OK;
}
}  // namespace subgc

// Weight gradient and bias gradient of one linear layer in one call: dW[M,N] (+)= dY^T x and db[M] (+)= column sums of dY, with dY
// stored [K, M] and x [K, N].  The workgroups of tile column 0 sum the dY tiles they stage anyway (registers, no extra read of dY);
// split-K forms leave [parts][M] partial sums in the tail of the workspace and the reduce pass adds them in order.
SUBGC_API int subgc_gemm_f32_wgrad(int M, int N, int K, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
                                   float* db, int flags, int db_accumulate, const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_f32_wgrad: negative size M=%d N=%d K=%d", M, N, K);
    SUBGC_REQUIRE((workspace != nullptr || ws_bytes == 0) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                  "gemm_f32_wgrad: workspace must be 16-byte aligned (NULL with 0 bytes = none)");
    if (M == 0) return SUBGC_OK;
    SUBGC_REQUIRE(db && (N == 0 || dW), "gemm_f32_wgrad: null destination");
    if (K == 0) return subgc::wgrad_no_rows(dW, lddw, db, M, N, (flags & SUBGC_GEMM_ACCUM) != 0, db_accumulate != 0, (hipStream_t)stream);
    SUBGC_REQUIRE(dY && (N == 0 || X), "gemm_f32_wgrad: null operand");
    SUBGC_REQUIRE(lddy >= M && ldx >= N && lddw >= N, "gemm_f32_wgrad: leading dimension too small");
    const int mode_bits = (flags >> 4) & 3;
    const int xmode = mode_bits ? mode_bits - 1 : g_x3;
    const size_t cs_bytes = ((size_t)8 * M * sizeof(float) + 15) & ~(size_t)15;
    if (N == 0 || xmode != 0) {                              // no product (or one of the opt-in arithmetics, which stage through other loops): two passes
        if (N > 0)
            if (int rc = subgc_gemm_f32(1, 0, M, N, K, dY, lddy, X, ldx, dW, lddw, nullptr, nullptr, 0, nullptr, 1.f, flags, nullptr, nullptr, m_dev,
                                        workspace, ws_bytes, stream)) return rc;
        return subgc_colsum_f32(dY, lddy, K, M, db, db_accumulate, m_dev, workspace, ws_bytes, stream);
    }
    GemmArgs a{dY, X, dW, nullptr, nullptr, nullptr, nullptr, nullptr, m_dev, lddy, ldx, lddw, 0, M, N, K, flags & 15, 1.f,
               static_cast<float*>(workspace), ws_bytes, 0, (flags & SUBGC_GEMM_NO_SPLITK) ? 1 : 0};
    a.cs_out = db; a.cs_accum = db_accumulate ? 1 : 0;
    if (workspace && ws_bytes > cs_bytes) {                      // partial sums of the split-K forms live behind the planes
        a.ws_bytes = (ws_bytes - cs_bytes) & ~(size_t)15;
        a.cs_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + a.ws_bytes);
    } else { a.ws = nullptr; a.ws_bytes = 0; }
    hipStream_t s = (hipStream_t)stream;
    const bool vec = aligned16(dY) && lddy % 4 == 0 && M % 4 == 0 && aligned16(X) && ldx % 4 == 0 && N % 4 == 0;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    return vec ? pick_tile<true, false, true>(a, s) : pick_tile<true, false, false>(a, s);
}

// op(A) op(B) as `*n_planes` fp32 partial planes planes[q][M][N] (plane stride M * N) whose SUM is the product: the split-K forms of the
// dispatch without their reduce pass, for a consumer that adds the planes while it reads them (subgc_lstm_bwd_planes,
// subgc_attn_bwd_planes); *n_planes = 1 when the dispatch does not split (the plain kernel then wrote plane 0).
SUBGC_API int subgc_gemm_f32_planes(int transA, int transB, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                    float* planes, size_t planes_bytes, int* n_planes, int flags, void* stream) {
    SUBGC_REQUIRE(M > 0 && N > 0 && K > 0 && n_planes, "gemm_f32_planes: bad sizes");
    SUBGC_REQUIRE(A && B && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && planes_bytes >= (size_t)M * N * sizeof(float),
                  "gemm_f32_planes: null / misaligned / too small plane buffer");
    SUBGC_REQUIRE(!(transA && transB), "gemm_f32_planes: transA && transB not supported");
    SUBGC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N), "gemm_f32_planes: leading dimension too small");
    const int mode_bits = (flags >> 4) & 3;
    GemmArgs a{A, B, planes, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lda, ldb, N, 0, M, N, K, 0, 1.f,
               planes, planes_bytes, mode_bits ? mode_bits - 1 : g_x3, (flags & SUBGC_GEMM_NO_SPLITK) ? 1 : 0, 1, n_planes};
    *n_planes = 1;
    hipStream_t s = (hipStream_t)stream;
    const bool vecA = aligned16(A) && lda % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
    const bool vecB = aligned16(B) && ldb % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0);
    const bool vec = vecA && vecB;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    if (!transA && transB) return vec ? pick_tile<false, true, true>(a, s) : pick_tile<false, true, false>(a, s);
    if (!transA && !transB) return vec ? pick_tile<false, false, true>(a, s) : pick_tile<false, false, false>(a, s);
    return vec ? pick_tile<true, false, true>(a, s) : pick_tile<true, false, false>(a, s);
}

namespace subgc {
// x[M,K] . W[N,K]^T left as `splits` partial planes ws[part][M][N] in the registered workspace, WITHOUT the reduce pass: for a
// consumer that reads the pre-activations exactly once and can add the planes on the way (the LSTM cell kernel).  Same tile
// and split choice as subgc_gemm_f32 would make for the plain product; -100 when that choice is not the 128x128 split-K form.
int gemm_nt_partials(const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int K, int gemm_flags, float* g_ws, size_t g_ws_bytes,
                     hipStream_t s, int* splits) {
    if (!(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0) || (gemm_flags & SUBGC_GEMM_NO_SPLITK) || !g_ws || M <= g_smallm) return -100;
    const int mode_bits = (gemm_flags >> 4) & 3, xmode = mode_bits ? mode_bits - 1 : g_x3;
    const int64_t big = cdiv(M, 128) * cdiv(N, 128);
    if (big < 16 || big >= 384) return -100;
    const int sp = choose_splits((int)big, (K + BK - 1) / BK);
    if (sp <= 1 || big * sp < 200 || (size_t)sp * M * N * sizeof(float) > g_ws_bytes) return -100;
    GemmArgs a{A, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lda, ldb, N, 0, M, N, K, 0, 1.f, g_ws, g_ws_bytes, xmode, 0};
    ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    const int rc = xmode == 1 ? launch_splitk<128, 128, false, true, true, 3>(a, s, sp, false)
                 : xmode == 2 ? launch_splitk<128, 128, false, true, true, 2>(a, s, sp, false) : launch_splitk<128, 128, false, true, true>(a, s, sp, false);
    *splits = sp;
    return rc;
}
}  // namespace subgc

SUBGC_API int subgc_gemm_workspace_bytes(int M, int N, int K, size_t* bytes) {
    // the most scratch the dispatch of subgc_gemm_f32 can use for this shape (its split-K forms cut K into at most 8 fp32
    // partial planes); a smaller or absent workspace is legal, the dispatch then splits less or uses small tiles
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0 && bytes, "gemm_workspace_bytes: bad arguments");
    const int64_t big = subgc::cdiv(M, 128) * subgc::cdiv(N, 128);
    *bytes = big >= 1024 ? 0 : (size_t)8 * M * N * sizeof(float);
    return SUBGC_OK;
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
// float4 form (N % 4 == 0, 16-byte aligned rows): a lane owns 4 adjacent columns, a wave reads 1 KB of a row per instruction
// part != NULL: the slab's sums go to part[blockIdx.y][N] (plain stores) and colsum_finish_kernel adds the slabs in a fixed order:
// many short slabs fill the chip without hundreds of same-address float atomics per column (see bn_colsum_vec_kernel, graph.hip)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ X, int64_t ldx, int M, int N,
                                                         float* __restrict__ out, const int32_t* m_dev, int rows_per_block, float* __restrict__ part) {
    __shared__ float4 sm[4][64];
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < N) {
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {                                     // four rows requested before the first is added (same order of additions)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(X + (int64_t)(r + 4 * u) * ldx + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; r < r1; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)r * ldx + col);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < N) {
        const float4 a = sm[0][lane], b = sm[1][lane], c = sm[2][lane], d = sm[3][lane];
        const float4 t = make_float4(a.x + b.x + c.x + d.x, a.y + b.y + c.y + d.y, a.z + b.z + c.z + d.z, a.w + b.w + c.w + d.w);
        if (part) *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + col) = t;
        else { atomicAdd(out + col, t.x); atomicAdd(out + col + 1, t.y); atomicAdd(out + col + 2, t.z); atomicAdd(out + col + 3, t.w); }
    }
}
// out[c] (+)= sum over slabs of part[slab][c], fixed order; 256 threads = 64 columns x 4 waves (wave w: slabs w, w+4, ...)
struct ColsumSet { const uint16_t* x[3]; float* out[3]; };        // subgc_colsum_bf16_set: up to three matrices of one shape
__device__ __forceinline__ void colsum_finish_body(const float* __restrict__ part, int slabs, int N, float* __restrict__ out, int accumulate) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < N) {
        int k = w;
        for (; k + 28 < slabs; k += 32) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = part[(int64_t)(k + 4 * q) * N + c];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += x[q];
        }
        for (; k < slabs; k += 4) s += part[(int64_t)k * N + c];
    }
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        const float t = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
        out[c] = accumulate ? out[c] + t : t;
    }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int slabs, int N, float* __restrict__ out, int accumulate) {
    colsum_finish_body(part, slabs, N, out, accumulate);
}
// ... and the finish pass of such a set (blockIdx.y = matrix)
__global__ __launch_bounds__(256) void colsum_finish_set_kernel(const float* __restrict__ part, int slabs, int N, ColsumSet set, int accumulate) {
    colsum_finish_body(part + (size_t)blockIdx.y * slabs * N, slabs, N, set.out[blockIdx.y], accumulate);
}
// bf16 rows (the bf16-stored gate / logit gradients): a lane owns 4 adjacent columns (8 bytes)
__device__ __forceinline__ void colsum_bf16_body(const uint16_t* __restrict__ X, int64_t ldx, int M, int N, float* __restrict__ out,
                                                 const int32_t* m_dev, int rows_per_block, float* __restrict__ part) {
    __shared__ float4 sm[4][64];
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < N) {
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = subgc_load4_bf(X + (int64_t)(r + 4 * u) * ldx + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; r < r1; r += 4) {
            const float4 v = subgc_load4_bf(X + (int64_t)r * ldx + col);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < N) {
        const float4 a = sm[0][lane], b = sm[1][lane], c = sm[2][lane], d = sm[3][lane];
        const float4 t = make_float4(a.x + b.x + c.x + d.x, a.y + b.y + c.y + d.y, a.z + b.z + c.z + d.z, a.w + b.w + c.w + d.w);
        if (part) *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * N + col) = t;
        else { atomicAdd(out + col, t.x); atomicAdd(out + col + 1, t.y); atomicAdd(out + col + 2, t.z); atomicAdd(out + col + 3, t.w); }
    }
}
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ X, int64_t ldx, int M, int N,
                                                          float* __restrict__ out, const int32_t* m_dev, int rows_per_block, float* __restrict__ part) {
    colsum_bf16_body(X, ldx, M, N, out, m_dev, rows_per_block, part);
}
// up to three matrices of ONE shape in one launch (blockIdx.z = matrix; its slab partials at part + z * slabs * N) ...
__global__ __launch_bounds__(256) void colsum_bf16_set_kernel(ColsumSet set, int64_t ldx, int M, int N, int rows_per_block, float* __restrict__ part, int slabs) {
    colsum_bf16_body(set.x[blockIdx.z], ldx, M, N, nullptr, nullptr, rows_per_block, part + (size_t)blockIdx.z * slabs * N);
}
// any N / ld (the 7001-column logit gradients of the Flickr vocabulary): one column per lane
__global__ __launch_bounds__(256) void colsum_bf16_scalar_kernel(const uint16_t* __restrict__ X, int64_t ldx, int M, int N,
                                                                 float* __restrict__ out, const int32_t* m_dev, int rows_per_block) {
    __shared__ float sm[4][64];
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc = 0.f;
    if (col < N)
        for (int r = r0 + w; r < r1; r += 4) acc += subgc_bf2f(X[(int64_t)r * ldx + col]);
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < N) atomicAdd(out + col, sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
}
__global__ void zero_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
}  // namespace

namespace {
// slabs of the float4 column sums: with a workspace ~1024 workgroups of short slabs + one finishing pass; without, slabs of 256
// rows merged by atomics
struct ColsumPlan { int rows_per_block; int slabs; float* part; };
inline ColsumPlan colsum_plan(int M, int N, void* workspace, size_t ws_bytes) {
    const int col_groups = (N / 4 + 63) / 64;
    int rpb = std::max(16, (int)(((int64_t)M * col_groups + 1023) / 1024));
    int slabs = (M + rpb - 1) / rpb;
    if (workspace && N % 4 == 0 && slabs > 8 && (size_t)slabs * N * sizeof(float) <= ws_bytes && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0)
        return ColsumPlan{rpb, slabs, static_cast<float*>(workspace)};
    return ColsumPlan{256, (M + 255) / 256, nullptr};
}
}  // namespace

SUBGC_API int subgc_colsum_bf16(const uint16_t* X, int64_t ldx, int M, int N, float* out, int accumulate, const int32_t* m_dev,
                                void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && ldx >= N, "colsum_bf16: bad sizes");
    if (N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && out, "colsum_bf16: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = N % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 7) == 0;
    const ColsumPlan pl = vec && M > 0 ? colsum_plan(M, N, workspace, ws_bytes) : ColsumPlan{256, (M + 255) / 256, nullptr};
    if (pl.part) {
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N / 4 + 63) / 64, pl.slabs), dim3(256), 0, s, X, ldx, M, N, out, m_dev, pl.rows_per_block, pl.part);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(256), 0, s, (const float*)pl.part, pl.slabs, N, out, accumulate);
        return subgc::check_launch("subgc_colsum_bf16");
    }
    if (!accumulate) hipLaunchKernelGGL(zero_kernel, dim3((N + 255) / 256), dim3(256), 0, s, out, N);
    if (M == 0) return subgc::check_launch("subgc_colsum_bf16");
    const int rows_per_block = 256;
    if (vec) {
        dim3 grid((N / 4 + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, s, X, ldx, M, N, out, m_dev, rows_per_block, (float*)nullptr);
    } else {
        dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(colsum_bf16_scalar_kernel, grid, dim3(256), 0, s, X, ldx, M, N, out, m_dev, rows_per_block);
    }
    return subgc::check_launch("subgc_colsum_bf16");
}

// Column sums of up to three bf16 matrices of ONE shape and leading dimension in two launches instead of six (a GCN unit pair's three bias
// gradients: d(y_a), d(y_b), d(H)).  Falls back to single calls when the slab plan needs more scratch than the workspace has.
SUBGC_API int subgc_colsum_bf16_set(int n, const uint16_t* X0, const uint16_t* X1, const uint16_t* X2, int64_t ldx, int M, int N, float* out0,
                                    float* out1, float* out2, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(n >= 1 && n <= 3 && M >= 0 && N >= 0 && ldx >= N, "colsum_bf16_set: bad sizes");
    if (N == 0) return SUBGC_OK;
    const uint16_t* X[3] = {X0, X1, X2};
    float* out[3] = {out0, out1, out2};
    for (int i = 0; i < n; ++i) SUBGC_REQUIRE(X[i] && out[i], "colsum_bf16_set: null pointer");
    hipStream_t s = (hipStream_t)stream;
    bool vec = N % 4 == 0 && ldx % 4 == 0 && M > 0;
    for (int i = 0; i < n; ++i) vec = vec && (reinterpret_cast<uintptr_t>(X[i]) & 7) == 0;
    ColsumPlan pl = vec ? colsum_plan(M, N, workspace, ws_bytes / (size_t)n) : ColsumPlan{256, 0, nullptr};
    if (!pl.part) {
        for (int i = 0; i < n; ++i)
            if (int rc = subgc_colsum_bf16(X[i], ldx, M, N, out[i], accumulate, nullptr, workspace, ws_bytes, stream)) return rc;
        return SUBGC_OK;
    }
    ColsumSet set{{X0, X1, X2}, {out0, out1, out2}};
    hipLaunchKernelGGL(colsum_bf16_set_kernel, dim3((N / 4 + 63) / 64, pl.slabs, n), dim3(256), 0, s, set, ldx, M, N, pl.rows_per_block, pl.part, pl.slabs);
    hipLaunchKernelGGL(colsum_finish_set_kernel, dim3((N + 63) / 64, n), dim3(256), 0, s, (const float*)pl.part, pl.slabs, N, set, accumulate);
    return subgc::check_launch("subgc_colsum_bf16_set");
}

SUBGC_API int subgc_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out, int accumulate,
                               const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && ldx >= N, "colsum: bad sizes");
    if (N == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && out, "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = N % 4 == 0 && ldx % 4 == 0 && aligned16(X);
    const ColsumPlan pl = vec && M > 0 ? colsum_plan(M, N, workspace, ws_bytes) : ColsumPlan{256, (M + 255) / 256, nullptr};
    if (pl.part) {
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((N / 4 + 63) / 64, pl.slabs), dim3(256), 0, s, X, ldx, M, N, out, m_dev, pl.rows_per_block, pl.part);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 63) / 64), dim3(256), 0, s, (const float*)pl.part, pl.slabs, N, out, accumulate);
        return subgc::check_launch("subgc_colsum_f32");
    }
    if (!accumulate) hipLaunchKernelGGL(zero_kernel, dim3((N + 255) / 256), dim3(256), 0, s, out, N);
    if (M == 0) return subgc::check_launch("subgc_colsum_f32");
    const int rows_per_block = 256;
    if (vec) {
        dim3 grid((N / 4 + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(colsum_vec_kernel, grid, dim3(256), 0, s, X, ldx, M, N, out, m_dev, rows_per_block, (float*)nullptr);
        return subgc::check_launch("subgc_colsum_f32");
    }
    dim3 grid((N + 63) / 64, (M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, X, ldx, M, N, out, accumulate, m_dev, rows_per_block);
    return subgc::check_launch("subgc_colsum_f32");
}
