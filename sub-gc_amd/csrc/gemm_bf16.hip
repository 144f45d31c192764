// bf16-operand GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the arithmetic
// BASELINE configs 3 and 5 name (Full_GC_Kar / Flickr stress: "bf16").  Operands are STORED bf16 in HBM -- weights as a
// bf16 snapshot of the fp32 masters refreshed once per optimizer step, activations written bf16 by the kernel that
// produced them -- so the staging loop moves 2 bytes per element and does no conversion (the fp32-operand "bf16" mode of
// gemm_x3.h rounds fp32 operands on their way to LDS and is bound by moving those fp32 operands through L2).
//
// Same contractions as gemm_f32.hip (reference: every nn.Linear / nn.LSTMCell product of AttModel.py:363-366,376-377,
// 386,411-413,421-423,336-340,453; graph_conv_unit.py:29-30; gpn.py:54,79 and their backward), same epilogues (bias,
// residual add, ReLU, dropout keep-mask, accumulate), results to fp32 and/or bf16.
//
// Workgroup: 256 threads = 4 waves in a 2x2 grid over a 128x128 tile, K in steps of 64; a wave owns 64x64 = 2x2 MFMA tiles
// (64 accumulator registers).  LDS image of BOTH operands, whatever their memory layout: K-contiguous rows of 64 bf16
// (128 B) whose eight 16-byte chunks are XOR-swizzled with (row >> 1) & 7 -- a ds_read_b128 fragment read (32 consecutive
// rows, one chunk) then touches 16 distinct 16-byte slots per 16-lane group: conflict-free without padding, 16 KB per
// operand and stage, 64 KB for two stages, two workgroups per CU.
//   K-contiguous operand (A [M,K] / B = nn.Linear weight [N,K]): a thread moves four 16-byte chunks (8 k of one row) per
//       K-tile, global_load_dwordx4 -> ds_write_b128; 8 lanes cover one 128-byte row.
//   K-major operand (A^T stored [K,M] / B stored [K,N]; every weight-gradient and data-gradient product): a thread loads
//       4 k x 8 rows as four 16-byte row segments, transposes the 4x8 block in registers (16 v_perm_b32) and writes eight
//       ds_write_b64 (row r, 4 consecutive k); 16 lanes write the 16 eight-byte slots of one LDS row: conflict-free.
// Fragment of v_mfma_f32_32x32x16_bf16: lane l holds row (l & 31), k = 8 * (l >> 5) .. +7 of a 16-deep step = ONE
// ds_read_b128 (chunk 2 * step + (l >> 5)).
//
// Pipeline (as gemm_f32.hip's): global loads run two K-tiles ahead in one register set, which is drained into the other
// LDS stage in the middle of a tile and refilled at once; fragments are double-buffered; one barrier per K-tile.
// Split-K (tile count below the 512 workgroup slots): raw fp32 partial tiles to the CALLER's workspace (argument of
// the call, not a global), summed by a reduce kernel that applies the epilogue.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const uint16_t* A; const uint16_t* B; float* C32; uint16_t* C16;
    const float* bias; const float* add; const uint8_t* keep; const int32_t* m_dev;
    int64_t lda, ldb, ldc32, ldc16, ldadd;
    int M, N, K, flags;
    float keep_scale;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROWB = BK * 2;                                   // bytes per LDS row
constexpr int OPER = BM * ROWB;                                // bytes per operand and stage (16 KB)
constexpr int STAGE = 2 * OPER;
constexpr size_t LDS_BYTES = 2 * STAGE;                        // 64 KB

__device__ __forceinline__ uint32_t f2bf(float x) {            // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + (((chunk ^ (row >> 1)) & 7) << 4); }

// ---- staging -----------------------------------------------------------------------------------------------------------
// One operand tile (128 rows x 64 k).  KM = false: memory rows are K-contiguous; KM = true: memory is [k][row].
template <bool KM>
struct Stage {
    uint4 r[4];
    const uint16_t* base[4];
    int64_t ld_;
    unsigned ok;                                               // KM: bit j = r[j] holds a k inside K; !KM: how many of the chunk's 8 k lie inside K
    bool colok;                                                // KM: this thread's 8-row group lies inside the operand

    __device__ __forceinline__ void init(const uint16_t* src, int64_t ld, int row0, int nrows) {
        const int t = threadIdx.x;
        ld_ = ld;
        if (!KM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = min(row0 + (t >> 3) + 32 * v, nrows - 1);      // rows past the edge: a valid row, never stored
                base[v] = src + (int64_t)max(row, 0) * ld + (t & 7) * 8;
            }
            colok = true;
        } else {
            const int col = row0 + (t >> 4) * 8;
            colok = col < nrows;
#pragma unroll
            for (int j = 0; j < 4; ++j) base[j] = src + (colok ? col : 0);     // + k * ld per tile
        }
    }
    // tile starting at k0; K = logical contraction length
    template <bool INTERIOR>
    __device__ __forceinline__ void load(int k0, int K) {
        const int t = threadIdx.x;
        if (!KM) {
            const int k = k0 + (t & 7) * 8;
            const bool in = INTERIOR || k < K;
            ok = INTERIOR ? 8u : (unsigned)min(max(K - k, 0), 8);   // a chunk straddling K keeps its leading K - k elements
#pragma unroll
            for (int v = 0; v < 4; ++v) r[v] = *reinterpret_cast<const uint4*>(base[v] + (in ? k0 : -((t & 7) * 8)));
        } else {
            ok = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + (t & 15) * 4 + j;
                const bool in = INTERIOR || k < K;
                r[j] = *reinterpret_cast<const uint4*>(base[j] + (int64_t)(in ? k : 0) * ld_);
                if (in) ok |= 1u << j;
            }
        }
    }
    template <bool INTERIOR>
    __device__ __forceinline__ void store(unsigned char* lds) const {
        const int t = threadIdx.x;
        if (!KM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (t >> 3) + 32 * v;
                uint4 q = r[v];
                if (!INTERIOR && ok < 8u) {                    // K tail: zero the elements at and past K (the row's ld covers the over-read)
                    const unsigned nv = ok;
                    uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = (2u * d + 1u < nv) ? w[d] : ((2u * d < nv) ? (w[d] & 0xffffu) : 0u);
                    q = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *reinterpret_cast<uint4*>(lds + swz(row, t & 7)) = q;
            }
        } else {
            uint32_t w[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = INTERIOR || ((ok >> j) & 1u);
                w[j][0] = in ? r[j].x : 0u; w[j][1] = in ? r[j].y : 0u; w[j][2] = in ? r[j].z : 0u; w[j][3] = in ? r[j].w : 0u;
            }
            const int kq = t & 15, rg = t >> 4;
#pragma unroll
            for (int e = 0; e < 8; ++e) {                      // row rg*8 + e gets k = 4*kq .. +3
                const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                uint2 o;
                o.x = __builtin_amdgcn_perm(w[1][e >> 1], w[0][e >> 1], sel);
                o.y = __builtin_amdgcn_perm(w[3][e >> 1], w[2][e >> 1], sel);
                const int row = rg * 8 + e;
                *reinterpret_cast<uint2*>(lds + swz(row, kq >> 1) + (kq & 1) * 8) = o;
            }
        }
    }
};

__device__ __forceinline__ bf16x8 frag(const unsigned char* lds, int r0, int step, int lane) {
    const int row = r0 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(lds + swz(row, step * 2 + (lane >> 5)));
}

// acc += A[m0.., kt0*BK .. kt1*BK) x B[.., n0..]
template <bool A_KM, bool B_KM>
__device__ __forceinline__ void mainloop(const Args& p, unsigned char* smem, int M, int K, int m0, int n0, int kt0, int kt1,
                                         f32x16 (&acc)[2][2]) {
    if (kt1 <= kt0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    Stage<A_KM> sa;
    Stage<B_KM> sb;
    sa.init(p.A, p.lda, m0, M);
    sb.init(p.B, p.ldb, n0, p.N);
    sa.template load<false>(kt0 * BK, K);
    sb.template load<false>(kt0 * BK, K);
    sa.template store<false>(smem);
    sb.template store<false>(smem + OPER);
    if (kt0 + 1 < kt1) {
        sa.template load<false>((kt0 + 1) * BK, K);
        sb.template load<false>((kt0 + 1) * BK, K);
    }
    __syncthreads();
    bf16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[0][a] = frag(smem, wm + a * 32, 0, lane);
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[0][b] = frag(smem + OPER, wn + b * 32, 0, lane);

    auto ktile = [&](int kt, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;   // tiles kt+1 and kt+2 exist and are interior: no masks, no tests
        const int cur = (kt - kt0) & 1;
        const unsigned char* lc = smem + cur * STAGE;
        unsigned char* ln = smem + (cur ^ 1) * STAGE;
        const bool has_next = STEADY || kt + 1 < kt1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int fc = s & 1, fn = fc ^ 1;
            if (s + 1 < 4) {
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[fn][a] = frag(lc, wm + a * 32, s + 1, lane);
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[fn][b] = frag(lc + OPER, wn + b * 32, s + 1, lane);
            }
            if (s == 1 && has_next) {                          // drain the staging registers into the other stage, refill them
                sa.template store<STEADY>(ln);
                sb.template store<STEADY>(ln + OPER);
                if (STEADY) {
                    sa.template load<true>((kt + 2) * BK, K);
                    sb.template load<true>((kt + 2) * BK, K);
                } else if (kt + 2 < kt1) {
                    sa.template load<false>((kt + 2) * BK, K);
                    sb.template load<false>((kt + 2) * BK, K);
                }
            }
            if (s == 3) {
                __syncthreads();
                if (has_next) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[fn][a] = frag(ln, wm + a * 32, 0, lane);
#pragma unroll
                    for (int b = 0; b < 2; ++b) fb[fn][b] = frag(ln + OPER, wn + b * 32, 0, lane);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[fc][a], fb[fc][b], acc[a][b], 0, 0, 0);
        }
    };
    int kt = kt0;
    const int steady_end = min(kt1, K / BK) - 2;
    for (; kt < steady_end; ++kt) ktile(kt, std::true_type{});
    for (; kt < kt1; ++kt) ktile(kt, std::false_type{});
}

// workgroup -> tile mapping (see gemm_f32.hip): XCD b % 8 gets a contiguous chunk of a GROUP_M-ordered tile sequence
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

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = (p.m_dev && !A_KM) ? min(p.M, *p.m_dev) : p.M;
    const int K = (p.m_dev && A_KM) ? min(p.K, *p.m_dev) : p.K;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, live = tiles_m * tiles_n;
    if ((int)blockIdx.x >= live) return;
    int tm, tn;
    tile_of(xcd_chunked_id(blockIdx.x, live), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    f32x16 acc[2][2];
    zero_acc(acc);
    mainloop<A_KM, B_KM>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, col_l = lane & 31, hrow = 4 * (lane >> 5);
    const bool relu = p.flags & SUBGC_GEMM_RELU, accum = p.flags & SUBGC_GEMM_ACCUM;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn + b * 32 + col_l;
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                if (m >= M) continue;
                float v = acc[a][b][r] + bias;
                if (p.add) v += p.add[m * p.ldadd + col];
                if (relu) v = fmaxf(v, 0.f);
                if (p.keep) v *= p.keep[m * (int64_t)p.N + col] ? p.keep_scale : 0.f;      // dense [M, N] mask
                if (p.C32) {
                    float* d = p.C32 + m * p.ldc32 + col;
                    if (accum) v += *d;
                    *d = v;
                }
                if (p.C16) p.C16[m * p.ldc16 + col] = (uint16_t)f2bf(v);
            }
    }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_splitk_kernel(const Args p, float* __restrict__ ws, int splits, int kt_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int u = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tile = u / splits, part = u - tile * splits;
    int tm, tn;
    tile_of(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = (A_KM && p.m_dev) ? min(p.K, *p.m_dev) : p.K;
    const int kt_all = (K + BK - 1) / BK;
    if (A_KM && p.m_dev) kt_per_split = (kt_all + splits - 1) / splits;
    const int kt0 = min(kt_all, part * kt_per_split), kt1 = min(kt_all, kt0 + kt_per_split);
    f32x16 acc[2][2];
    zero_acc(acc);
    mainloop<A_KM, B_KM>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, col_l = lane & 31, hrow = 4 * (lane >> 5);
    float* out = ws + (size_t)part * p.M * p.N;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn + b * 32 + col_l;
        if (col >= p.N) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                if (m < p.M) out[(size_t)m * p.N + col] = acc[a][b][r];
            }
    }
}

// C = epilogue(bias + sum_parts ws[part]); fp32 and / or bf16 destination
__global__ __launch_bounds__(256) void splitk_reduce_b16_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C32,
                                                                int64_t ldc32, uint16_t* __restrict__ C16, int64_t ldc16,
                                                                const float* __restrict__ bias, int accum, int relu) {
    const size_t plane = (size_t)M * N;
    const int n4 = N >> 2;                                      // N % 4 == 0 on this path
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)M * n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / n4;
        const int c4 = (int)(i - row * n4) * 4;
        float4 v = bias ? *reinterpret_cast<const float4*>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < splits; ++s) {
            const float4 q = *reinterpret_cast<const float4*>(ws + s * plane + (size_t)row * N + c4);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (C32) {
            float4* d = reinterpret_cast<float4*>(C32 + row * ldc32 + c4);
            if (accum) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *d = v;
        }
        if (C16) {
            uint2 o;
            o.x = f2bf(v.x) | (f2bf(v.y) << 16);
            o.y = f2bf(v.z) | (f2bf(v.w) << 16);
            *reinterpret_cast<uint2*>(C16 + row * ldc16 + c4) = o;
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// number of K parts for `tiles` 128x128 tiles over kt K-tiles of 64: fill the 512 workgroup slots in whole rounds
inline int choose_splits(int64_t tiles, int kt) {
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 8; ++s) {
        const int per = (kt + s - 1) / s;
        if (s > 1 && per < 6) break;                            // keep >= 384 of K per part
        const int64_t rounds = (tiles * s + 511) / 512;
        const double cost = rounds * (per + 2.0) + 0.5 * (s > 1 ? s + 1 : 0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

template <bool A_KM, bool B_KM>
int run(const Args& a, float* ws, size_t ws_bytes, hipStream_t s, bool partials_only, int* splits_out) {
    const int64_t tiles = subgc::cdiv(a.M, BM) * subgc::cdiv(a.N, BN);
    const int kt = (int)subgc::cdiv(a.K, BK);
    const bool plain = !a.add && !a.keep && (!a.m_dev || A_KM) && a.N % 4 == 0 && (!a.C32 || (a.ldc32 % 4 == 0 && aligned16(a.C32))) &&
                       (!a.C16 || (a.ldc16 % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C16) & 7) == 0)) && (!a.bias || aligned16(a.bias));
    int splits = 1;
    if (ws && plain && tiles < 448) {
        splits = choose_splits(tiles, kt);
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > ws_bytes) --splits;
    }
    if (partials_only && splits <= 1) return -100;
    // 64 KB of dynamic LDS: the default limit, no opt-in needed
    if (splits <= 1) {
        hipLaunchKernelGGL((gemm_bf16_kernel<A_KM, B_KM>), dim3((unsigned)tiles), dim3(256), LDS_BYTES, s, a);
        return subgc::check_launch("subgc_gemm_bf16");
    }
    const int per = (kt + splits - 1) / splits;
    hipLaunchKernelGGL((gemm_bf16_splitk_kernel<A_KM, B_KM>), dim3((unsigned)(tiles * splits)), dim3(256), LDS_BYTES, s, a, ws, splits, per);
    if (splits_out) *splits_out = splits;
    if (partials_only) return subgc::check_launch("subgc_gemm_bf16(split-K, partials)");
    const int64_t n = (int64_t)a.M * a.N / 4;
    hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, (const float*)ws,
                       splits, a.M, a.N, a.C32, a.ldc32, a.C16, a.ldc16, a.bias, (a.flags & SUBGC_GEMM_ACCUM) ? 1 : 0,
                       (a.flags & SUBGC_GEMM_RELU) ? 1 : 0);
    return subgc::check_launch("subgc_gemm_bf16(split-K)");
}

int check(int transA, int transB, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb) {
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm_bf16: negative size M=%d N=%d K=%d", M, N, K);
    SUBGC_REQUIRE(!(transA && transB), "gemm_bf16: transA && transB not supported");
    SUBGC_REQUIRE(A && B, "gemm_bf16: null operand");
    SUBGC_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N), "gemm_bf16: leading dimension too small");
    // 16-byte row segments: aligned bases, leading dimensions in multiples of 8 elements (a row's ld then covers the over-read
    // of a chunk that straddles the logical row end; the kernel masks what lies past K, and what lies past M / N only feeds
    // accumulator rows / columns that are never stored)
    SUBGC_REQUIRE(aligned16(A) && aligned16(B) && lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16: operands must be 16-byte aligned with ld %% 8 == 0");
    return SUBGC_OK;
}

}  // namespace

SUBGC_API int subgc_gemm_bf16_workspace_bytes(int M, int N, int K, size_t* bytes) {
    // scratch the split-K form of this shape wants (its fp32 partial planes); 0 when the shape never splits.  A smaller
    // (or no) workspace is legal: the dispatch then splits less (or not at all).
    SUBGC_REQUIRE(M >= 0 && N >= 0 && K >= 0 && bytes, "gemm_bf16_workspace_bytes: bad arguments");
    const int64_t tiles = subgc::cdiv(M, BM) * subgc::cdiv(N, BN);
    const int sp = tiles < 448 ? choose_splits(tiles, (int)subgc::cdiv(K, BK)) : 1;
    *bytes = sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
    return SUBGC_OK;
}

SUBGC_API int subgc_gemm_bf16(int transA, int transB, int M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                              float* C32, int64_t ldc32, uint16_t* C16, int64_t ldc16, const float* bias, const float* add, int64_t ldadd,
                              const uint8_t* keep, float keep_scale, int flags, const int32_t* m_dev, void* workspace, size_t ws_bytes,
                              void* stream) {
    if (M == 0 || N == 0) return SUBGC_OK;
    if (int rc = check(transA, transB, M, N, K, A, lda, B, ldb)) return rc;
    SUBGC_REQUIRE(C32 || C16, "gemm_bf16: no destination");
    SUBGC_REQUIRE((!C32 || ldc32 >= N) && (!C16 || ldc16 >= N) && (!add || ldadd >= N), "gemm_bf16: destination leading dimension too small");
    SUBGC_REQUIRE(!(flags & SUBGC_GEMM_ACCUM) || C32, "gemm_bf16: accumulate needs the fp32 destination");
    SUBGC_REQUIRE(!workspace || aligned16(workspace), "gemm_bf16: workspace must be 16-byte aligned");
    Args a{A, B, C32, C16, bias, add, keep, m_dev, lda, ldb, ldc32, ldc16, ldadd, M, N, K, flags, keep_scale};
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    float* ws = static_cast<float*>(workspace);
    if (!transA && transB) return run<false, false>(a, ws, ws_bytes, s, false, nullptr);
    if (!transA && !transB) return run<false, true>(a, ws, ws_bytes, s, false, nullptr);
    return run<true, true>(a, ws, ws_bytes, s, false, nullptr);
}

namespace subgc {
// x[M,K] . W[N,K]^T (bf16 operands) left as `splits` fp32 partial planes ws[part][M][N] WITHOUT the reduce pass: the LSTM cell
// kernel adds the planes while it reads the pre-activations.  -100 when the dispatch would not split this shape.
int gemm_bf16_nt_partials(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, float* ws, size_t ws_bytes,
                          hipStream_t s, int* splits) {
    if (!ws || !aligned16(A) || !aligned16(B) || lda % 8 || ldb % 8 || N % 4) return -100;
    Args a{A, B, ws, nullptr, nullptr, nullptr, nullptr, nullptr, lda, ldb, N, 0, 0, M, N, K, 0, 1.f};
    ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    return run<false, false>(a, ws, ws_bytes, s, true, splits);
}
}  // namespace subgc

// ---- fp32 <-> bf16 plumbing --------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                            int rows, int cols, int cols_pad, const int32_t* __restrict__ m_dev) {
    // 8 columns per thread when everything is aligned (cols_pad % 8 == 0 enforced by the host for that path)
    if (m_dev) rows = min(rows, *m_dev);
    const int c8 = cols_pad >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * c8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c + e < cols ? x[r * ldx + c + e] : 0.f;     // padding columns are written as zeros
        uint4 o;
        o.x = f2bf(v[0]) | (f2bf(v[1]) << 16); o.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
        o.z = f2bf(v[4]) | (f2bf(v[5]) << 16); o.w = f2bf(v[6]) | (f2bf(v[7]) << 16);
        *reinterpret_cast<uint4*>(y + r * ldy + c) = o;
    }
}
__global__ __launch_bounds__(256) void cast_f32_bf16_vec_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                                int rows, int cols, const int32_t* __restrict__ m_dev) {
    if (m_dev) rows = min(rows, *m_dev);
    const int c8 = cols >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * c8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c), b = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
        uint4 o;
        o.x = f2bf(a.x) | (f2bf(a.y) << 16); o.y = f2bf(a.z) | (f2bf(a.w) << 16);
        o.z = f2bf(b.x) | (f2bf(b.y) << 16); o.w = f2bf(b.z) | (f2bf(b.w) << 16);
        *reinterpret_cast<uint4*>(y + r * ldy + c) = o;
    }
}
// y[c, r] = bf16(x[r, c]): 64x64 tiles through LDS (the W^T snapshots that make every data-gradient product an NT one)
__global__ __launch_bounds__(256) void transpose_f32_bf16_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                                 int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? x[(int64_t)(r0 + r) * ldx + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;                       // output row c0 + c, output column r0 + r
        if (c0 + c < cols && r0 + r < rows) y[(int64_t)(c0 + c) * ldy + r0 + r] = (uint16_t)f2bf(tile[r][c]);
    }
}
__global__ __launch_bounds__(256) void copy2d_b16_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y, int64_t ldy,
                                                         int rows, int cols) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = x[r * ldx + c];
    }
}
}  // namespace

SUBGC_API int subgc_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, int cols_pad,
                                  const int32_t* m_dev, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && cols_pad >= cols && ldx >= cols && ldy >= cols_pad, "cast_f32_bf16: bad sizes");
    if (rows == 0 || cols_pad == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && y, "cast_f32_bf16: null pointer");
    SUBGC_REQUIRE(cols_pad % 8 == 0 && ldy % 8 == 0 && aligned16(y), "cast_f32_bf16: the bf16 side is written in 16-byte pieces (cols_pad, ldy %% 8 == 0)");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)rows * (cols_pad / 8);
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 8192);
    if (cols == cols_pad && ldx % 4 == 0 && aligned16(x))
        hipLaunchKernelGGL(cast_f32_bf16_vec_kernel, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, rows, cols, m_dev);
    else
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, rows, cols, cols_pad, m_dev);
    return subgc::check_launch("subgc_cast_f32_bf16");
}

SUBGC_API int subgc_transpose_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= rows, "transpose_f32_bf16: bad sizes");
    if (rows == 0 || cols == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && y, "transpose_f32_bf16: null pointer");
    hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                       rows, cols);
    return subgc::check_launch("subgc_transpose_f32_bf16");
}

SUBGC_API int subgc_copy2d_b16(const uint16_t* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= cols, "copy2d_b16: bad sizes");
    if (rows == 0 || cols == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && y, "copy2d_b16: null pointer");
    const int64_t n = (int64_t)rows * cols;
    hipLaunchKernelGGL(copy2d_b16_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                       ldy, rows, cols);
    return subgc::check_launch("subgc_copy2d_b16");
}
