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
// Workgroup geometries (struct Geo below): 128x128 (4 waves of 64x64, two workgroups per CU, also the split-K form) and 256x256
// (16 waves of 64x64, one per CU); K advances in 32-deep LDS stages.  Operand tiles travel HBM / L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into a RING of four stages -- no VGPR staging, no ds_write; the LDS image follows each operand's
// MEMORY layout, XOR-swizzled on the DMA's source address: K-contiguous operands are read back with ds_read_b128, K-major ones
// with ds_read_b64_tr_b16 (hardware transpose); counted s_waitcnt vmcnt + raw s_barrier hand a stage over.  Details next to
// Dma / FragAddr / mainloop_dma.  The accumulators hold C^T so that the epilogue stores 16-byte (fp32) / 8-byte (bf16) quads.
// Split-K (tile count below the 512 workgroup slots): raw fp32 partial tiles to the CALLER's workspace (argument of the call,
// not a global), summed by a reduce kernel that applies the epilogue -- or left as planes for a consumer that adds them itself
// (subgc_lstm_fwd_gemm).
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const uint16_t* A; const uint16_t* B; float* C32; uint16_t* C16;
    const float* bias; const float* add; const uint8_t* keep; const int32_t* m_dev;
    int64_t lda, ldb, ldc32, ldc16, ldadd;
    int M, N, K, flags;
    float keep_scale;
    // subgc_gemm_bf16_wgrad (K-major A only): column sums of the stored A -- the bias gradient beside the weight gradient
    float* cs_out = nullptr;         // [M] (cs_accum: added to)
    float* cs_part = nullptr;        // [splits][M] partial sums of the split-K form (tail of the caller's workspace)
    int cs_accum = 0;
    // subgc_gemm_bf16_pair: a SECOND problem of the same shape, layout and epilogue in the same launch (the second half of the grid)
    const uint16_t* A2 = nullptr; const uint16_t* B2 = nullptr; float* C32b = nullptr; uint16_t* C16b = nullptr; const float* bias2 = nullptr;
    int nprob = 1;
};

// Pair launches: workgroups [0, nwg/2) work on problem 0, [nwg/2, nwg) on problem 1.  Rewrites `q` to the workgroup's problem, `b` to its
// problem-local workgroup id and `nwg` to the problem's workgroup count; returns the problem index.  (The XCD of workgroup b is still
// (b + const) % 8: the contiguous-chunk-per-XCD property of xcd_chunked_id holds, only the XCD's identity is rotated.)
__device__ __forceinline__ int select_problem(Args& q, int& b, int& nwg) {
    if (q.nprob != 2) return 0;
    nwg >>= 1;
    if (b < nwg) return 0;
    b -= nwg;
    q.A = q.A2; q.B = q.B2; q.C32 = q.C32b; q.C16 = q.C16b; q.bias = q.bias2;
    return 1;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;                                         // K per LDS stage
constexpr int ROWB = BK * 2;                                   // bytes per LDS row of a K-contiguous image

__device__ __forceinline__ uint32_t f2bf(float x) {            // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__host__ __device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__host__ __device__ __forceinline__ bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
__host__ __device__ __forceinline__ bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + (((chunk ^ (row >> 2)) & 3) << 4); }

// ---- staging: LDS-DMA ring ------------------------------------------------------------------------------------------------
// Tiles travel HBM / L2 -> LDS directly (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPRs, no ds_write) and the
// LDS image follows the MEMORY layout of each operand:
//   K-contiguous operand: [rows][32 k] = 64-byte rows of four 16-byte chunks, chunk ^ ((row >> 2) & 3): the 16 rows a
//       ds_read_b128 lane group reads land on the 16 different 16-byte slots of the 256-byte bank row.  The DMA writes lane l
//       at base + 16 l, so the swizzle is applied to the SOURCE address: a lane fetches the chunk whose swizzled place is
//       its linear slot.  Fragment = one ds_read_b128.
//   K-major operand: [32 k][rows] (k-rows of 2 x rows bytes; chunk ^ ((k & 3) << 2)); fragment = TWO ds_read_b64_tr_b16:
//       the 16 lanes of a group address a [4 k][16 rows] block (lane i: k-row i / 4, rows 4 (i % 4) .. +3) and lane i receives
//       k .. k+3 of row i -- the hardware transpose; the XOR puts the four k-rows of a block into four different 64-byte
//       windows of the bank row (conflict-free, SQ_LDS_BANK_CONFLICT = 0 in both forms).
// Pipeline.  A 2-stage loop (tile f+1 requested while tile f is multiplied) measured 0.45-0.65 PFLOP/s with the waves parked
// 58 % of the time (rocprofv3: SQ_WAIT_ANY): 14 % of the L2 requests of a K-tile miss (every operand slice is fetched once
// per XCD and reused by the 8 workgroups of its row / column, TCC hit rate 86 % = the ideal of that mapping) and a K-tile is
// complete only when its slowest line has arrived, so EVERY iteration paid one Infinity-Cache / HBM round trip (~1.6 us).
// Hence a RING of four 32-deep stages: three tiles (96 KB per CU) are in flight while one is multiplied; a wave waits with
// a COUNTED s_waitcnt vmcnt for its own pieces of the oldest tile only, and a raw s_barrier (no vmcnt(0) drain) hands the
// stage over.  A partial last K-tile (K % 32 != 0) cannot be masked by the DMA and goes through registers, un-pipelined.
typedef short i16x4 __attribute__((ext_vector_type(4)));
#define SUBGC_LDS(p) ((__attribute__((address_space(3))) unsigned char*)(p))

constexpr int NSTAGE = 4;

// Geometry of one workgroup: TBM x TBN output tile, WM x WN waves, each owning MA x NB MFMA tiles of 32x32.  Two instances:
//   128x128, 2x2 waves of 64x64 (256 threads), 4 x 16 KB of LDS, two workgroups per CU -- shapes with few tiles (with split-K):
//       0.67 us per 32-deep K-tile and workgroup = 0.80 PFLOP/s in the K loop;
//   256x256, 4x4 waves of 64x64 (1024 threads), 4 x 32 KB, one workgroup per CU -- the large products: half the operand bytes
//       per flop, 0.85 us per K-tile = 1.26 PFLOP/s in the K loop.  (Measured alternative: 2x4 waves of 128x64 -- twice the MFMAs
//       per barrier and per fragment read, but 8 waves per CU instead of 16 to cover ds_read / barrier latency: 2x SLOWER.)
// For K = 1000 products with fp32 results the K loop is only half of a tile's time: the 256 KB of a 256x256 fp32 tile leave
// at ~11 GB/s per CU = 2.8 TB/s for the chip -- about the HBM write rate (scalar vs 16-byte-quad stores moved it by 5 %) --
// which is why the GEMM-only results are kept bf16 wherever their consumer allows.
template <int TBM_, int TBN_, int MA_, int NB_>
struct Geo {
    static constexpr int TBM = TBM_, TBN = TBN_, MA = MA_, NB = NB_, WM = TBM_ / (32 * MA_), WN = TBN_ / (32 * NB_), NW = WM * WN, NT = NW * 64;
    static constexpr int A_BYTES = TBM_ * ROWB, B_BYTES = TBN_ * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr size_t LDS = NSTAGE * (size_t)STAGE_BYTES;
};

// K-major image of an operand tile with ROWS rows: [32 k][ROWS], 16-byte chunk c of k-row k at chunk c ^ ((k & 3) << 2)
template <int ROWS>
__device__ __forceinline__ int swz_km(int k, int chunk) { return k * (ROWS * 2) + ((chunk ^ ((k & 3) << 2)) << 4); }

template <bool KM, int ROWS, int NW>
struct Dma {
    static constexpr int NI = ROWS * ROWB / 1024 / NW;         // 1 KB wave instructions per wave and tile
    static_assert(NI >= 1 && NI * NW * 1024 == ROWS * ROWB, "tile bytes must split evenly over the waves");
    const uint16_t* src[NI];                                   // this lane's source per instruction, at k0 = 0
    int64_t kstride;

    __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int row0, int nrows) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int v = 0; v < NI; ++v) {
            const int byte = (wave * NI + v) * 1024 + lane * 16;   // this lane's linear place in the operand's region
            if (!KM) {
                const int r = byte / ROWB, pc = (byte % ROWB) >> 4;
                const int c = pc ^ ((r >> 2) & 3);
                const int row = max(min(row0 + r, nrows - 1), 0);
                src[v] = base + (int64_t)row * ld + c * 8;
            } else {
                const int k = byte / (ROWS * 2), pc = (byte % (ROWS * 2)) >> 4;
                const int c = pc ^ ((k & 3) << 2);
                const int col = row0 + c * 8;
                src[v] = base + (int64_t)k * ld + (col < nrows ? col : 0);
            }
        }
        kstride = KM ? ld : 1;
    }
    __device__ __forceinline__ void issue(unsigned char* region, int k0) const {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int v = 0; v < NI; ++v)
            __builtin_amdgcn_global_load_lds(src[v] + (int64_t)k0 * kstride,
                                             (__attribute__((address_space(3))) void*)SUBGC_LDS(region + (wave * NI + v) * 1024), 16, 0, 0);
    }
    // Partial last K-tile THROUGH THE RING (K % 8 == 0: whole 16-byte chunks are in or out).  The DMA cannot mask, so a lane
    // whose chunk / k-row lies at or past K fetches a valid place of the same row instead (chunk 0 / k-row 0 of the tile:
    // k0 < K) and the out-of-range part of the LDS image is zeroed after the tile has landed (zero_past_k): the tile then costs
    // one extra barrier instead of an exposed, un-pipelined global round trip (K = 1000 vs 992 on the logit product measured
    // +36 us with 256x256 tiles and +119 us with 128x128 through the register path below).
    __device__ __forceinline__ void issue_tail(unsigned char* region, int k0, int K) const {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int v = 0; v < NI; ++v) {
            const int byte = (wave * NI + v) * 1024 + lane * 16;
            const uint16_t* src_v = src[v] + (int64_t)k0 * kstride;
            if (!KM) {
                const int r = byte / ROWB, c = ((byte % ROWB) >> 4) ^ ((r >> 2) & 3);
                if (k0 + c * 8 >= K) src_v -= c * 8;
            } else {
                const int k = byte / (ROWS * 2);
                if (k0 + k >= K) src_v -= (int64_t)k * kstride;
            }
            __builtin_amdgcn_global_load_lds(src_v, (__attribute__((address_space(3))) void*)SUBGC_LDS(region + (wave * NI + v) * 1024), 16, 0, 0);
        }
    }
    // zero what lies at k >= rem (rem = K - k0, a multiple of 8 in 8..24) of a landed tile image
    __device__ __forceinline__ void zero_past_k(unsigned char* region, int rem) const {
        constexpr int NT = NW * 64;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        if (!KM) {
            const int c0 = rem >> 3, nz = ROWB / 16 - c0;
            for (int i = threadIdx.x; i < ROWS * nz; i += NT) *reinterpret_cast<uint4*>(region + swz(i / nz, c0 + i % nz)) = z;
        } else {
            uint4* q = reinterpret_cast<uint4*>(region + rem * (ROWS * 2));        // k-rows are contiguous; the swizzle stays inside a k-row
            for (int i = threadIdx.x; i < (BK - rem) * (ROWS * 2) / 16; i += NT) q[i] = z;
        }
    }
    // partial tile through registers (any K): k >= K reads as zero
    __device__ __forceinline__ void tail(unsigned char* region, const uint16_t* base, int64_t ld, int row0, int nrows, int k0, int K) const {
        constexpr int NT = NW * 64;
        for (int i = threadIdx.x; i < ROWS * (ROWB / 16); i += NT) {
            if (!KM) {
                const int r = i / (ROWB / 16), c = i % (ROWB / 16), k = k0 + c * 8;
                const int row = max(min(row0 + r, nrows - 1), 0);
                const int nv = min(max(K - k, 0), 8);
                uint4 q = nv > 0 ? *reinterpret_cast<const uint4*>(base + (int64_t)row * ld + k) : make_uint4(0u, 0u, 0u, 0u);
                if (nv < 8) {
                    uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = (2 * d + 1 < nv) ? w[d] : ((2 * d < nv) ? (w[d] & 0xffffu) : 0u);
                    q = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *reinterpret_cast<uint4*>(region + swz(r, c)) = q;
            } else {
                const int k = i / (ROWS / 8), c = i % (ROWS / 8), col = row0 + c * 8;
                const bool in = k0 + k < K;
                const uint4 q = in ? *reinterpret_cast<const uint4*>(base + (int64_t)(k0 + k) * ld + (col < nrows ? col : 0)) : make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(region + swz_km<ROWS>(k, c)) = q;
            }
        }
    }
};

// Fragment addressing of one operand image, hoisted out of the K loop: `base` = this lane's LDS byte offset for the 32-row
// MFMA tile at r0 and step 0; the second 16-deep step is an XOR (K-contiguous image: chunk index bit 1) or a constant add
// (K-major image: 16 k-rows further; the swizzle only depends on k & 3).
template <bool KM, int ROWS>
struct FragAddr {
    int base;
    __device__ __forceinline__ void init(int r0, int lane) {
        if (!KM) {
            const int row = r0 + (lane & 31);
            base = swz(row, lane >> 5);
        } else {
            const int g = lane >> 4, i = lane & 15;
            const int k = (g >> 1) * 8 + (i >> 2);
            const int c = ((r0 + (g & 1) * 16) >> 3) + ((i & 3) >> 1);
            base = swz_km<ROWS>(k, c) + (i & 1) * 8;
        }
    }
    // K-contiguous image: one ds_read_b128 (the compiler's own load: it schedules and counts it)
    __device__ __forceinline__ bf16x8 load(const unsigned char* region, int step) const {
        return *reinterpret_cast<const bf16x8*>(region + (base ^ (step << 5)));
    }
    // K-major image: two transpose reads, issued from inline asm.  (The __builtin_amdgcn_ds_read_tr16_b64 form makes hipcc's
    // wait-count pass put `s_waitcnt vmcnt(0)` in front of the first such read of every K-tile -- it cannot tell the read from
    // the LDS-DMA writes in flight to the OTHER ring stages -- which drains the ring.)  The caller waits with tr_wait().
    template <int STEP>
    __device__ __forceinline__ void load_tr(const unsigned char* region, unsigned long long& lo, unsigned long long& hi) const {
        const uint32_t addr = (uint32_t)(uintptr_t)SUBGC_LDS(region) + (uint32_t)base;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(STEP * 16 * (ROWS * 2)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(STEP * 16 * (ROWS * 2) + 4 * (ROWS * 2)));
    }
};

// lgkmcnt(0) for transpose reads issued from asm: the values are threaded through so that no consumer is scheduled above it
__device__ __forceinline__ void tr_wait(unsigned long long (&r)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
}
__device__ __forceinline__ void tr_wait(unsigned long long (&r)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}
__device__ __forceinline__ bf16x8 tr_join(unsigned long long lo, unsigned long long hi) {
    union { struct { unsigned long long a, b; } h; bf16x8 v; } u;
    u.h.a = lo; u.h.b = hi;
    return u.v;
}

// s_waitcnt vmcnt(N) with a compile-time N (the ring waits for 0, P or 2 P outstanding DMA instructions, P = per wave and tile:
// 4 in the 128x128 geometry, 2 in the 256x256 one)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// CS: the workgroup also sums the columns of its K-major A tiles (cs[e] = sum over this thread's k-rows of tile column 8 c + e, c =
// t % (TBM / 8)): one ds_read_b128 of the landed image per thread and 16 k-rows (issued before the first 16-deep step, consumed behind its
// wait), nothing extra from memory.  `do_cs` is workgroup-uniform (tile column 0).
template <typename G, bool A_KM, bool B_KM, bool CS = false>
__device__ __forceinline__ void mainloop_dma(const Args& p, unsigned char* smem, int M, int K, int m0, int n0, int kt0, int kt1,
                                             f32x16 (&acc)[G::MA][G::NB], float* cs = nullptr, bool do_cs = false) {
#if defined(__HIP_DEVICE_COMPILE__)       // gfx950 builtins inside: hipcc's host pass gets an empty body
    static_assert(!CS || A_KM, "column sums are read from the K-major image of A");
    constexpr int CS_PER = G::TBM * 4 / G::NT;                  // (16-byte chunk, k-row) pairs of the A image per thread: 2 (128 x 128), 1 (256 x 256)
    if constexpr (CS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    }
    const uint32_t cs_off = CS ? (uint32_t)swz_km<G::TBM>((int)threadIdx.x / (G::TBM / 8), (int)threadIdx.x % (G::TBM / 8)) : 0u;
    if (kt1 <= kt0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MA = G::MA, NB = G::NB;
    const int wm = (wave / G::WN) * (32 * MA), wn = (wave % G::WN) * (32 * NB);
    using DA = Dma<A_KM, G::TBM, G::NW>;
    using DB = Dma<B_KM, G::TBN, G::NW>;
    constexpr int P = DA::NI + DB::NI;                         // DMA instructions per wave and tile
    DA da;
    DB db;
    da.init(p.A, p.lda, m0, M);
    db.init(p.B, p.ldb, n0, p.N);
    const int full_end = min(kt1, K / BK);                     // tiles kt0 .. full_end-1 lie entirely inside K
    const int F = max(full_end - kt0, 0);
    FragAddr<A_KM, G::TBM> xa[MA];
    FragAddr<B_KM, G::TBN> xb[NB];
#pragma unroll
    for (int a = 0; a < MA; ++a) xa[a].init(wm + a * 32, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) xb[b].init(wn + b * 32, lane);
    auto step = [&](const unsigned char* st, auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        bf16x8 fa[MA], fb[NB];
        unsigned long long ra[2 * MA], rb[2 * NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            if (A_KM) xa[a].template load_tr<S>(st, ra[2 * a], ra[2 * a + 1]);
            else fa[a] = xa[a].load(st, S);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (B_KM) xb[b].template load_tr<S>(st + G::A_BYTES, rb[2 * b], rb[2 * b + 1]);
            else fb[b] = xb[b].load(st + G::A_BYTES, S);
        }
        if (A_KM) {
            tr_wait(ra);
#pragma unroll
            for (int a = 0; a < MA; ++a) fa[a] = tr_join(ra[2 * a], ra[2 * a + 1]);
        }
        if (B_KM) {
            tr_wait(rb);
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b] = tr_join(rb[2 * b], rb[2 * b + 1]);
        }
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);   // C^T tile: see the epilogue
    };
    auto compute = [&](const unsigned char* st) {
        static_assert(BK == 32, "two 16-deep steps per stage");
        u32x4 q[CS_PER];
        if constexpr (CS) {
            if (do_cs) {
                const uint32_t addr = (uint32_t)(uintptr_t)SUBGC_LDS(st) + cs_off;
#pragma unroll
                for (int j = 0; j < CS_PER; ++j)                // k-rows k and k + 16: the swizzle only depends on k & 3
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[j]) : "v"(addr), "n"(j * 16 * (G::TBM * 2)));
            }
        }
        step(st, std::integral_constant<int, 0>{});
        if constexpr (CS) {
            if (do_cs) {                                        // the step's own lgkmcnt(0) covered these reads (LDS returns in order)
                if constexpr (CS_PER == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]));
#pragma unroll
                for (int j = 0; j < CS_PER; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        cs[2 * d] += __uint_as_float(q[j][d] << 16);
                        cs[2 * d + 1] += __uint_as_float(q[j][d] & 0xffff0000u);
                    }
            }
        }
        step(st, std::integral_constant<int, 1>{});
    };
    auto issue = [&](int f) {                                  // full tile f (0-based in this unit) -> stage f % NSTAGE
        unsigned char* st = smem + (f % NSTAGE) * G::STAGE_BYTES;
        da.issue(st, (kt0 + f) * BK);
        db.issue(st + G::A_BYTES, (kt0 + f) * BK);
    };
    // the partial last tile of K rides the ring as tile F when its boundary falls between 16-byte chunks
    const bool has_tail = full_end < kt1, ring_tail = has_tail && K % 8 == 0;
    const int FT = F + (ring_tail ? 1 : 0);
    auto request = [&](int f) {
        if (f < F) { issue(f); return; }
        unsigned char* st = smem + (f % NSTAGE) * G::STAGE_BYTES;
        da.issue_tail(st, full_end * BK, K);
        db.issue_tail(st + G::A_BYTES, full_end * BK, K);
    };
    for (int f = 0; f < min(FT, NSTAGE - 1); ++f) request(f);
    for (int f = 0; f < FT; ++f) {
        // this wave's pieces of tile f have landed once at most min(FT - f - 1, NSTAGE - 2) younger tiles are outstanding
        const int younger = FT - f - 1;
        if (younger >= 2) wait_vmcnt<2 * P>();
        else if (younger == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                          // everybody's pieces have; everybody is done reading stage (f - 1) % NSTAGE
        asm volatile("" ::: "memory");                         // no LDS read of this tile may be scheduled above the barrier
        if (f + NSTAGE - 1 < FT) request(f + NSTAGE - 1);
        unsigned char* st = smem + (f % NSTAGE) * G::STAGE_BYTES;
        if (f == F) {                                          // the ring's tail tile: blank k >= K, then everybody may read it
            da.zero_past_k(st, K - full_end * BK);
            db.zero_past_k(st + G::A_BYTES, K - full_end * BK);
            __syncthreads();
        }
        compute(st);
    }
    if (has_tail && !ring_tail) {                              // K % 8 != 0: element masks, through registers
        __syncthreads();
        da.tail(smem, p.A, p.lda, m0, M, full_end * BK, K);
        db.tail(smem + G::A_BYTES, p.B, p.ldb, n0, p.N, full_end * BK, K);
        __syncthreads();
        compute(smem);
    }
#endif
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

template <int MA, int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MA][NB]) {
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// Epilogue.  The MFMAs are issued with the operands SWAPPED (B fragment first), so an accumulator tile holds C^T: lane l owns
// output ROW m = (l & 31) and register r the column n = (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the 32x32 tile -- four CONSECUTIVE
// columns per register quad, i.e. one 16-byte (fp32) or 8-byte (bf16) store per quad instead of four scalar stores that a
// column-per-lane layout needs (64 dword stores per lane and tile made the store issue as long as the whole K loop at K = 1000).
struct Quad { float v[4]; };
// one row of 32 x 32 tiles (fixed a): instantiated per a through a fold, so the accumulator index is a constant whatever the unroller decides
// (with MA = 4 and the full epilogue as the body, "#pragma unroll" over a was declined and the accumulators went through scratch)
template <typename G, int A, typename F>
__device__ __forceinline__ void quads_of_row(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, F& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / G::WN) * (32 * G::MA), wn = (wave % G::WN) * (32 * G::NB);
    const int m = m0 + wm + A * 32 + (lane & 31);
    if (m >= M) return;
#pragma unroll
    for (int b = 0; b < G::NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn + b * 32 + 8 * q + 4 * (lane >> 5);
            if (n >= N) continue;
            Quad x{{acc[A][b][4 * q], acc[A][b][4 * q + 1], acc[A][b][4 * q + 2], acc[A][b][4 * q + 3]}};
            f(m, n, x);
        }
}
template <typename G, typename F, int... As>
__device__ __forceinline__ void for_each_quad_seq(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, F& f, std::integer_sequence<int, As...>) {
    (quads_of_row<G, As>(acc, m0, n0, M, N, f), ...);
}
template <typename G, typename F>
__device__ __forceinline__ void for_each_quad(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, F&& f) {
    for_each_quad_seq<G>(acc, m0, n0, M, N, f, std::make_integer_sequence<int, G::MA>{});
}

// The same walk with the accumulators TRANSPOSED THROUGH LDS first (round 6).  In the register layout above one store instruction
// touches 32 different rows and writes 32 (fp32) or 16 (bf16) bytes of each -- every 128-byte line of C is written in four separate
// instructions, and a 256 x 256 tile's epilogue ran at 1.9 TB/s (bf16 result) / 2.9 TB/s (fp32) when the whole chip stored at once:
// 16-20 us appended to every round of the one-workgroup-per-CU form.  Here each wave writes a 32-row slab of its sub-tile into its
// own LDS region (row pitch + 4 floats: the 16-byte writes of 16 lanes fall on 64 distinct banks), reads it back row-major and hands
// f() quads whose lanes are CONSECUTIVE along a row: one store instruction = 4 (or 2) whole row segments of 256 (512) bytes.  The
// stages are dead by then (one barrier); a wave only ever touches its own region, in program order, so no further barrier is needed.
template <typename G, int A, typename F>
__device__ __forceinline__ void quads_of_row_lds(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, float* mine, F& f) {
    constexpr int WC = 32 * G::NB, PITCH = WC + 4, LPR = WC / 4, RPI = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / G::WN) * (32 * G::MA), wn = (wave % G::WN) * (32 * G::NB);
    float* wr = mine + (lane & 31) * PITCH + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < G::NB; ++b)
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
template <typename G, typename F, int... As>
__device__ __forceinline__ void for_each_quad_lds_seq(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, float* mine, F& f,
                                                      std::integer_sequence<int, As...>) {
    (quads_of_row_lds<G, As>(acc, m0, n0, M, N, mine, f), ...);
}
// N % 4 == 0 (whole quads).  `smem`: the workgroup's dynamic LDS (>= NW * 32 * (32 NB + 4) floats: 35 / 68 / 70 KB of the 64 / 128 / 128 the forms own)
template <typename G, typename F>
__device__ __forceinline__ void for_each_quad_lds(const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, int N, unsigned char* smem, F&& f) {
    constexpr int PITCH = 32 * G::NB + 4;
    if constexpr ((size_t)G::NW * 32 * PITCH * 4 <= G::LDS || G::NW == 8) {
        float* mine = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * (32 * PITCH);
        __syncthreads();                                        // every wave is past its last stage read
        for_each_quad_lds_seq<G>(acc, m0, n0, M, N, mine, f, std::make_integer_sequence<int, G::MA>{});
    } else for_each_quad<G>(acc, m0, n0, M, N, f);              // the 16-wave form: its slabs would not fit (139 KB), it keeps the register walk
}

// A bf16-ONLY destination with nothing but bias / ReLU in the epilogue: the quads are finished and packed in registers, the slab holds
// bf16 (row pitch + 16 bytes), a lane reads 16 bytes = EIGHT columns back and one store instruction writes 8 whole 128-byte row
// segments -- half the LDS traffic and half the store instructions of the fp32 slab (a bf16 result through the fp32 slab stored
// 8 bytes per lane and took LONGER than an fp32 result of twice the bytes).
template <typename G, int A>
__device__ __forceinline__ void quads_of_row_lds_b16(const Args& p, const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, unsigned char* mine, bool relu) {
    constexpr int WC = 32 * G::NB, PITCH = WC * 2 + 16, LPR = WC / 8, RPI = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / G::WN) * (32 * G::MA), wn = (wave % G::WN) * (32 * G::NB);
    unsigned char* wr = mine + (lane & 31) * PITCH + 8 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < G::NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4] = {acc[A][b][4 * q], acc[A][b][4 * q + 1], acc[A][b][4 * q + 2], acc[A][b][4 * q + 3]};
            const int n = n0 + wn + b * 32 + 8 * q + 4 * (lane >> 5);
            if (p.bias && n < p.N) { const float4 t = *reinterpret_cast<const float4*>(p.bias + n); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            uint2 o;
            o.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
            o.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(wr + (b * 32 + 8 * q) * 2) = o;
        }
    const int c = 8 * (lane % LPR), n = n0 + wn + c;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int r = RPI * i + lane / LPR;
        const uint4 t = *reinterpret_cast<const uint4*>(mine + r * PITCH + c * 2);
        const int m = m0 + wm + A * 32 + r;
        if (m < M && n < p.N) *reinterpret_cast<uint4*>(p.C16 + (int64_t)m * p.ldc16 + n) = t;
    }
}
template <typename G, int... As>
__device__ __forceinline__ void store_tile_b16_lds_seq(const Args& p, const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, unsigned char* mine, bool relu,
                                                       std::integer_sequence<int, As...>) {
    (quads_of_row_lds_b16<G, As>(p, acc, m0, n0, M, mine, relu), ...);
}

// The column sums a workgroup's threads hold after mainloop_dma<.., CS = true>: thread t has columns 8 (t % (TBM/8)) .. +7 over its k-rows;
// the NT / (TBM/8) row groups meet in LDS (everybody is past the last stage read after the barrier) and thread m < TBM stores column
// m0 + m.  Fixed summation order.
template <typename G>
__device__ __forceinline__ void colsum_store(unsigned char* smem, const float (&cs)[8], int m0, int M, float* dst, bool accum) {
    constexpr int CH = G::TBM / 8, GR = G::NT / CH;
    float* red = reinterpret_cast<float*>(smem);
    const int t = threadIdx.x;
    __syncthreads();
    float* mine = red + (t / CH) * G::TBM + (t % CH) * 8;
    *reinterpret_cast<float4*>(mine) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    __syncthreads();
    if (t < G::TBM && m0 + t < M) {
        float v = 0.f;
        for (int g = 0; g < GR; ++g) v += red[g * G::TBM + t];
        dst[m0 + t] = accum ? dst[m0 + t] + v : v;
    }
}

// bias / residual / ReLU / keep-mask / accumulate, results to fp32 and / or bf16 (one 16-byte / 8-byte store per accumulator quad)
template <typename G>
__device__ __forceinline__ void store_tile(const Args& p, const f32x16 (&acc)[G::MA][G::NB], int m0, int n0, int M, unsigned char* smem) {
    const bool relu = p.flags & SUBGC_GEMM_RELU, accum = p.flags & SUBGC_GEMM_ACCUM;
    // vector form: every quad is whole (N % 4 == 0) and every row start 16 / 8 bytes aligned
    const bool vec = p.N % 4 == 0 && (!p.C32 || (p.ldc32 % 4 == 0 && aligned16(p.C32))) && (!p.C16 || (p.ldc16 % 4 == 0 && aligned8(p.C16))) &&
                     (!p.bias || aligned16(p.bias)) && (!p.add || (p.ldadd % 4 == 0 && aligned16(p.add))) && (!p.keep || aligned4(p.keep));
    auto body = [&](int m, int n, Quad& x) {
        const int64_t row = m;
        if (vec) {
            if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + n); x.v[0] += t.x; x.v[1] += t.y; x.v[2] += t.z; x.v[3] += t.w; }
            if (p.add) { const float4 t = *reinterpret_cast<const float4*>(p.add + row * p.ldadd + n); x.v[0] += t.x; x.v[1] += t.y; x.v[2] += t.z; x.v[3] += t.w; }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = fmaxf(x.v[e], 0.f);
            }
            if (p.keep) {                                       // dense [M, N] mask
                const uint32_t k4 = *reinterpret_cast<const uint32_t*>(p.keep + row * p.N + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = ((k4 >> (8 * e)) & 0xffu) ? x.v[e] * p.keep_scale : 0.f;
            }
            if (p.C32) {
                float4* d = reinterpret_cast<float4*>(p.C32 + row * p.ldc32 + n);
                if (accum) { const float4 o = *d; x.v[0] += o.x; x.v[1] += o.y; x.v[2] += o.z; x.v[3] += o.w; }
                *d = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
            }
            if (p.C16) {
                uint2 o;
                o.x = f2bf(x.v[0]) | (f2bf(x.v[1]) << 16);
                o.y = f2bf(x.v[2]) | (f2bf(x.v[3]) << 16);
                *reinterpret_cast<uint2*>(p.C16 + row * p.ldc16 + n) = o;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = n + e;
            if (col >= p.N) break;
            float v = x.v[e] + (p.bias ? p.bias[col] : 0.f);
            if (p.add) v += p.add[row * p.ldadd + col];
            if (relu) v = fmaxf(v, 0.f);
            if (p.keep) v *= p.keep[row * p.N + col] ? p.keep_scale : 0.f;
            if (p.C32) {
                float* d = p.C32 + row * p.ldc32 + col;
                if (accum) v += *d;
                *d = v;
            }
            if (p.C16) p.C16[row * p.ldc16 + col] = (uint16_t)f2bf(v);
        }
    };
    constexpr bool slabs = (size_t)G::NW * 32 * (32 * G::NB + 4) * 4 <= G::LDS || G::NW == 8;
    // (all three conditions are workgroup-uniform)
    if (slabs && vec && p.C16 && !p.C32 && !p.add && !p.keep && p.N % 8 == 0 && p.ldc16 % 8 == 0 && aligned16(p.C16)) {
        unsigned char* mine = smem + (threadIdx.x >> 6) * (32 * (32 * G::NB * 2 + 16));
        __syncthreads();                                        // every wave is past its last stage read
        store_tile_b16_lds_seq<G>(p, acc, m0, n0, M, mine, relu, std::make_integer_sequence<int, G::MA>{});
    } else if (vec) for_each_quad_lds<G>(acc, m0, n0, M, p.N, smem, body);
    else for_each_quad<G>(acc, m0, n0, M, p.N, body);
}

template <typename G, bool A_KM, bool B_KM, bool CS = false>
__global__ __launch_bounds__(G::NT, (G::NT == 256 ? 2 : G::NT / 256)) void gemm_bf16_kernel(const Args p_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Args p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    select_problem(p, wg, nwg);
    const int M = (p.m_dev && !A_KM) ? min(p.M, *p.m_dev) : p.M;
    const int K = (p.m_dev && A_KM) ? min(p.K, *p.m_dev) : p.K;
    const int tiles_m = (M + G::TBM - 1) / G::TBM, tiles_n = (p.N + G::TBN - 1) / G::TBN, live = tiles_m * tiles_n;
    if (wg >= live) return;
    int tm, tn;
    tile_of(xcd_chunked_id(wg, live), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * G::TBM, n0 = tn * G::TBN;
    f32x16 acc[G::MA][G::NB];
    zero_acc(acc);
    if constexpr (CS) {
        float cs[8];
        mainloop_dma<G, A_KM, B_KM, true>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc, cs, n0 == 0);
        if (n0 == 0) colsum_store<G>(smem, cs, m0, M, p.cs_out, p.cs_accum != 0);
    } else mainloop_dma<G, A_KM, B_KM>(p, smem, M, K, m0, n0, 0, (K + BK - 1) / BK, acc);
    store_tile<G>(p, acc, m0, n0, M, smem);
}

template <typename G, bool A_KM, bool B_KM, bool CS = false>
__global__ __launch_bounds__(G::NT, (G::NT == 256 ? 2 : G::NT / 256)) void gemm_bf16_splitk_kernel(const Args p_in, float* __restrict__ ws, int splits, int kt_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Args p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    if (select_problem(p, wg, nwg)) ws += (size_t)splits * p.M * p.N;          // the second problem's planes follow the first's
    const int tiles_m = (p.M + G::TBM - 1) / G::TBM, tiles_n = (p.N + G::TBN - 1) / G::TBN;
    const int u = xcd_chunked_id(wg, nwg);
    const int tile = u / splits, part = u - tile * splits;
    int tm, tn;
    tile_of(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * G::TBM, n0 = tn * G::TBN;
    const int K = (A_KM && p.m_dev) ? min(p.K, *p.m_dev) : p.K;
    const int kt_all = (K + BK - 1) / BK;
    if (A_KM && p.m_dev) kt_per_split = (kt_all + splits - 1) / splits;
    const int kt0 = min(kt_all, part * kt_per_split), kt1 = min(kt_all, kt0 + kt_per_split);
    f32x16 acc[G::MA][G::NB];
    zero_acc(acc);
    if constexpr (CS) {
        float cs[8];
        mainloop_dma<G, A_KM, B_KM, true>(p, smem, p.M, K, m0, n0, kt0, kt1, acc, cs, n0 == 0);
        if (n0 == 0) colsum_store<G>(smem, cs, m0, p.M, p.cs_part + (size_t)part * p.M, false);    // an empty part stores zeros
    } else mainloop_dma<G, A_KM, B_KM>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
    float* out = ws + (size_t)part * p.M * p.N;                 // raw partial plane [M][N]; N % 4 == 0 on this path
    for_each_quad_lds<G>(acc, m0, n0, p.M, p.N, smem, [&](int m, int n, Quad& x) {
        *reinterpret_cast<float4*>(out + (size_t)m * p.N + n) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
    });
}

// C = epilogue(bias + sum_parts ws[part]); fp32 and / or bf16 destination
__global__ __launch_bounds__(256) void splitk_reduce_b16_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C32,
                                                                int64_t ldc32, uint16_t* __restrict__ C16, int64_t ldc16,
                                                                const float* __restrict__ bias, int accum, int relu,
                                                                const float* __restrict__ cs_part = nullptr, float* __restrict__ cs_out = nullptr,
                                                                int cs_accum = 0, float* __restrict__ C32b = nullptr, uint16_t* __restrict__ C16b = nullptr,
                                                                const float* __restrict__ bias2 = nullptr, const uint8_t* __restrict__ keep = nullptr,
                                                                float keep_scale = 1.f) {
    const size_t plane = (size_t)M * N;
    if (blockIdx.y == 1) { ws += (size_t)splits * plane; C32 = C32b; C16 = C16b; bias = bias2; }     // pair launch: grid.y = problem
    if (cs_part != nullptr)                                     // column sums of A that came with a weight gradient: parts added in order
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int s = 0; s < splits; ++s) v += cs_part[(size_t)s * M + m];
            cs_out[m] = cs_accum ? cs_out[m] + v : v;
        }
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
        if (keep) {                                             // dense [M, N] dropout mask, after the ReLU like in store_tile (single-problem launches only)
            const uint32_t k4 = *reinterpret_cast<const uint32_t*>(keep + row * N + c4);
            v.x = (k4 & 0xffu) ? v.x * keep_scale : 0.f; v.y = ((k4 >> 8) & 0xffu) ? v.y * keep_scale : 0.f;
            v.z = ((k4 >> 16) & 0xffu) ? v.z * keep_scale : 0.f; v.w = (k4 >> 24) ? v.w * keep_scale : 0.f;
        }
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

using GP8 = Geo<256, 256, 4, 2>;                                // 2 x 4 waves of 128 x 64: the accumulator geometry of gemm_bf16_p8.h (its LDS is laid out there)

#include "gemm_bf16_p8.h"

template <bool A_KM, bool B_KM, bool CS = false>
__global__ __launch_bounds__(p8::NT) void gemm_bf16_p8_kernel(const Args p_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Args p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    select_problem(p, wg, nwg);
    const int M = (p.m_dev && !A_KM) ? min(p.M, *p.m_dev) : p.M;
    const int K = (p.m_dev && A_KM) ? min(p.K, *p.m_dev) : p.K;
    const int tiles_m = (M + 255) / 256, tiles_n = (p.N + 255) / 256, live = tiles_m * tiles_n;
    if (wg >= live) return;
    int tm, tn;
    tile_of(xcd_chunked_id(wg, live), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    f32x16 acc[4][2];
    zero_acc(acc);
    if constexpr (CS) {
        float cs[2][8];
        p8::mainloop<A_KM, B_KM, true>(p, smem, M, K, m0, n0, 0, (K + p8::KT - 1) / p8::KT, acc, cs, n0 == 0);
        if (n0 == 0) p8::colsum_store(smem, cs, m0, M, p.cs_out, p.cs_accum != 0);
    } else p8::mainloop<A_KM, B_KM>(p, smem, M, K, m0, n0, 0, (K + p8::KT - 1) / p8::KT, acc);
    store_tile<GP8>(p, acc, m0, n0, M, smem);
}

template <bool A_KM, bool B_KM, bool CS = false>
__global__ __launch_bounds__(p8::NT) void gemm_bf16_p8_splitk_kernel(const Args p_in, float* __restrict__ ws, int splits, int kt_per_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Args p = p_in;
    int wg = blockIdx.x, nwg = gridDim.x;
    if (select_problem(p, wg, nwg)) ws += (size_t)splits * p.M * p.N;
    const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
    const int u = xcd_chunked_id(wg, nwg);
    const int tile = u / splits, part = u - tile * splits;
    int tm, tn;
    tile_of(tile, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int K = (A_KM && p.m_dev) ? min(p.K, *p.m_dev) : p.K;
    const int kt_all = (K + p8::KT - 1) / p8::KT;
    if (A_KM && p.m_dev) kt_per_split = (kt_all + splits - 1) / splits;
    const int kt0 = min(kt_all, part * kt_per_split), kt1 = min(kt_all, kt0 + kt_per_split);
    f32x16 acc[4][2];
    zero_acc(acc);
    if constexpr (CS) {
        float cs[2][8];
        p8::mainloop<A_KM, B_KM, true>(p, smem, p.M, K, m0, n0, kt0, kt1, acc, cs, n0 == 0);
        if (n0 == 0) p8::colsum_store(smem, cs, m0, p.M, p.cs_part + (size_t)part * p.M, false);   // an empty part stores zeros
    } else p8::mainloop<A_KM, B_KM>(p, smem, p.M, K, m0, n0, kt0, kt1, acc);
    float* out = ws + (size_t)part * p.M * p.N;
    for_each_quad_lds<GP8>(acc, m0, n0, p.M, p.N, smem, [&](int m, int n, Quad& x) {
        *reinterpret_cast<float4*>(out + (size_t)m * p.N + n) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
    });
}

using G128 = Geo<128, 128, 2, 2>;
using G256 = Geo<256, 256, 2, 2>;

// Cost model of a launch, in microseconds, from measured constants (tools/gemm_bf16_bench.py with SUBGC_BF16_TILE=128 / 256):
// a workgroup spends c us per 32-deep K-tile and e us in its prologue + epilogue (e is dominated by the result bytes: halve it
// for a bf16-only destination); `slots` workgroups run at a time; a split costs the partial planes' round trip at ~3 TB/s + a launch.
struct Plan { int big; int splits; double cost; };            // big: 0 = 128 x 128 ring, 1 = 256 x 256 ring, 2 = 256 x 256 eight-phase (gemm_bf16_p8.h)
// Round 6: the eight-phase form.  Measured (tools/p8_probe.py, tools/gemm_bf16_sweep.py): 1.37 us per 64-deep K-tile and workgroup with all CUs
// busy (0.685 per 32 deep; 1.56 when B is K-major: its sub-tile images are 64-byte row pieces), 20 us per tile round outside the K loop
// with an fp32 destination, 17 with bf16 only -- and 10 us once per launch that only shows INSIDE a train step (tools/gemm_insitu_ab.py): a
// lone 128 KiB workgroup per CU starts on operands the step left in HBM, and one round of K <= 1024 (the GCN products) then runs slower than the
// ring forms although the warm stand-alone sweep says the opposite.  With the LDS-transposed epilogue (for_each_quad_lds) the per-round cost fell to
// 14 / 13 us (16384 x 1024 x 1024: 47 -> 41 us with an fp32 result, the store phase now at the HBM write rate), which moves the 9472-row GCN
// products and the heads of the 16640-row ones to this form.
inline Plan plan_for(int M, int N, int K, bool may_split, size_t ws_bytes, bool out32 = true, int nprob = 1, bool allow_p8 = false, bool b_km_only = false) {
    const int kt = (int)subgc::cdiv(K, BK);
    Plan best{0, 1, 1e30};
    for (int big = 0; big < (allow_p8 ? 3 : 2); ++big) {
        const int64_t tiles = nprob * (big ? subgc::cdiv(M, 256) * subgc::cdiv(N, 256) : subgc::cdiv(M, 128) * subgc::cdiv(N, 128));
        const int slots = big ? 256 : 512;
        const double c = big == 2 ? (b_km_only ? 0.78 : 0.685) : big ? 0.85 : 0.67;
        const double e = big == 2 ? (out32 ? 14.0 : 13.0) : (big ? 24.0 * (out32 ? 1.0 : 0.8) : 12.0 * (out32 ? 1.0 : 0.55));    // eight-phase: 20 / 17 before its results went through LDS
        const double once = big == 2 ? 6.0 : 0.0;
        for (int s = 1; s <= (big == 2 ? 15 : 8); ++s) {        // 15: what SUBGC_GEMM_SPLITS can name; ~1024-deep parts of the GCN weight gradients (16 tiles, K = 16640) fill the chip
            if (s > 1 && (!may_split || (kt + s - 1) / s < 12 || (size_t)nprob * s * M * N * sizeof(float) > ws_bytes)) break;
            const int per = (kt + s - 1) / s;
            const int64_t rounds = (tiles * s + slots - 1) / slots;
            const double tail = s > 1 ? 3.0 + (double)nprob * (s + 1) * M * N * 4.0 / 3.0e6 : 0.0;       // us: planes written + read back
            const double cost = rounds * (per * c + (s > 1 ? e * 0.7 : e)) + tail + once;
            if (cost < best.cost - 1e-9) best = Plan{big, s, cost};
        }
    }
    return best;
}

template <typename KernelT>
int raise_lds(KernelT kernel, size_t lds, uint64_t& done) {
    if (lds <= 64 * 1024) return SUBGC_OK;            // > 64 KiB of dynamic LDS needs the opt-in once per kernel AND DEVICE (bit = device index)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done & bit) return SUBGC_OK;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        subgc::set_error("gemm_bf16: cannot raise dynamic LDS limit to %zu", lds);
        return SUBGC_ELAUNCH;
    }
    done |= bit;
    return SUBGC_OK;
}

template <typename G, bool A_KM, bool B_KM>
int launch(const Args& a, float* ws, int splits, bool partials_only, hipStream_t s) {
    const int64_t tiles = a.nprob * subgc::cdiv(a.M, G::TBM) * subgc::cdiv(a.N, G::TBN);          // pair launches: both problems' tiles
    const int kt = (int)subgc::cdiv(a.K, BK);
    static uint64_t attr_a = 0, attr_b = 0;
    bool with_cs = false;
    if constexpr (A_KM && G::TBM == 256) with_cs = a.cs_out != nullptr && !partials_only;
    if (splits <= 1) {
        if constexpr (A_KM && G::TBM == 256) {
            if (with_cs) {
                static uint64_t attr_c = 0;
                if (int rc = raise_lds(gemm_bf16_kernel<G, A_KM, B_KM, true>, G::LDS, attr_c)) return rc;
                hipLaunchKernelGGL((gemm_bf16_kernel<G, A_KM, B_KM, true>), dim3((unsigned)tiles), dim3(G::NT), G::LDS, s, a);
                return subgc::check_launch("subgc_gemm_bf16_wgrad");
            }
        }
        if (int rc = raise_lds(gemm_bf16_kernel<G, A_KM, B_KM>, G::LDS, attr_a)) return rc;
        hipLaunchKernelGGL((gemm_bf16_kernel<G, A_KM, B_KM>), dim3((unsigned)tiles), dim3(G::NT), G::LDS, s, a);
        return subgc::check_launch("subgc_gemm_bf16");
    }
    bool launched = false;
    if constexpr (A_KM && G::TBM == 256) {
        if (with_cs) {
            static uint64_t attr_d = 0;
            if (int rc = raise_lds(gemm_bf16_splitk_kernel<G, A_KM, B_KM, true>, G::LDS, attr_d)) return rc;
            hipLaunchKernelGGL((gemm_bf16_splitk_kernel<G, A_KM, B_KM, true>), dim3((unsigned)(tiles * splits)), dim3(G::NT), G::LDS, s, a, ws, splits,
                               (kt + splits - 1) / splits);
            launched = true;
        }
    }
    if (!launched) {
        if (int rc = raise_lds(gemm_bf16_splitk_kernel<G, A_KM, B_KM>, G::LDS, attr_b)) return rc;
        hipLaunchKernelGGL((gemm_bf16_splitk_kernel<G, A_KM, B_KM>), dim3((unsigned)(tiles * splits)), dim3(G::NT), G::LDS, s, a, ws, splits,
                           (kt + splits - 1) / splits);
    }
    if (partials_only) return subgc::check_launch("subgc_gemm_bf16(split-K, partials)");
    const int64_t n = (int64_t)a.M * a.N / 4;
    hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048), (unsigned)a.nprob), dim3(256), 0, s, (const float*)ws,
                       splits, a.M, a.N, a.C32, a.ldc32, a.C16, a.ldc16, a.bias, (a.flags & SUBGC_GEMM_ACCUM) ? 1 : 0,
                       (a.flags & SUBGC_GEMM_RELU) ? 1 : 0, with_cs ? (const float*)a.cs_part : nullptr, a.cs_out, a.cs_accum, a.C32b, a.C16b, a.bias2, a.keep, a.keep_scale);
    return subgc::check_launch("subgc_gemm_bf16(split-K)");
}

template <bool A_KM, bool B_KM>
int launch_p8(const Args& a, float* ws, int splits, bool partials_only, hipStream_t s) {
    const int64_t tiles = a.nprob * subgc::cdiv(a.M, 256) * subgc::cdiv(a.N, 256);
    const int kt = (int)subgc::cdiv(a.K, p8::KT);
    static uint64_t attr_a = 0, attr_b = 0;
    bool with_cs = false;
    if constexpr (A_KM) with_cs = a.cs_out != nullptr && !partials_only;
    if (splits <= 1) {
        if constexpr (A_KM) {
            if (with_cs) {
                static uint64_t attr_c = 0;
                if (int rc = raise_lds(gemm_bf16_p8_kernel<A_KM, B_KM, true>, p8::LDS_BYTES, attr_c)) return rc;
                hipLaunchKernelGGL((gemm_bf16_p8_kernel<A_KM, B_KM, true>), dim3((unsigned)tiles), dim3(p8::NT), p8::LDS_BYTES, s, a);
                return subgc::check_launch("subgc_gemm_bf16_wgrad(p8)");
            }
        }
        if (int rc = raise_lds(gemm_bf16_p8_kernel<A_KM, B_KM>, p8::LDS_BYTES, attr_a)) return rc;
        hipLaunchKernelGGL((gemm_bf16_p8_kernel<A_KM, B_KM>), dim3((unsigned)tiles), dim3(p8::NT), p8::LDS_BYTES, s, a);
        return subgc::check_launch("subgc_gemm_bf16(p8)");
    }
    bool launched = false;
    if constexpr (A_KM) {
        if (with_cs) {
            static uint64_t attr_d = 0;
            if (int rc = raise_lds(gemm_bf16_p8_splitk_kernel<A_KM, B_KM, true>, p8::LDS_BYTES, attr_d)) return rc;
            hipLaunchKernelGGL((gemm_bf16_p8_splitk_kernel<A_KM, B_KM, true>), dim3((unsigned)(tiles * splits)), dim3(p8::NT), p8::LDS_BYTES, s, a, ws, splits,
                               (kt + splits - 1) / splits);
            launched = true;
        }
    }
    if (!launched) {
        if (int rc = raise_lds(gemm_bf16_p8_splitk_kernel<A_KM, B_KM>, p8::LDS_BYTES, attr_b)) return rc;
        hipLaunchKernelGGL((gemm_bf16_p8_splitk_kernel<A_KM, B_KM>), dim3((unsigned)(tiles * splits)), dim3(p8::NT), p8::LDS_BYTES, s, a, ws, splits,
                           (kt + splits - 1) / splits);
    }
    if (partials_only) return subgc::check_launch("subgc_gemm_bf16(p8, split-K, partials)");
    const int64_t n = (int64_t)a.M * a.N / 4;
    hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048), (unsigned)a.nprob), dim3(256), 0, s, (const float*)ws,
                       splits, a.M, a.N, a.C32, a.ldc32, a.C16, a.ldc16, a.bias, (a.flags & SUBGC_GEMM_ACCUM) ? 1 : 0,
                       (a.flags & SUBGC_GEMM_RELU) ? 1 : 0, with_cs ? (const float*)a.cs_part : nullptr, a.cs_out, a.cs_accum, a.C32b, a.C16b, a.bias2, a.keep, a.keep_scale);
    return subgc::check_launch("subgc_gemm_bf16(p8, split-K)");
}

// the eight-phase loop addresses its operands with 32-bit byte offsets; a K-contiguous operand is masked in 16-byte chunks (K % 8 == 0), a
// K-major one in whole k-rows (any K)
inline bool p8_ok(const Args& a, bool a_km, bool b_km) {
    const int64_t ea = (a_km ? (int64_t)a.K : (int64_t)a.M) * a.lda * 2, eb = (b_km ? (int64_t)a.K : (int64_t)a.N) * a.ldb * 2;
    return ((a_km && b_km) || a.K % 8 == 0) && ea < 0x7ff00000ll && eb < 0x7ff00000ll;
}

template <bool A_KM, bool B_KM>
int run(const Args& a, float* ws, size_t ws_bytes, hipStream_t s, bool partials_only, int* splits_out, bool may_cut_rows = true) {
    // (a dropout mask rides the reduce pass of a single-problem launch: the 320-row fc projection of the Flickr shape, K = 4096, is 24 tiles)
    const bool plain = !a.add && (!a.keep || (a.nprob == 1 && aligned4(a.keep))) && (!a.m_dev || A_KM) && a.N % 4 == 0 && (!a.C32 || (a.ldc32 % 4 == 0 && aligned16(a.C32))) &&
                       (!a.C16 || (a.ldc16 % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C16) & 7) == 0)) && (!a.bias || aligned16(a.bias)) &&
                       (a.nprob == 1 || ((!a.C32b || aligned16(a.C32b)) && (!a.C16b || (reinterpret_cast<uintptr_t>(a.C16b) & 7) == 0) &&
                                         (!a.bias2 || aligned16(a.bias2))));
    const int force = (a.flags & SUBGC_GEMM_TILE128) ? 128 : (a.flags & SUBGC_GEMM_TILE256) ? 256 : 0;      // measurement scripts: tile A/B timing
    // Row cut (round 4).  A tile count a few tiles above whole rounds of the 256 CUs costs a whole extra round: Full_GC_Kar's 16 640 relation
    // rows x 1024 columns are 65 x 4 = 260 tiles of 256 x 256 -- 1.016 rounds, timed like two (86 us against 45 for 16 384 rows).  When the
    // rows beyond the last whole round are few, the whole rounds go out as one launch and the remaining rows as a second one with its own
    // plan (small tiles, split K): two disjoint row ranges of the same product.  A stored row-major (not K-major), no device-side row count.
    if (may_cut_rows && !partials_only && !A_KM && !a.m_dev && force == 0 && SUBGC_GEMM_SPLITS_OF(a.flags) == 0 && !(a.flags & SUBGC_GEMM_NO_ROW_CUT)) {
        const int64_t tn = subgc::cdiv(a.N, 256) * a.nprob, tm = subgc::cdiv(a.M, 256);       // a pair: both problems' tile columns share a round
        if (tn <= 256 && 256 % tn == 0) {
            const int64_t per_round = 256 / tn;                                   // row tiles (of each problem) in one full round
            const int64_t whole = tm / per_round * per_round;                     // row tiles in whole rounds
            const int64_t M1 = whole * 256, rest = a.M - M1;
            if (whole >= per_round && rest > 0 && rest <= 2 * 256 && rest * 8 <= M1) {
                Args head = a, tail = a;
                head.M = (int)M1;
                tail.M = (int)rest;
                tail.A = a.A + M1 * a.lda;
                if (a.C32) tail.C32 = a.C32 + M1 * a.ldc32;
                if (a.C16) tail.C16 = a.C16 + M1 * a.ldc16;
                if (a.add) tail.add = a.add + M1 * a.ldadd;
                if (a.keep) tail.keep = a.keep + M1 * (int64_t)a.N;
                if (a.nprob == 2) {
                    tail.A2 = a.A2 + M1 * a.lda;
                    if (a.C32b) tail.C32b = a.C32b + M1 * a.ldc32;
                    if (a.C16b) tail.C16b = a.C16b + M1 * a.ldc16;
                }
                if (int rc = run<A_KM, B_KM>(head, ws, ws_bytes, s, false, splits_out, false)) return rc;
                return run<A_KM, B_KM>(tail, ws, ws_bytes, s, false, nullptr, false);
            }
        }
    }
    const bool p8_able = p8_ok(a, A_KM, B_KM) && !(a.flags & SUBGC_GEMM_NO_P8);
    Plan pl = plan_for(a.M, a.N, a.K, ws != nullptr && plain, ws_bytes, a.C32 != nullptr, a.nprob, p8_able, B_KM && !A_KM);
    if (partials_only) {                                        // the LSTM cell kernel adds row-major planes: 128x128 split form only
        pl = Plan{0, 1, 0.0};
        const int64_t tiles = subgc::cdiv(a.M, 128) * subgc::cdiv(a.N, 128);
        const int kt = (int)subgc::cdiv(a.K, BK);
        double best = 1e30;
        for (int sp = 2; sp <= 8 && (kt + sp - 1) / sp >= 12 && (size_t)sp * a.M * a.N * sizeof(float) <= ws_bytes; ++sp) {
            const double c = (double)((tiles * sp + 511) / 512) * ((kt + sp - 1) / sp + 8.0) + 0.8 * sp;
            if (c < best) { best = c; pl.splits = sp; }
        }
        if (pl.splits <= 1 || tiles >= 448) return -100;
        // Round 6: from 1024 rows up (Full_GC_Kar's first steps: 5 x 16 tiles of 256 x 256) the planes come from the eight-phase form, K parts
        // chosen to fill whole rounds of the 256 CUs -- in situ (A/B in one job, two repeats) the GEMM family 8.82 -> 8.61 ms per step, the step
        // 15.10 -> 15.03 (one more plane for the cell kernels to add eats two thirds of it); below 1024 rows: no gain, the 128 x 128 form stays.
        if (p8_able && a.M >= 1024) {
            const int64_t t256 = subgc::cdiv(a.M, 256) * subgc::cdiv(a.N, 256);
            const int kt64 = (int)subgc::cdiv(a.K, 64);
            double bestc = 1e30;
            int bs = 0;
            for (int sp = 2; sp <= 8 && (kt64 + sp - 1) / sp >= 6 && (size_t)sp * a.M * a.N * sizeof(float) <= ws_bytes; ++sp) {
                const double c = (double)((t256 * sp + 255) / 256) * ((kt64 + sp - 1) / sp * 1.4 + 14.0) + 0.8 * sp;
                if (c < bestc) { bestc = c; bs = sp; }
            }
            if (bs > 1) { pl.big = 2; pl.splits = bs; }
        }
    } else if (force == 128 || force == 256) {
        pl.big = force == 256;
        if (pl.big) pl.splits = 1;
    } else if (a.flags & SUBGC_GEMM_TILE_P8) {
        SUBGC_REQUIRE(p8_ok(a, A_KM, B_KM), "gemm_bf16: the eight-phase form needs K %% 8 == 0 (K-contiguous operands) and operands below 2 GiB");
        pl.big = 2;
        pl.splits = 1;
    }
    if (const int fs = SUBGC_GEMM_SPLITS_OF(a.flags); fs > 0 && !partials_only) {        // measurement scripts: K parts of this call
        SUBGC_REQUIRE(fs == 1 || (ws != nullptr && plain && (size_t)a.nprob * fs * a.M * a.N * sizeof(float) <= ws_bytes), "gemm_bf16: forced split needs the plain epilogue and %d planes of workspace", fs);
        pl.splits = fs;
    }
    if (splits_out) *splits_out = pl.splits;
    if constexpr (A_KM) {
        // Column sums ride the 256 x 256 workgroups for free (measured: 254 us with, 266 us without on 14000 x 4000 x 2000); in the 128 x 128
        // geometry the extra read + adds show (a quarter of the workgroups own tile column 0 at N = 512: 46 + 7 us against 34 + 5 + 9.6 for
        // the separate column-sum pass), so those plans run the two passes -- still one call.
        if (a.cs_out && pl.big == 0) {
            Args b = a;
            b.cs_out = nullptr;
            if (int rc = launch<G128, A_KM, B_KM>(b, ws, pl.splits, partials_only, s)) return rc;
            return subgc_colsum_bf16(a.A, a.lda, a.K, a.M, a.cs_out, a.cs_accum, a.m_dev, ws, ws_bytes, s);   // the planes are consumed: ws is free (stream order)
        }
    }
    if (pl.big == 2) return launch_p8<A_KM, B_KM>(a, ws, pl.splits, partials_only, s);
    return pl.big ? launch<G256, A_KM, B_KM>(a, ws, pl.splits, partials_only, s) : launch<G128, A_KM, B_KM>(a, ws, pl.splits, partials_only, s);
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
    const Plan pl = plan_for(M, N, K, true, (size_t)15 * M * N * sizeof(float), true, 1, K % 8 == 0);
    *bytes = pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
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

// Weight gradient and bias gradient in one call (see subgc_gemm_f32_wgrad): dW[M,N] (+)= dY^T X, db[M] (+)= column sums of dY; dY [K, M]
// and X [K, N] stored bf16.  Needs the split-K plan's preconditions for its partial sums (N % 4 == 0 is checked by the plan itself).
SUBGC_API int subgc_gemm_bf16_wgrad(int M, int N, int K, const uint16_t* dY, int64_t lddy, const uint16_t* X, int64_t ldx, float* dW, int64_t lddw,
                                    float* db, int flags, int db_accumulate, const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream) {
    if (M == 0) return SUBGC_OK;
    SUBGC_REQUIRE(M > 0 && N >= 0 && K >= 0, "gemm_bf16_wgrad: negative size M=%d N=%d K=%d", M, N, K);
    SUBGC_REQUIRE(db != nullptr && (N == 0 || dW != nullptr), "gemm_bf16_wgrad: null destination");
    SUBGC_REQUIRE(!workspace || aligned16(workspace), "gemm_bf16_wgrad: workspace must be 16-byte aligned");
    if (K == 0) return subgc::wgrad_no_rows(dW, lddw, db, M, N, (flags & SUBGC_GEMM_ACCUM) != 0, db_accumulate != 0, (hipStream_t)stream);
    SUBGC_REQUIRE(dY != nullptr, "gemm_bf16_wgrad: null operand");
    if (N == 0) return subgc_colsum_bf16(dY, lddy, K, M, db, db_accumulate, m_dev, workspace, ws_bytes, stream);
    if (int rc = check(1, 0, M, N, K, dY, lddy, X, ldx)) return rc;
    SUBGC_REQUIRE(dW && lddw >= N, "gemm_bf16_wgrad: no / too narrow destination");
    Args a{dY, X, dW, nullptr, nullptr, nullptr, nullptr, m_dev, lddy, ldx, lddw, 0, 0, M, N, K, flags, 1.f};
    a.cs_out = db; a.cs_accum = db_accumulate ? 1 : 0;
    float* ws = static_cast<float*>(workspace);
    const size_t cs_bytes = ((size_t)15 * M * sizeof(float) + 15) & ~(size_t)15;
    if (ws && ws_bytes > cs_bytes) {                            // the split-K form's partial sums live behind its planes
        ws_bytes = (ws_bytes - cs_bytes) & ~(size_t)15;
        a.cs_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_bytes);
    } else { ws = nullptr; ws_bytes = 0; }
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    return run<true, true>(a, ws, ws_bytes, s, false, nullptr);
}

// Two products of the SAME shape, layout and epilogue in one launch (the two units of a GCN pair: d(H) halves, fc_rgt weight gradients): the
// grid's first half works on (A1, B1 -> C1), the second on (A2, B2 -> C2); tile choice, K parts and the row cut are planned for both
// together, so two half-filling launches become one that fills the chip.  Epilogue: bias / ReLU / ACCUM, fp32 and / or bf16 destinations.
SUBGC_API int subgc_gemm_bf16_pair(int transA, int transB, int M, int N, int K, const uint16_t* A1, const uint16_t* A2, int64_t lda,
                                   const uint16_t* B1, const uint16_t* B2, int64_t ldb, float* C32_1, float* C32_2, int64_t ldc32, uint16_t* C16_1,
                                   uint16_t* C16_2, int64_t ldc16, const float* bias1, const float* bias2, int flags, void* workspace, size_t ws_bytes,
                                   void* stream) {
    if (M == 0 || N == 0) return SUBGC_OK;
    if (int rc = check(transA, transB, M, N, K, A1, lda, B1, ldb)) return rc;
    if (int rc = check(transA, transB, M, N, K, A2, lda, B2, ldb)) return rc;
    SUBGC_REQUIRE((C32_1 || C16_1) && (C32_1 != nullptr) == (C32_2 != nullptr) && (C16_1 != nullptr) == (C16_2 != nullptr), "gemm_bf16_pair: the two problems need the same kinds of destination");
    SUBGC_REQUIRE((!C32_1 || ldc32 >= N) && (!C16_1 || ldc16 >= N), "gemm_bf16_pair: destination leading dimension too small");
    SUBGC_REQUIRE(!(flags & SUBGC_GEMM_ACCUM) || C32_1, "gemm_bf16_pair: accumulate needs the fp32 destination");
    SUBGC_REQUIRE((bias1 != nullptr) == (bias2 != nullptr), "gemm_bf16_pair: bias for both problems or for none");
    SUBGC_REQUIRE(!workspace || aligned16(workspace), "gemm_bf16_pair: workspace must be 16-byte aligned");
    Args a{A1, B1, C32_1, C16_1, bias1, nullptr, nullptr, nullptr, lda, ldb, ldc32, ldc16, 0, M, N, K, flags, 1.f};
    a.A2 = A2; a.B2 = B2; a.C32b = C32_2; a.C16b = C16_2; a.bias2 = bias2; a.nprob = 2;
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 4.0 * M * (double)N * K);
    float* ws = static_cast<float*>(workspace);
    if (!transA && transB) return run<false, false>(a, ws, ws_bytes, s, false, nullptr);
    if (!transA && !transB) return run<false, true>(a, ws, ws_bytes, s, false, nullptr);
    return run<true, true>(a, ws, ws_bytes, s, false, nullptr);
}

// bf16 operands -> `*n_planes` fp32 partial planes planes[q][M][N] whose SUM is the product (see subgc_gemm_f32_planes)
SUBGC_API int subgc_gemm_bf16_planes(int transA, int transB, int M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                                     float* planes, size_t planes_bytes, int* n_planes, void* stream) {
    SUBGC_REQUIRE(M > 0 && N > 0 && K > 0 && n_planes, "gemm_bf16_planes: bad sizes");
    if (int rc = check(transA, transB, M, N, K, A, lda, B, ldb)) return rc;
    SUBGC_REQUIRE(planes && aligned16(planes) && planes_bytes >= (size_t)M * N * sizeof(float) && N % 4 == 0, "gemm_bf16_planes: plane buffer / N %% 4");
    Args a{A, B, planes, nullptr, nullptr, nullptr, nullptr, nullptr, lda, ldb, N, 0, 0, M, N, K, 0, 1.f};
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    int sp = 1, rc;
    if (!transA && transB) rc = run<false, false>(a, planes, planes_bytes, s, true, &sp);
    else if (!transA && !transB) rc = run<false, true>(a, planes, planes_bytes, s, true, &sp);
    else rc = run<true, true>(a, planes, planes_bytes, s, true, &sp);
    if (rc != -100) { *n_planes = sp; return rc; }
    *n_planes = 1;                                                  // the dispatch would not split this shape: the plain kernel writes plane 0
    if (!transA && transB) return run<false, false>(a, nullptr, 0, s, false, nullptr);
    if (!transA && !transB) return run<false, true>(a, nullptr, 0, s, false, nullptr);
    return run<true, true>(a, nullptr, 0, s, false, nullptr);
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
