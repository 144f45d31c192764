// The 256 x 256 x 64 "eight-phase" main loop of the bf16-operand GEMM (included by gemm_bf16.hip inside its anonymous namespace).
//
// Why a second main loop.  The ring loop of gemm_bf16.hip (BK = 32, 16 waves of 64 x 64, ONE barrier per K-tile) feeds a CU at ~24 B/clk
// whatever its depth: a 32-deep stage of a K-contiguous operand is 64 bytes per row, i.e. every 128-byte line of A and B is asked for in two
// halves by two different K-tiles, and a K-tile is as late as its slowest half line.  Here a stage is 64 deep (whole lines), the workgroup is 8
// waves of 128 x 64 (two per SIMD) that run HALF A PHASE APART, and a K-tile is cut into four phases -- one 64 x 32 quadrant of the wave's
// accumulators over the whole 64-deep stage each -- so that while one wave of a SIMD issues its 8 MFMAs (256 cycles of the matrix pipe) the other
// reads its next fragments and issues its share of the LDS-DMA; counted vmcnt waits sit at the last phase of a K-tile only and nothing is ever
// drained inside the loop.
//
// LDS (128 KiB): two buffers (E = even K-tiles, O = odd) x four sub-tile images x 16 KiB.  A sub-tile image holds the rows ONE phase reads:
//   A0 = tile rows {0..63, 128..191} (rows 0..63 of either wave row), A1 = the other 128; B0 = columns {0..31, 64..95, 128..159, 192..223}
//   (columns 0..31 of every wave column), B1 = the rest.  K-contiguous operand: [128 rows][64 k] (128-byte rows, 16-byte chunk c of row r at
//   c ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group fall on 16 different 16-byte slots of the 256-byte bank row); K-major operand:
//   [64 k][128 rows] read with ds_read_b64_tr_b16 exactly like the ring loop's 128-row images.  The DMA writes a wave instruction's 1 KiB
//   linearly, so both swizzles are applied to the SOURCE address of the lane.
// Schedule of one iteration = two K-tiles = phases 0..7 (buffer = P / 4):
//   reads   P%4 = 0: B0 (4 ds_read_b128) then A0 (8);  1: B1 (4);  2: A1 (8);  3: none
//   MFMA    P%4 = 0: A0 x B0;  1: A0 x B1;  2: A1 x B1;  3: A1 x B0 (B0 is kept in registers for the whole K-tile)
//   DMA     one sub-tile image per phase, three in flight:  P0 -> O.A1 (tile t+1) | P1 E.B0, P2 E.A0, P3 E.B1, P4 E.A1 (tile t+2) |
//           P5 O.B0, P6 O.A0, P7 O.B1 (tile t+3)
//   waits   vmcnt(10) in every phase but P%4 = 2: the image the NEXT phase reads was issued six sub-tiles ago, so five younger ones (80 KiB
//           per CU) stay in flight; an image has five phases (~1.7 us) to land
// Every image is re-staged two phases after the phase that read it (B0: one phase after, its four reads are retired by the lgkmcnt in front
// of P0's first barrier), and read one phase after the wait that retires it; with the two wave groups half a phase apart both rules hold for
// either group (the later group's reads are issued before the barrier that the earlier group's next DMA follows).
//
// Sources go through a buffer descriptor (buffer_load_dwordx4 ... lds): one 32-bit VGPR offset per (sub-tile, instruction), the K position in
// the SCALAR offset -- no vector address arithmetic in the loop -- and a lane whose chunk lies outside the operand (row >= M / N, k >= K, a
// K-tile past the end of this unit's range) is given an out-of-range offset: the hardware writes ZEROS for it.  That is the whole edge and tail
// handling: the prologue / epilogue of the pipeline issue ordinary (fully masked) instructions, so the vmcnt arithmetic is the same everywhere.
#pragma once

namespace p8 {

constexpr int KT = 64;                                         // K per LDS buffer
constexpr int SUBB = 128 * KT * 2;                             // bytes of one sub-tile image
constexpr int A_REGION = 0, B_REGION = 4 * SUBB;               // [E.x0 | E.x1 | O.x0 | O.x1] per operand: every image within a 16-bit DS offset of its region
constexpr size_t LDS_BYTES = 8 * (size_t)SUBB;
constexpr uint32_t OOB = 0x80000000u;                          // >= num_records of the descriptors below
constexpr int NT = 512;
__host__ __device__ constexpr int sub_off(int buf, int sub) { return (buf * 2 + sub) * SUBB; }

// local row (0..127) of sub-tile image `sub` -> row of the 256-row tile
template <bool IS_A>
__device__ __forceinline__ int tile_row(int lr, int sub) {
    return IS_A ? ((lr >> 6) * 128 + sub * 64 + (lr & 63)) : ((lr >> 5) * 64 + sub * 32 + (lr & 31));
}

template <bool KM, bool IS_A>
struct Feed {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[2][2];                                       // [sub][instruction]: byte offset of this lane's 16 bytes at k0 = 0 (OOB: row outside)
    int kq[2];                                                 // k (relative to the tile's k0) of this lane's chunk / k-row
    uint32_t kmul;                                             // bytes per k of the scalar offset

    __device__ __forceinline__ void init(const uint16_t* base, int64_t ld, int r0, int nrows) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base), 0, 0x7ffffff0, 0x00020000);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        kmul = KM ? (uint32_t)ld * 2u : 2u;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int byte = (wave * 2 + v) * 1024 + lane * 16;
            if (!KM) {
                const int lr = byte >> 7, pc = (byte >> 4) & 7, c = pc ^ ((lr >> 1) & 7);
                kq[v] = c * 8;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int row = r0 + tile_row<IS_A>(lr, s);
                    voff[s][v] = row < nrows ? (uint32_t)((int64_t)row * ld * 2 + c * 16) : OOB;
                }
            } else {
                const int k = byte >> 8, pc = (byte >> 4) & 15, c = pc ^ ((k & 3) << 2);
                kq[v] = k;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int col = r0 + tile_row<IS_A>(c * 8, s);
                    voff[s][v] = col < nrows ? (uint32_t)((int64_t)k * ld * 2 + col * 2) : OOB;
                }
            }
        }
    }
    // one sub-tile image of the K-tile at k0 (this wave's two 1 KiB pieces); kend = end of the K range this workgroup sums over
    template <int BUF, int SUB>
    __device__ __forceinline__ void issue(unsigned char* smem, int k0, int kend) const {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const uint32_t soff = (uint32_t)k0 * kmul;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const uint32_t vo = (k0 + kq[v] < kend) ? voff[SUB][v] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)SUBGC_LDS(smem + (IS_A ? A_REGION : B_REGION) + sub_off(BUF, SUB) + (wave * 2 + v) * 1024),
                                                     16, vo, soff, 0, 0);
        }
    }
};

// Fragment registers of one sub-tile: NF 32-row fragments x 4 sixteen-deep steps.  K-contiguous image: one ds_read_b128 each; K-major image:
// two ds_read_b64_tr_b16.  All reads are issued from asm (the compiler's wait-count pass must not see them: it would drain the DMA queue) and
// become usable through wait(), which threads the registers through an lgkmcnt(0).
template <bool KM, int NF>
struct Frags;

template <int NF>
struct Frags<false, NF> {
    u32x4 r[NF][4];
    uint32_t ad[NF][4];
    __device__ __forceinline__ void init(uint32_t region, int r0, int lane) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int lr = r0 + f * 32 + (lane & 31), x = (lr >> 1) & 7, hi = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ad[f][ks] = region + lr * 128 + (((ks * 2 + hi) ^ x) << 4);
        }
    }
    template <int OFF>
    __device__ __forceinline__ void read() {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[f][ks]) : "v"(ad[f][ks]), "n"(OFF));
    }
    __device__ __forceinline__ bf16x8 get(int f, int ks) const { return __builtin_bit_cast(bf16x8, r[f][ks]); }
    static constexpr int NREADS = NF * 4;
};

template <int NF>
struct Frags<true, NF> {
    unsigned long long lo[NF][4], hi[NF][4];
    uint32_t ad[NF];
    __device__ __forceinline__ void init(uint32_t region, int r0, int lane) {
        const int g = lane >> 4, i = lane & 15, k = (g >> 1) * 8 + (i >> 2);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int c = ((r0 + f * 32 + (g & 1) * 16) >> 3) + ((i & 3) >> 1);
            ad[f] = region + k * 256 + ((c ^ ((k & 3) << 2)) << 4) + (i & 1) * 8;
        }
    }
    template <int OFF>
    __device__ __forceinline__ void read() {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // hipcc wants the immediate as a literal: one statement per k-step
                if (ks == 0) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[f][0]) : "v"(ad[f]), "n"(OFF)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[f][0]) : "v"(ad[f]), "n"(OFF + 1024)); }
                if (ks == 1) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[f][1]) : "v"(ad[f]), "n"(OFF + 4096)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[f][1]) : "v"(ad[f]), "n"(OFF + 4096 + 1024)); }
                if (ks == 2) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[f][2]) : "v"(ad[f]), "n"(OFF + 8192)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[f][2]) : "v"(ad[f]), "n"(OFF + 8192 + 1024)); }
                if (ks == 3) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[f][3]) : "v"(ad[f]), "n"(OFF + 12288)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[f][3]) : "v"(ad[f]), "n"(OFF + 12288 + 1024)); }
            }
    }
    __device__ __forceinline__ bf16x8 get(int f, int ks) const { return tr_join(lo[f][ks], hi[f][ks]); }
    static constexpr int NREADS = NF * 8;
};

// lgkmcnt(0) with the fragment registers threaded through, so that no consumer is scheduled above it
__device__ __forceinline__ void wait_frags(Frags<false, 1>& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.r[0][0]), "+v"(x.r[0][1]), "+v"(x.r[0][2]), "+v"(x.r[0][3]));
}
__device__ __forceinline__ void wait_frags(Frags<false, 2>& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.r[0][0]), "+v"(x.r[0][1]), "+v"(x.r[0][2]), "+v"(x.r[0][3]), "+v"(x.r[1][0]), "+v"(x.r[1][1]), "+v"(x.r[1][2]), "+v"(x.r[1][3]));
}
__device__ __forceinline__ void wait_frags(Frags<true, 1>& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.lo[0][0]), "+v"(x.lo[0][1]), "+v"(x.lo[0][2]), "+v"(x.lo[0][3]), "+v"(x.hi[0][0]), "+v"(x.hi[0][1]), "+v"(x.hi[0][2]), "+v"(x.hi[0][3]));
}
__device__ __forceinline__ void wait_frags(Frags<true, 2>& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.lo[0][0]), "+v"(x.lo[0][1]), "+v"(x.lo[0][2]), "+v"(x.lo[0][3]), "+v"(x.hi[0][0]), "+v"(x.hi[0][1]), "+v"(x.hi[0][2]), "+v"(x.hi[0][3]),
                 "+v"(x.lo[1][0]), "+v"(x.lo[1][1]), "+v"(x.lo[1][2]), "+v"(x.lo[1][3]), "+v"(x.hi[1][0]), "+v"(x.hi[1][1]), "+v"(x.hi[1][2]), "+v"(x.hi[1][3]));
}

// the two 16-byte pieces (k-rows t / 16 and t / 16 + 32, chunk t % 16) a thread adds to its column sums per sub-tile image of a K-major A
struct CsRegs {
    u32x4 q[2];
    uint32_t ad;
    __device__ __forceinline__ void init(uint32_t region) {
        const int kk = (int)threadIdx.x >> 4, cc = (int)threadIdx.x & 15;
        ad = region + kk * 256 + ((cc ^ ((kk & 3) << 2)) << 4);
    }
    template <int OFF>
    __device__ __forceinline__ void read() {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[0]) : "v"(ad), "n"(OFF));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[1]) : "v"(ad), "n"(OFF + 32 * 256));
    }
    // behind an lgkmcnt(0) that the fragment wait of the same phase issued (LDS returns in order)
    __device__ __forceinline__ void add(float (&dst)[8]) {
        asm volatile("" : "+v"(q[0]), "+v"(q[1]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dst[2 * d] += __uint_as_float(q[j][d] << 16);
                dst[2 * d + 1] += __uint_as_float(q[j][d] & 0xffff0000u);
            }
    }
};

template <int N>
__device__ __forceinline__ void wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// acc[a][b]: 32 x 32 tiles of C^T, a = 2 sub_A + f (rows), b = sub_B (columns) -- the layout for_each_quad<Geo<256, 256, 4, 2>> stores.
// K range of this workgroup: K-tiles kt0 .. kt1-1 (64 deep), clipped at K.
// CS (K-major A only; subgc_gemm_bf16_wgrad): the workgroups of tile column 0 (do_cs, workgroup-uniform) also sum the columns of their A tiles
// from the landed images: thread t reads chunk t % 16 (8 columns) of k-rows t / 16 and t / 16 + 32 of A0 in phase 0 and of A1 in phase 2 -- the
// phases whose fragment reads use the same image, so the re-staging distance is the fragments' -- and adds them behind that phase's barrier;
// the unpack + add VALU work has no dependence on the MFMAs around it.  cs[img][e]: column 8 (t % 16) + e of image img over this thread's k-rows.
template <bool A_KM, bool B_KM, bool CS = false>
__device__ __forceinline__ void mainloop(const Args& p, unsigned char* smem, int M, int K, int m0, int n0, int kt0, int kt1, f32x16 (&acc)[4][2],
                                         float (*cs)[8] = nullptr, bool do_cs = false) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!CS || A_KM, "column sums are read from the K-major image of A");
    if constexpr (CS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[i][e] = 0.f;
    }
    const int ntl = kt1 - kt0;
    if (ntl <= 0) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int kend = min(K, kt1 * KT);
    Feed<A_KM, true> da;
    Feed<B_KM, false> db;
    da.init(p.A, p.lda, m0, M);
    db.init(p.B, p.ldb, n0, p.N);
    const uint32_t lds0 = (uint32_t)(uintptr_t)SUBGC_LDS(smem);
    Frags<A_KM, 2> fa;
    Frags<B_KM, 1> fb0, fb1;
    fa.init(lds0 + A_REGION, wr * 64, lane);
    fb0.init(lds0 + B_REGION, wc * 32, lane);
    fb1.init(lds0 + B_REGION, wc * 32, lane);
    constexpr int A_EARLY = Frags<A_KM, 2>::NREADS;             // reads issued after B0's in phase 0
    CsRegs csr;
    if constexpr (CS) csr.init(lds0 + A_REGION);
    auto mfma_quadrant = [&](auto sa_tag, auto sb_tag, const Frags<B_KM, 1>& fb) {
        constexpr int SA = decltype(sa_tag)::value, SB = decltype(sb_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[2 * SA + f][SB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb.get(0, ks), fa.get(f, ks), acc[2 * SA + f][SB], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // kE: k0 of the K-tile in buffer E of this iteration
    auto phase = [&](auto p_tag, int kE) {
        constexpr int P = decltype(p_tag)::value, BUF = P >> 2, PH = P & 3;
        if constexpr (PH == 0) {
            fb0.template read<sub_off(BUF, 0)>();
            __builtin_amdgcn_sched_barrier(0);
            fa.template read<sub_off(BUF, 0)>();
            if constexpr (CS) { if (do_cs) csr.template read<sub_off(BUF, 0)>(); }
        }
        if constexpr (PH == 1) fb1.template read<sub_off(BUF, 1)>();
        if constexpr (PH == 2) {
            fa.template read<sub_off(BUF, 1)>();
            if constexpr (CS) { if (do_cs) csr.template read<sub_off(BUF, 1)>(); }
        }
        if constexpr (P == 0) da.template issue<1, 1>(smem, kE + KT, kend);
        if constexpr (P == 1) db.template issue<0, 0>(smem, kE + 2 * KT, kend);
        if constexpr (P == 2) da.template issue<0, 0>(smem, kE + 2 * KT, kend);
        if constexpr (P == 3) db.template issue<0, 1>(smem, kE + 2 * KT, kend);
        if constexpr (P == 4) da.template issue<0, 1>(smem, kE + 2 * KT, kend);
        if constexpr (P == 5) db.template issue<1, 0>(smem, kE + 3 * KT, kend);
        if constexpr (P == 6) da.template issue<1, 0>(smem, kE + 3 * KT, kend);
        if constexpr (P == 7) db.template issue<1, 1>(smem, kE + 3 * KT, kend);
        // what the NEXT phase reads must have landed: the image issued six sub-tiles ago (five younger ones = 10 instructions stay in flight).
        // P%4 = 2 has nothing to wait for (the next phase reads nothing).  [First version: vmcnt(6) in P%4 = 3 only -- every image then had to
        // land within three phases (~1 us) of its issue, which L2-warm operands do and HBM-cold ones (a train step's weights) do not.]
        if constexpr (PH != 2) wait_vmcnt<10>();
        if constexpr (PH == 0) {                                // B0's reads are done: its image may be re-staged next phase
            if (CS && do_cs) wait_lgkm<(A_EARLY + 2 > 15 ? 15 : A_EARLY + 2)>();
            else wait_lgkm<(A_EARLY > 15 ? 15 : A_EARLY)>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 0) {
            wait_frags(fb0);
            wait_frags(fa);
            if constexpr (CS) { if (do_cs) csr.add(cs[0]); }
            mfma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fb0);
        }
        if constexpr (PH == 1) { wait_frags(fb1); mfma_quadrant(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, fb1); }
        if constexpr (PH == 2) {
            wait_frags(fa);
            if constexpr (CS) { if (do_cs) csr.add(cs[1]); }
            mfma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, fb1);
        }
        if constexpr (PH == 3) { mfma_quadrant(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, fb0); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int kbeg = kt0 * KT;
    db.template issue<0, 0>(smem, kbeg, kend);
    da.template issue<0, 0>(smem, kbeg, kend);
    db.template issue<0, 1>(smem, kbeg, kend);
    da.template issue<0, 1>(smem, kbeg, kend);
    db.template issue<1, 0>(smem, kbeg + KT, kend);
    da.template issue<1, 0>(smem, kbeg + KT, kend);
    db.template issue<1, 1>(smem, kbeg + KT, kend);
    wait_vmcnt<10>();                                           // E.B0 and E.A0 (phase 0's reads) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();                  // the second wave row runs half a phase behind the first
    __builtin_amdgcn_sched_barrier(0);
    for (int t = 0; t < ntl; t += 2) {
        const int kE = kbeg + t * KT;
        phase(std::integral_constant<int, 0>{}, kE);
        phase(std::integral_constant<int, 1>{}, kE);
        phase(std::integral_constant<int, 2>{}, kE);
        phase(std::integral_constant<int, 3>{}, kE);
        if (t + 1 >= ntl) break;
        phase(std::integral_constant<int, 4>{}, kE);
        phase(std::integral_constant<int, 5>{}, kE);
        phase(std::integral_constant<int, 6>{}, kE);
        phase(std::integral_constant<int, 7>{}, kE);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();                                            // the masked DMAs of the last phases
#endif
}

// The column sums of mainloop<.., CS = true>: the 32 k-row groups of a column meet in LDS (every wave is past its last fragment read and the
// masked DMAs have been waited for) and thread m < 256 stores tile column m.  Fixed summation order.
__device__ __forceinline__ void colsum_store(unsigned char* smem, const float (*cs)[8], int m0, int M, float* dst, bool accum) {
    float* red = reinterpret_cast<float*>(smem);               // [32 groups][256 columns]
    const int t = threadIdx.x, kk = t >> 4, cc = t & 15;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float* mine = red + kk * 256 + tile_row<true>(cc * 8, i);
        *reinterpret_cast<float4*>(mine) = make_float4(cs[i][0], cs[i][1], cs[i][2], cs[i][3]);
        *reinterpret_cast<float4*>(mine + 4) = make_float4(cs[i][4], cs[i][5], cs[i][6], cs[i][7]);
    }
    __syncthreads();
    if (t < 256 && m0 + t < M) {
        float v = 0.f;
        for (int g = 0; g < 32; ++g) v += red[g * 256 + t];
        dst[m0 + t] = accum ? dst[m0 + t] + v : v;
    }
}

}  // namespace p8
