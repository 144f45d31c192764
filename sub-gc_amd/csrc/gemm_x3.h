// fp32 GEMM main loop on the bf16 matrix pipe: "3-way split" operands, 6 MFMA terms, fp32 accumulate.
// Included by gemm_f32.hip inside its anonymous namespace (uses GemmArgs, BK, ld4, f32x16).
//
// Why: v_mfma_f32_32x32x2_f32 delivers 64 FLOP/cycle/SIMD (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 delivers
// 1024 (2.5 PFLOP/s).  Every fp32 number is EXACTLY the sum of three bf16 numbers (24 significand bits =
// 8 + 8 + 8, truncation split: x1 = top 8 bits, x2 = top 8 bits of x - x1, x3 = x - x1 - x2), and a product of
// two bf16 numbers is exact in fp32, so
//     a*b = sum_{i,j} a_i*b_j          (9 exact terms)
// of which the three smallest (a2*b3, a3*b2, a3*b3 <= 2^-24 |a*b|) are dropped -- the same order as ONE fp32
// rounding of the product -- and the remaining six are accumulated in fp32 by the matrix pipe, smallest first:
//     a1*b3, a3*b1, a2*b2, a1*b2, a2*b1, a1*b1.
// Six bf16 MFMAs of K=16 replace eight fp32 MFMAs of K=2: 6*32 = 192 matrix-pipe cycles per 32x32x16 block
// instead of 512, i.e. 2.67x the fp32-pipe rate at fp32-level accuracy (tests: error vs fp64 is within 2x of the
// fp32-MFMA kernel's on every GEMM shape of the path).
//
// (The three-plane mode runs the 16-deep-stage variant of this loop, gemm_x3_k16.h -- two workgroups per CU; the loop in
// this file serves the single-plane bf16 mode, whose 41 KB LDS image already allows that.)
//
// Data path per K-tile (BK = 32): fp32 operands HBM -> registers (float4, exactly like the fp32 kernel),
// split on the VALU (2 and, 2 sub per element + 1.5 perm to pack), written to LDS as three bf16 planes in a
// K-CONTIGUOUS image [row][32 k + 8 pad] whatever the operand's memory layout: a K-major operand (A^T or a
// [K,N] B) is loaded as 4(k) x 4(row) register blocks and transposed in registers, so that every LDS write
// is an 8-byte, conflict-free ds_write_b64 and every fragment is ONE ds_read_b128 (8 consecutive k of a row).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_ROW = 40;                                  // bf16 per LDS row: 32 k + 8 pad (80 B: 16-B aligned, conflict-free b128 reads)

__device__ __forceinline__ uint32_t x3_pack_hi(uint32_t hi_word, uint32_t lo_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);   // (hi_word & 0xffff0000) | (lo_word >> 16)
}

// four consecutive-k fp32 values of one row -> 4 bf16 in each of the three planes.
// TERMS == 1 is the plain bf16 mode (BASELINE configs 3 and 5: "bf16" compute, fp32 storage and accumulation):
// ONE plane, rounded to nearest-even instead of truncated, one MFMA term.
template <int TERMS>
__device__ __forceinline__ void x3_split4(const float (&x)[4], uint2& p1, uint2& p2, uint2& p3) {
    uint32_t h[4], m[4], l[4];
    if (TERMS == 1) {                                            // v_cvt_pk_bf16_f32: two round-to-nearest-even conversions per instruction
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1.x) : "v"(x[0]), "v"(x[1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1.y) : "v"(x[2]), "v"(x[3]));
        p2 = p1; p3 = p1;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t u = __float_as_uint(x[e]);
        h[e] = u & 0xffff0000u;
        const float r = x[e] - __uint_as_float(h[e]);
        m[e] = __float_as_uint(r) & 0xffff0000u;
        l[e] = __float_as_uint(r - __uint_as_float(m[e]));
    }
    p1.x = x3_pack_hi(h[1], h[0]); p1.y = x3_pack_hi(h[3], h[2]);
    p2.x = x3_pack_hi(m[1], m[0]); p2.y = x3_pack_hi(m[3], m[2]);
    p3.x = x3_pack_hi(l[1], l[0]); p3.y = x3_pack_hi(l[3], l[2]);
}

// Staging of one 128-row operand tile (VEC addressing only: ld % 4 == 0, 16-B aligned base, K % 4 == 0).
template <int ROWS, bool KMAJOR, int TERMS>
struct StageX3 {
    static_assert(ROWS == 128, "x3 path: 128-row tiles");
    static constexpr int PLANES = TERMS == 1 ? 1 : 3;
    static constexpr int NV = 4;
    static constexpr int PLANE = ROWS * X3_ROW;              // bf16 elements per plane
    float4 rs[2][NV];                                        // TWO register sets: tile j lives in set (j - kt0) & 1
    const float* base[NV];
    int64_t ld_;
    unsigned oks[2];                                         // per set, bit v: rs[.][v] is inside K (decided per tile)
    // K-contiguous: r[v] = 4 consecutive k (c4) of row rr;  K-major: r[j] = rows 4*rg..+3 at k = 4*kg + j
    int rr_[KMAJOR ? 1 : NV], c4_, kg_, rg_;

    __device__ __forceinline__ void init(const float* src, int64_t ld, int row0, int nrows, const int32_t* rows_idx) {
        const int t = threadIdx.x & 255;                     // producer waves of the specialised form are threads 256..511
        ld_ = ld;
        if (!KMAJOR) {
            c4_ = t & 7;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int rr = (t >> 3) + v * 32;
                rr_[v] = rr;
                int64_t srow = max(min(row0 + rr, nrows - 1), 0);   // rows past the edge: any valid row (never stored)
                if (rows_idx != nullptr) srow = max(rows_idx[srow], 0);
                base[v] = src + srow * ld + c4_ * 4;
            }
        } else {
            kg_ = (t >> 2) & 7;
            rg_ = (t & 3) + 4 * (t >> 5);
            const int col = row0 + rg_ * 4;
            const int colc = col < nrows ? col : 0;          // nrows % 4 == 0 on this path: all in or all out
#pragma unroll
            for (int j = 0; j < NV; ++j) base[j] = src + colc;
        }
    }

    template <int S>
    __device__ __forceinline__ void load(int k0, int K) {
        float4 (&r)[NV] = rs[S];
        unsigned& ok = oks[S];
        ok = 0;
        if (!KMAJOR) {
            const int kk = k0 + c4_ * 4;
            const bool in = kk < K;
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + (in ? k0 : -(c4_ * 4)));
            ok = in ? 0xfu : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int k = k0 + kg_ * 4 + j;
                r[j] = ld4(base[j] + (int64_t)(k < K ? k : 0) * ld_);
                if (k < K) ok |= 1u << j;
            }
        }
    }

    // interior tile: every k in range -> no tests, no selects
    template <int S>
    __device__ __forceinline__ void load_interior(int k0) {
        float4 (&r)[NV] = rs[S];
        oks[S] = 0xfu;
        if (!KMAJOR) {
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + k0);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) r[j] = ld4(base[j] + (int64_t)(k0 + kg_ * 4 + j) * ld_);
        }
    }
    template <int S>
    __device__ __forceinline__ void store_interior(uint16_t* st) const {
        const float4 (&r)[NV] = rs[S];
        if (!KMAJOR) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float x[4] = {r[v].x, r[v].y, r[v].z, r[v].w};
                uint2 p1, p2, p3;
                x3_split4<TERMS>(x, p1, p2, p3);
                uint16_t* d = st + rr_[v] * X3_ROW + c4_ * 4;
                *reinterpret_cast<uint2*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint2*>(d + PLANE) = p2;
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
            }
        } else {
            const float q[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w},
                                   {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x[4] = {q[0][e], q[1][e], q[2][e], q[3][e]};
                uint2 p1, p2, p3;
                x3_split4<TERMS>(x, p1, p2, p3);
                uint16_t* d = st + (rg_ * 4 + e) * X3_ROW + kg_ * 4;
                *reinterpret_cast<uint2*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint2*>(d + PLANE) = p2;
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
            }
        }
    }

    // split and write the three planes of LDS stage `st` (st points at plane 0 of this operand)
    template <int S>
    __device__ __forceinline__ void store(uint16_t* st) const {
        const float4 (&r)[NV] = rs[S];
        const unsigned ok = oks[S];
        if (!KMAJOR) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const bool in = ok != 0;
                const float x[4] = {in ? r[v].x : 0.f, in ? r[v].y : 0.f, in ? r[v].z : 0.f, in ? r[v].w : 0.f};
                uint2 p1, p2, p3;
                x3_split4<TERMS>(x, p1, p2, p3);
                uint16_t* d = st + rr_[v] * X3_ROW + c4_ * 4;
                *reinterpret_cast<uint2*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint2*>(d + PLANE) = p2;
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
            }
        } else {
            const float q[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w},
                                   {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {                    // row 4*rg + e, k = 4*kg .. 4*kg+3
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = ((ok >> j) & 1u) ? q[j][e] : 0.f;
                uint2 p1, p2, p3;
                x3_split4<TERMS>(x, p1, p2, p3);
                uint16_t* d = st + (rg_ * 4 + e) * X3_ROW + kg_ * 4;
                *reinterpret_cast<uint2*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint2*>(d + PLANE) = p2;
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
            }
        }
    }
};

template <int PLANE, int PLANES>
__device__ __forceinline__ void x3_frag(const uint16_t* st, int r0, int kstep, int lane, bf16x8 (&out)[PLANES]) {
    const uint16_t* p = st + (r0 + (lane & 31)) * X3_ROW + kstep * 16 + (lane >> 5) * 8;
#pragma unroll
    for (int pl = 0; pl < PLANES; ++pl) out[pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
}

constexpr size_t x3_lds_bytes(int BM, int BN, int planes) { return (size_t)2 * planes * (BM + BN) * X3_ROW * sizeof(uint16_t); }

// ---- warp-specialised form: 512 threads = 4 MFMA waves + 4 staging waves ------------------------------------------
// A single-role loop (one wave per SIMD interleaving ~250 staging instructions with 48 MFMAs in program order through
// sched_group_barrier) was measured first: every latency (global load -> split -> ds_write -> barrier -> ds_read) lies
// on its critical path and PMC showed the matrix pipe 36 % busy.  Here each SIMD hosts TWO waves with different jobs and
// the hardware interleaves them (DESIGN.md 3.2 has the numbers of both):
//   waves 0-3 (consumers): ds_read_b128 fragments + MFMAs of LDS stage Q, nothing else;
//   waves 4-7 (producers): request tile kt+2 (global -> registers, two register sets), split tile kt+1 on the VALU and
//                          write its bf16 planes into LDS stage Q^1.
// One workgroup barrier per K-tile hands stage Q^1 to the consumers and stage Q back to the producers.
template <int BM, int BN, bool TA, bool TB, int MT, int NT, int TERMS>
__device__ __forceinline__ void mainloop_x3_ws(const GemmArgs& p, float* smem_f, int M, int K, int m0, int n0, int kt0, int kt1,
                                               f32x16 (&acc)[MT][NT]) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr bool A_KM = TA, B_KM = !TB;
    using SA = StageX3<BM, A_KM, TERMS>;
    using SB = StageX3<BN, B_KM, TERMS>;
    constexpr int PL = SA::PLANES;
    constexpr int STAGE = PL * (SA::PLANE + SB::PLANE);
    uint16_t* const smem = reinterpret_cast<uint16_t*>(smem_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (kt1 <= kt0) return;
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        SA sa; SB sb;
        sa.init(p.A, p.lda, m0, M, TA ? nullptr : p.a_rows);
        sb.init(p.B, p.ldb, n0, p.N, nullptr);
        sa.template load<0>(kt0 * BK, K);
        sb.template load<0>(kt0 * BK, K);
        if (kt0 + 1 < kt1) {
            sa.template load<1>((kt0 + 1) * BK, K);
            sb.template load<1>((kt0 + 1) * BK, K);
        }
        sa.template store<0>(smem); sb.template store<0>(smem + PL * SA::PLANE);
        __syncthreads();
        auto ptile = [&](int kt, auto q_tag, auto steady_tag) {
            constexpr int Q = decltype(q_tag)::value;
            constexpr bool STEADY = decltype(steady_tag)::value;
            uint16_t* sn = smem + (Q ^ 1) * STAGE;
            if (STEADY) {
                sa.template load_interior<Q>((kt + 2) * BK);
                sb.template load_interior<Q>((kt + 2) * BK);
                sa.template store_interior<Q ^ 1>(sn); sb.template store_interior<Q ^ 1>(sn + PL * SA::PLANE);
            } else {
                if (kt + 2 < kt1) {
                    sa.template load<Q>((kt + 2) * BK, K);
                    sb.template load<Q>((kt + 2) * BK, K);
                }
                if (kt + 1 < kt1) { sa.template store<Q ^ 1>(sn); sb.template store<Q ^ 1>(sn + PL * SA::PLANE); }
            }
            __syncthreads();
        };
        int kt = kt0;
        const int steady_end = min(kt1, K / BK) - 2;
        for (; kt + 1 < steady_end; kt += 2) {
            ptile(kt, Q0{}, std::true_type{});
            ptile(kt + 1, Q1{}, std::true_type{});
        }
        for (; kt < kt1; ++kt) {
            if (((kt - kt0) & 1) == 0) ptile(kt, Q0{}, std::false_type{});
            else ptile(kt, Q1{}, std::false_type{});
        }
    } else {
        // ------------------------------------------------------------------ consumers
        const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
        bf16x8 fa[2][MT][PL], fb[2][NT][PL];
        constexpr int TI[6] = {0, 2, 1, 0, 1, 0}, TJ[6] = {2, 0, 1, 1, 0, 0};
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const uint16_t* sc = smem + ((kt - kt0) & 1) * STAGE;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#pragma unroll
                for (int a = 0; a < MT; ++a) x3_frag<SA::PLANE, PL>(sc, wm + a * 32, f, lane, fa[f][a]);
#pragma unroll
                for (int b = 0; b < NT; ++b) x3_frag<SB::PLANE, PL>(sc + PL * SA::PLANE, wn + b * 32, f, lane, fb[f][b]);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int t = (TERMS == 1 ? 5 : 0); t < 6; ++t)
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[f][b][TJ[t]], fa[f][a][TI[t]], acc[a][b], 0, 0, 0);   // swapped: C^T tile
            __syncthreads();
        }
    }
}
