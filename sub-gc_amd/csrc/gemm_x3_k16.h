// K = 16 stages for the bf16-pipe GEMM modes (see gemm_x3.h): the LDS image of a stage shrinks from 61 KB to 37 KB (three
// planes), so TWO 512-thread workgroups fit a CU.  DESIGN.md 3.1 measured what a workgroup that is alone on its CU costs
// (it cannot hide its own latencies: barrier hand-offs, fragment reads, the tile prologue and the 64 KB epilogue are all
// exposed); with the 32-deep stages of gemm_x3.h the three-plane form is exactly in that situation.
// Included by gemm_f32.hip inside its anonymous namespace, after gemm_x3.h (uses x3_split4, bf16x8, GemmArgs, ld4).
#pragma once

constexpr int XK = 16;                                       // K per LDS stage = one v_mfma_f32_32x32x16_bf16 step
constexpr int X16_ROW = 24;                                  // bf16 per LDS row: 16 k + 8 pad (48 B: 16-B aligned, conflict-free b128 reads)

// two consecutive-k fp32 values -> one packed bf16 pair per plane
template <int TERMS>
__device__ __forceinline__ void x3_split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    if (TERMS == 1) {
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(x0), "v"(x1));
        p2 = p1; p3 = p1;
        return;
    }
    const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const uint32_t l0 = __float_as_uint(r0 - __uint_as_float(m0)), l1 = __float_as_uint(r1 - __uint_as_float(m1));
    p1 = x3_pack_hi(h1, h0); p2 = x3_pack_hi(m1, m0); p3 = x3_pack_hi(l1, l0);
}

// Staging of one 128-row x 16-k operand tile by 256 threads: 2 float4 per thread, two register sets.
//   K-contiguous operand: thread t holds k = 4*(t & 3) .. +3 of rows (t >> 2) and (t >> 2) + 64;
//   K-major operand:      thread t holds rows 4*rg .. +3 at k = 2*kp and 2*kp + 1  (kp = (t >> 2) & 7, rg = (t & 3) + 4*(t >> 5)),
//                         transposed in registers: each row gets its (k, k+1) pair as one packed dword per plane.
template <int ROWS, bool KMAJOR, int TERMS>
struct StageX16 {
    static_assert(ROWS == 128, "x3 path: 128-row tiles");
    static constexpr int NV = 2;
    static constexpr int PLANES = TERMS == 1 ? 1 : 3;
    static constexpr int PLANE = ROWS * X16_ROW;             // bf16 elements per plane
    float4 rs[2][NV];
    const float* base[NV];
    int64_t ld_;
    unsigned oks[2];
    int c4_, rr_, kp_, rg_;

    __device__ __forceinline__ void init(const float* src, int64_t ld, int row0, int nrows) {
        const int t = threadIdx.x & 255;
        ld_ = ld;
        if (!KMAJOR) {
            c4_ = t & 3; rr_ = t >> 2;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int64_t srow = max(min(row0 + rr_ + v * 64, nrows - 1), 0);
                base[v] = src + srow * ld + c4_ * 4;
            }
        } else {
            kp_ = (t >> 2) & 7; rg_ = (t & 3) + 4 * (t >> 5);
            const int col = row0 + rg_ * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) base[j] = src + (col < nrows ? col : 0);
        }
    }
    template <int S>
    __device__ __forceinline__ void load(int k0, int K) {
        float4 (&r)[NV] = rs[S];
        unsigned ok = 0;
        if (!KMAJOR) {
            const bool in = k0 + c4_ * 4 < K;
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + (in ? k0 : -(c4_ * 4)));
            ok = in ? 3u : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int k = k0 + kp_ * 2 + j;
                r[j] = ld4(base[j] + (int64_t)(k < K ? k : 0) * ld_);
                if (k < K) ok |= 1u << j;
            }
        }
        oks[S] = ok;
    }
    template <int S>
    __device__ __forceinline__ void load_interior(int k0) {
        float4 (&r)[NV] = rs[S];
        oks[S] = 3u;
        if (!KMAJOR) {
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = ld4(base[v] + k0);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) r[j] = ld4(base[j] + (int64_t)(k0 + kp_ * 2 + j) * ld_);
        }
    }
    template <int S, bool MASK>
    __device__ __forceinline__ void store(uint16_t* st) const {
        const float4 (&r)[NV] = rs[S];
        const unsigned ok = MASK ? oks[S] : 3u;
        if (!KMAJOR) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const bool in = ok != 0;
                uint2 p1, p2, p3;
                x3_split2<TERMS>(in ? r[v].x : 0.f, in ? r[v].y : 0.f, p1.x, p2.x, p3.x);
                x3_split2<TERMS>(in ? r[v].z : 0.f, in ? r[v].w : 0.f, p1.y, p2.y, p3.y);
                uint16_t* d = st + (rr_ + v * 64) * X16_ROW + c4_ * 4;
                *reinterpret_cast<uint2*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint2*>(d + PLANE) = p2;
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
            }
        } else {
            const float a[4] = {r[0].x, r[0].y, r[0].z, r[0].w}, b[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
            const bool oa = ok & 1u, ob = ok & 2u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                    // row 4*rg + e gets (k, k+1) = (2*kp, 2*kp + 1)
                uint32_t p1, p2, p3;
                x3_split2<TERMS>(oa ? a[e] : 0.f, ob ? b[e] : 0.f, p1, p2, p3);
                uint16_t* d = st + (rg_ * 4 + e) * X16_ROW + kp_ * 2;
                *reinterpret_cast<uint32_t*>(d) = p1;
                if (PLANES == 3) {
                    *reinterpret_cast<uint32_t*>(d + PLANE) = p2;
                    *reinterpret_cast<uint32_t*>(d + 2 * PLANE) = p3;
                }
            }
        }
    }
};

constexpr size_t x16_lds_bytes(int BM, int BN, int planes) { return (size_t)2 * planes * (BM + BN) * X16_ROW * sizeof(uint16_t); }

template <int BM, int BN, bool TA, bool TB, int MT, int NT, int TERMS>
__device__ __forceinline__ void mainloop_x16_ws(const GemmArgs& p, float* smem_f, int M, int K, int m0, int n0, int kt0, int kt1,
                                                f32x16 (&acc)[MT][NT]) {
    // kt0 / kt1 count 32-wide K-tiles (the caller's unit): this loop walks them in 16-wide stages
    constexpr int WM = BM / 2, WN = BN / 2;
    using SA = StageX16<BM, TA, TERMS>;
    using SB = StageX16<BN, !TB, TERMS>;
    constexpr int PL = SA::PLANES;
    constexpr int STAGE = PL * (SA::PLANE + SB::PLANE);
    uint16_t* const smem = reinterpret_cast<uint16_t*>(smem_f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = kt0 * 2, s1 = min(kt1 * 2, (K + XK - 1) / XK);
    if (s1 <= s0) return;
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    if (wave >= 4) {
        SA sa; SB sb;
        sa.init(p.A, p.lda, m0, M);
        sb.init(p.B, p.ldb, n0, p.N);
        sa.template load<0>(s0 * XK, K);
        sb.template load<0>(s0 * XK, K);
        if (s0 + 1 < s1) {
            sa.template load<1>((s0 + 1) * XK, K);
            sb.template load<1>((s0 + 1) * XK, K);
        }
        sa.template store<0, true>(smem); sb.template store<0, true>(smem + PL * SA::PLANE);
        __syncthreads();
        auto ptile = [&](int st, auto q_tag, auto steady_tag) {
            constexpr int Q = decltype(q_tag)::value;
            constexpr bool STEADY = decltype(steady_tag)::value;
            uint16_t* sn = smem + (Q ^ 1) * STAGE;
            if (STEADY) {
                sa.template load_interior<Q>((st + 2) * XK);
                sb.template load_interior<Q>((st + 2) * XK);
                sa.template store<Q ^ 1, false>(sn); sb.template store<Q ^ 1, false>(sn + PL * SA::PLANE);
            } else {
                if (st + 2 < s1) {
                    sa.template load<Q>((st + 2) * XK, K);
                    sb.template load<Q>((st + 2) * XK, K);
                }
                if (st + 1 < s1) { sa.template store<Q ^ 1, true>(sn); sb.template store<Q ^ 1, true>(sn + PL * SA::PLANE); }
            }
            __syncthreads();
        };
        int st = s0;
        const int steady_end = min(s1, K / XK) - 2;
        for (; st + 1 < steady_end; st += 2) {
            ptile(st, Q0{}, std::true_type{});
            ptile(st + 1, Q1{}, std::true_type{});
        }
        for (; st < s1; ++st) {
            if (((st - s0) & 1) == 0) ptile(st, Q0{}, std::false_type{});
            else ptile(st, Q1{}, std::false_type{});
        }
    } else {
        const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
        constexpr int TI[6] = {0, 2, 1, 0, 1, 0}, TJ[6] = {2, 0, 1, 1, 0, 0};
        __syncthreads();
        for (int st = s0; st < s1; ++st) {
            const uint16_t* sc = smem + ((st - s0) & 1) * STAGE;
            bf16x8 fa[MT][PL], fb[NT][PL];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const uint16_t* q = sc + (wm + a * 32 + (lane & 31)) * X16_ROW + (lane >> 5) * 8;
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) fa[a][pl] = *reinterpret_cast<const bf16x8*>(q + pl * SA::PLANE);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const uint16_t* q = sc + PL * SA::PLANE + (wn + b * 32 + (lane & 31)) * X16_ROW + (lane >> 5) * 8;
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) fb[b][pl] = *reinterpret_cast<const bf16x8*>(q + pl * SB::PLANE);
            }
#pragma unroll
            for (int t = (TERMS == 1 ? 5 : 0); t < 6; ++t)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][TJ[t]], fa[a][TI[t]], acc[a][b], 0, 0, 0);   // swapped: C^T tile
            __syncthreads();
        }
    }
}
