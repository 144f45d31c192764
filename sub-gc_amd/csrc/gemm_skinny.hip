// Skinny (weight-streaming) GEMM for decode: C[M,N] = act(A[M,K] W[N,K]^T + bias), M <= 80.
//
// One caption = one row, so a decode step multiplies <= 16 activation rows by 152 MB of fp32 weights:
// the contraction is HBM-bound (reference: the per-step nn.Linear / nn.LSTMCell calls of
// AttModel.py:332-340,411-423,453 at batch = kept sub-graphs of ONE image).  Big MFMA tiles would waste
// >= 84 % of their rows; this kernel streams the weights once through 16-row v_mfma_f32_16x16x4_f32 tiles
// (see the kernel comment below for the data layout).
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- the matrix pipe as a reduction engine ------------------------------------------------------------------------------
// (Two VALU forms came first -- K spread over the workgroup, then a wave owning whole W rows with the activations staged in
// LDS: 50 and 20 us on the 38 MB logit matrix, single-shot timelines with nothing overlapped; both are gone from the tree.)
// v_mfma_f32_16x16x4_f32 contracts
// 16 W rows against <= 16 activation rows with NO cross-lane reduction and no LDS staging at all:
//   * a workgroup owns 16 consecutive W rows; its WAVES waves interleave the K axis in 16-wide steps (wave w takes steps
//     w, w+WAVES, ...), so together they walk 16 x (WAVES x 64 B) contiguous bytes per row and iteration;
//   * lane l loads ONE float4 of W (row l%16, k = step*16 + (l/16)*4 .. +3) and ONE float4 of the activations (row l%16 = m,
//     same k) per step; component j of both feeds the j-th of four MFMAs (the k-slot <-> address map only has to agree
//     between the two operands).  The activations come straight from L2 (<= 192 KB, read by every workgroup);
//   * loads run D steps ahead in a register ring with clamped (never predicated) addresses, so the compiler's vmcnt waits stay
//     partial; the refill is guarded by the wave-uniform step count.  D = 2 where the launch fills the chip (eight waves per
//     SIMD hide the latency; deeper rings measured slower, see the dispatch), D = 4 for the 32-64 workgroup encoder shapes;
//   * the WAVES partial 16x16 tiles are summed through 8 KB of LDS in a fixed order (deterministic), + bias, ReLU.
// 37.5 % of the matrix pipe's columns are padding at M = 10 -- irrelevant: the pipe needs 2 us of the ~10 us the stream takes.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LSTM form (LSTM = true): W is the [4R, K] gate matrix with its rows PERMUTED so that a workgroup's 16 rows are the four
// gates (i, f, g, o) of four consecutive hidden units (row 16*b + 4*g + u  <-  gate g of unit 4*b + u).  The cell update then
// runs in the epilogue of the same launch -- c = f*c_prev + i*g, h = o*tanh(c), h stored to up to three places -- so a
// decode step needs no separate pointwise kernel and the [S, 4R] pre-activations never reach memory.  The additive gate
// terms (a per-token table row or a plain [S,4R] array, a second [S,4R] array, two bias vectors) and c_prev are fetched
// by wave mt (< MT) BEFORE the K loop, so their latency hides behind the weight stream.
struct LstmEpi {
    const float* add1; int64_t ld1; const int64_t* tok; int tok_rows;       // add1 row = tok ? clamp(tok[m]) : m
    const float* add2; int64_t ld2;
    const float* b0; const float* b1;
    const float* c_prev; float* c;
    float* h0; int64_t ldh0; float* h1; int64_t ldh1; float* h2; int64_t ldh2;
    int R;
};

// MT > 1: the same stream against MT 16-row activation tiles (M <= 16*MT: the one-image encoder GEMMs, 37 node / 65 relation
// rows) -- one W load, MT activation loads and 4*MT MFMAs per step; `add` [M,N] is an optional residual term of the epilogue.
template <int WAVES, int D, bool LSTM, int MT>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_mfma_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                                      int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                                      const float* __restrict__ bias, int M, int N, int K, int relu,
                                                                      LstmEpi ep, const float* __restrict__ add, int64_t ldadd) {
    static_assert(!LSTM || MT <= 2, "the fused cell update handles up to two 16-row activation tiles (wave mt owns tile mt)");
    __shared__ float part[WAVES][MT * 256];
    __shared__ float tile[LSTM ? 256 * MT : 1];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: uniform loop control
    const int r16 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const float* wrow = W + (int64_t)min(n0 + r16, N - 1) * ldb;            // rows past N: clamped, their results are not stored
    const float* arow[MT];                                                 // columns m >= M of a tile: garbage, never stored
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = A + (int64_t)min(mt * 16 + r16, M - 1) * lda;
    const int steps = (K + 15) >> 4, steps_full = K >> 4;                     // 16-wide K steps; only the last one can be partial
    const int mine = steps > wave ? (steps - wave + WAVES - 1) / WAVES : 0;   // steps of this wave ...
    const int mine_full = steps_full > wave ? (steps_full - wave + WAVES - 1) / WAVES : 0;   // ... that lie entirely inside K
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 wq[D], aq[D][MT];
    float gadd[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;                        // LSTM: wave mt < MT, lane (u = lane/16, m = 16*mt + lane%16)
    const int eu = lane >> 4, em = wave * 16 + (lane & 15), ej = blockIdx.x * 4 + eu;
    const bool elive = LSTM && wave < MT && em < M && ej < ep.R;
    if (elive) {
        int64_t row1 = em;
        if (ep.tok) { const int64_t w = ep.tok[em]; row1 = w < 0 ? 0 : (w >= ep.tok_rows ? ep.tok_rows - 1 : w); }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = g * ep.R + ej;
            float v = 0.f;
            if (ep.b0) v += ep.b0[col];
            if (ep.b1) v += ep.b1[col];
            if (ep.add1) v += ep.add1[row1 * ep.ld1 + col];
            if (ep.add2) v += ep.add2[(int64_t)em * ep.ld2 + col];
            gadd[g] = v;
        }
        if (ep.c_prev) cprev = ep.c_prev[(int64_t)em * ep.R + ej];
    }
    auto k_of = [&](int i) { return ((wave + i * WAVES) << 4) + (kq << 2); };
    auto issue = [&](int slot, int i) {                                       // i-th step of this wave -> ring slot (static index)
        const int k = k_of(i);
        const int kc = (i < mine && k < K) ? k : 0;                           // clamped address, never a predicated load
        wq[slot] = ld4(wrow + kc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[slot][mt] = ld4(arow[mt] + kc);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    int r = 0;
    for (; (r + 1) * D <= mine_full; ++r) {                                   // steady state: no selects, partial vmcnt waits only
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float4 w = wq[d];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 a = aq[d][mt];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, a.x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, a.y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, a.z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, a.w, acc[mt], 0, 0, 0);
            }
            if ((r + 1) * D + d < mine) issue(d, (r + 1) * D + d);
        }
    }
    for (; r * D < mine; ++r) {                                               // tail: steps past this wave's share or past K add zeros
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = r * D + d;
            const bool in = i < mine && k_of(i) < K;
            const float4 w = wq[d];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 a = aq[d][mt];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, in ? a.x : 0.f, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, in ? a.y : 0.f, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, in ? a.z : 0.f, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, in ? a.w : 0.f, acc[mt], 0, 0, 0);
            }
            if ((r + 1) * D + d < mine) issue(d, (r + 1) * D + d);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) part[wave][mt * 256 + v * 64 + lane] = acc[mt][v];
    __syncthreads();
    for (int e = t; e < MT * 256; e += WAVES * 64) {                          // element (row 4*(l/16)+v of the 16 W rows, m = 16*mt + l%16)
        const int mt = e >> 8, v = (e >> 6) & 3, l = e & 63;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += part[w][e];
        const int n = n0 + 4 * (l >> 4) + v, m = mt * 16 + (l & 15);
        if (LSTM) tile[(4 * (l >> 4) + v) * (16 * MT) + m] = sum;             // row 4*g + u of this workgroup, column m
        else if (m < M && n < N) {
            float o = sum + (bias ? bias[n] : 0.f);
            if (add) o += add[(int64_t)m * ldadd + n];
            if (relu) o = fmaxf(o, 0.f);
            C[(int64_t)m * ldc + n] = o;
        }
    }
    if (LSTM) {
        __syncthreads();
        if (elive) {
            constexpr int TW = 16 * MT;
            const float ig = sigmoidf_(tile[(0 + eu) * TW + em] + gadd[0]), fg = sigmoidf_(tile[(4 + eu) * TW + em] + gadd[1]);
            const float gg = tanhf(tile[(8 + eu) * TW + em] + gadd[2]), og = sigmoidf_(tile[(12 + eu) * TW + em] + gadd[3]);
            const float cn = fg * cprev + ig * gg, hn = og * tanhf(cn);
            ep.c[(int64_t)em * ep.R + ej] = cn;
            if (ep.h0) ep.h0[(int64_t)em * ep.ldh0 + ej] = hn;
            if (ep.h1) ep.h1[(int64_t)em * ep.ldh1 + ej] = hn;
            if (ep.h2) ep.h2[(int64_t)em * ep.ldh2 + ej] = hn;
        }
    }
}

}  // namespace

namespace subgc {

int gemm_skinny_nt(const float* A, int64_t lda, const float* W, int64_t ldb, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                   int relu, hipStream_t s, const float* add, int64_t ldadd) {
    if (M < 1 || M > 80) return -100;
    if (M > 16) {                                                 // 2..5 activation tiles, shallower ring
        const int wgs = (N + 15) / 16, mt = (M + 15) / 16;
#define SUBGC_SKINNY_MT(MT_, D_)                                                                                                            \
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<8, D_, false, MT_>), dim3(wgs), dim3(512), 0, s, A, lda, W, ldb, C, ldc, bias, M, N, K, relu, \
                       LstmEpi{}, add, ldadd)
        // 3..5 tiles = the one-image encoder GEMMs (37 / 65 rows, N = 512..1024: 32-64 workgroups, at most one per CU): there
        // occupancy cannot hide the latency and the deeper (guarded) ring does
        if (mt == 2) SUBGC_SKINNY_MT(2, 2);
        else if (mt == 3) SUBGC_SKINNY_MT(3, 4);
        else if (mt == 4) SUBGC_SKINNY_MT(4, 4);
        else SUBGC_SKINNY_MT(5, 4);
#undef SUBGC_SKINNY_MT
        return subgc::check_launch("subgc_gemm_f32(skinny)");
    }
    const int wgs = (N + 15) / 16;
    // Ring depth 2.  Deeper rings were measured and are SLOWER (rocprofv3, 10 rows: logits 17.0 us at depth 8, 13.3 at 4, 11.9 at
    // 2; depth 16: 31 us): the last steady-state round prefetches a full ring past the wave's share (clamped, useless loads --
    // half of all load instructions at depth 8 when a wave owns 8 steps), and with ~50 VGPRs eight waves per SIMD hide the
    // latency that the ring was meant to hide.  The refill is also guarded by the (wave-uniform) step count, so no load is
    // issued past the wave's share: logits 11.9 -> 10.0 us (38 MB: 3.8 TB/s); guarded depth 4: 10.8, depth 8: 11.5.
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<8, 2, false, 1>), dim3(wgs), dim3(512), 0, s, A, lda, W, ldb, C, ldc, bias, M, N, K, relu, LstmEpi{}, add, ldadd);
    return subgc::check_launch("subgc_gemm_f32(skinny)");
}

}  // namespace subgc

// ---- C ABI: one LSTM cell step of a decode batch of <= 16 rows, gate GEMM and cell update in one launch ---------------------
SUBGC_API int subgc_lstm_step_skinny(const float* x, int64_t ldx, const float* w_perm, int64_t ldw, int K, int S, int R, const float* add1,
                                     int64_t ld1, const int64_t* tok, int tok_rows, const float* add2, int64_t ld2, const float* b0,
                                     const float* b1, const float* c_prev, float* c, float* h0, int64_t ldh0, float* h1, int64_t ldh1,
                                     float* h2, int64_t ldh2, void* stream) {
    SUBGC_REQUIRE(S >= 0 && S <= 32 && R > 0 && R % 4 == 0 && K > 0 && K % 4 == 0, "lstm_step_skinny: need S <= 32, R % 4 == 0, K % 4 == 0");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && w_perm && c && (h0 || h1 || h2), "lstm_step_skinny: null pointer");
    SUBGC_REQUIRE(ldx >= K && ldx % 4 == 0 && ldw >= K && ldw % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_perm % 16) == 0,
                  "lstm_step_skinny: x / w_perm rows must be 16-byte aligned float4 rows");
    SUBGC_REQUIRE(!tok || (add1 && tok_rows > 0), "lstm_step_skinny: tok needs a table in add1");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * S * 4.0 * R * K);
    LstmEpi ep{add1, ld1, tok, tok_rows, add2, ld2, b0, b1, c_prev, c, h0, ldh0, h1, ldh1, h2, ldh2, R};
    const int N = 4 * R, wgs = N / 16;
    if (S > 16)                                                                // two activation tiles (beam search: <= 10 sub-graphs x 2-3 beams)
        hipLaunchKernelGGL((gemm_skinny_mfma_kernel<8, 2, true, 2>), dim3(wgs), dim3(512), 0, s, x, ldx, w_perm, ldw, nullptr, 0, nullptr, S, N, K, 0, ep, nullptr, 0);
    else
        hipLaunchKernelGGL((gemm_skinny_mfma_kernel<8, 2, true, 1>), dim3(wgs), dim3(512), 0, s, x, ldx, w_perm, ldw, nullptr, 0, nullptr, S, N, K, 0, ep, nullptr, 0);
    return subgc::check_launch("subgc_lstm_step_skinny");
}
