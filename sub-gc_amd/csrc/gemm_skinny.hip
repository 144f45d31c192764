// Skinny (weight-streaming) GEMM for decode: C[M,N] = act(A[M,K] W[N,K]^T + bias), M <= 80.
//
// One caption = one row, so a decode step multiplies <= 16 activation rows by 152 MB of fp32 weights:
// the contraction is HBM-bound (reference: the per-step nn.Linear / nn.LSTMCell calls of
// AttModel.py:332-340,411-423,453 at batch = kept sub-graphs of ONE image).  Big MFMA tiles would waste
// >= 84 % of their rows; this kernel streams the weights once through 16-row v_mfma_f32_16x16x4_f32 tiles
// (see the kernel comment below for the data layout).
#include "common.h"
#include "bf16_util.h"

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
// Greedy pick folded into the step's launches (AttModel.py:295-316 with sample_max).  The logits launch leaves, per decode row, the
// packed arg-max (ordered logit bits << 32 | ~column: ties -> smaller column = torch.max's first max) by 64-bit atomicMax -- into ONE
// OF EIGHT slots per row (slot = workgroup & 7, i.e. its XCD; every slot on its own 128-byte line): 593 workgroups hammering one
// address per row measured +5 us on the 10 us launch (device-scope atomics queue memory-side), 74 per slot do not; reducing plain
// per-workgroup partials in the consumer instead costs it 24 MB of extra reads (+6 us).  It also leaves per-workgroup (max, sum exp)
// partials for the log-prob, which nothing needs before the loop ends.  The NEXT step's attention-LSTM launch merges the eight slots
// into the input word (finished rows feed 0) and its workgroup 0 does the reference's bookkeeping exactly once: seq[:, t], the
// unfinished flags, the live count that drives the early break (AttModel.py:318-319).  No pick launch, no [n, V+1] logits in memory.
constexpr int PICK_SLOTS = 8, PICK_LINE = 16;                               // uint64 per 128-byte line; slot (m, x) at (m * 8 + x) * 16
struct PickIn {
    const unsigned long long* best;     // [16 rows][8 slots][16] of the previous step's logits launch; NULL = token ids come from LstmEpi::tok
    const int32_t* unf_in; int32_t* unf_out;
    int64_t* seq; int T; int t_prev;    // seq[m * T + t_prev] <- word
    int32_t* count_out;                 // live rows after step t_prev (pre-zeroed, accumulated)
    const int32_t* prev_count;          // live rows after step t_prev - 1 (NULL at t_prev = 0): 0 = the loop has ended, write nothing
    unsigned long long* best_reset;     // the other buffer, cleared for this step's logits launch (NULL: none follows)
};
struct LstmEpi {
    const float* add1; int64_t ld1; const int64_t* tok; int tok_rows;       // add1 row = tok ? clamp(tok[m]) : m
    const float* add2; int64_t ld2;
    const float* b0; const float* b1;
    const float* c_prev; float* c;
    float* h0; int64_t ldh0; float* h1; int64_t ldh1; float* h2; int64_t ldh2;
    int R;
    PickIn pk;
};
struct PickOut { unsigned long long* best; float* lse_part; };              // logits launch: slots as above, [workgroups][16][2] partials
// A SECOND product in the same launch (DUAL): workgroups [first_blocks, gridDim.x) stream W2 [N2, K2] against A2 [M, K2] and store
// plainly (bias2 optional).  Round 6, one-image decode: the products of a token step that share no data dependence share a launch --
// [logits (+ pick) | attention-LSTM gate product] and [h2att | the language LSTM's h_att / h_lang part] -- so a weight stream that used to
// sit on the step's dependent chain runs beside one that has to.  unperm_R (either problem's plain store): W's rows are in the LSTM
// forms' permuted gate order (row 16 b + 4 g + u = gate g of unit 4 b + u); the result column goes to g * R + 4 b + u, the order the
// cell kernels' additive terms use.
struct Second { const float* A; int64_t lda; const void* W; int64_t ldb; float* C; int64_t ldc; const float* bias; int N; int K; int first_blocks; int unperm_R; };

__device__ __forceinline__ uint32_t ordered_bits(float f) {                 // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int picked_word(const PickIn& pk, int m, bool& unf) {
    unsigned long long b = 0ull;
#pragma unroll
    for (int x = 0; x < PICK_SLOTS; ++x) { const unsigned long long v = pk.best[(m * PICK_SLOTS + x) * PICK_LINE]; b = v > b ? v : b; }
    const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(b & 0xFFFFFFFFull);
    unf = idx > 0u && (pk.t_prev == 0 || pk.unf_in[m] != 0);
    return unf ? (int)idx : 0;
}

// MT > 1: the same stream against MT 16-row activation tiles (M <= 16*MT: the one-image encoder GEMMs, 37 node / 65 relation
// rows) -- one W load, MT activation loads and 4*MT MFMAs per step; `add` [M,N] is an optional residual term of the epilogue.
// WB16: the weight matrix is bf16-stored (compute_dtype = bf16: the decode step then streams 60 instead of 120 MB); a lane's four k of
// a W row are one 8-byte load, widened to fp32 in registers -- activations, accumulation and everything downstream stay fp32.
constexpr int PICK_WAVES = 8, PLAIN_WAVES = 8;      // 16 measured: no gain (593 logits workgroups are 2-3 per CU already)
constexpr int LSTM_WAVES = 16;                                                // waves per workgroup of the fused LSTM-step launches

template <int WAVES, int D, bool LSTM, int MT, bool PICK = false, bool WB16 = false, bool DUAL = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_mfma_kernel(const float* A, int64_t lda, const void* W,
                                                                      int64_t ldb, float* C, int64_t ldc,
                                                                      const float* bias, int M, int N, int K, int relu,
                                                                      LstmEpi ep, const float* __restrict__ add, int64_t ldadd, PickOut po = PickOut{},
                                                                      Second sec = Second{}, int unperm_R = 0) {
    static_assert(!LSTM || MT <= 2, "the fused cell update handles up to two 16-row activation tiles (wave mt owns tile mt)");
    static_assert(!PICK || (!LSTM && MT == 1), "the arg-max epilogue belongs to the plain one-tile form (the logits launch)");
    static_assert(!DUAL || (!LSTM && MT == 1), "a second product rides the plain one-tile forms");
    __shared__ float part[WAVES][MT * 256];
    __shared__ float tile[(LSTM || PICK) ? 256 * MT : 1];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: uniform loop control
    const int r16 = lane & 15, kq = lane >> 4;
    int bx = blockIdx.x;
    bool second = false;                                                    // workgroup-uniform
    if (DUAL && bx >= sec.first_blocks) {
        second = true;
        bx -= sec.first_blocks;
        A = sec.A; lda = sec.lda; W = sec.W; ldb = sec.ldb; C = sec.C; ldc = sec.ldc; bias = sec.bias; N = sec.N; K = sec.K; unperm_R = sec.unperm_R;
        relu = 0;
    }
    const int n0 = bx * 16;
    const float* wrow = static_cast<const float*>(W) + (WB16 ? 0 : (int64_t)min(n0 + r16, N - 1) * ldb);   // rows past N: clamped, their results are not stored
    const uint16_t* wrow16 = static_cast<const uint16_t*>(W) + (WB16 ? (int64_t)min(n0 + r16, N - 1) * ldb : 0);
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
    const int eu = lane >> 4, em = wave * 16 + (lane & 15), ej = bx * 4 + eu;
    const bool elive = LSTM && wave < MT && em < M && ej < ep.R;
    if (LSTM && ep.pk.best && blockIdx.x == 0 && t < 128) {                   // the previous step's pick, filed once (workgroup 0)
        if (ep.pk.best_reset) ep.pk.best_reset[t * PICK_LINE] = 0ull;
        if (t < M) {
            bool unf;
            const int w = picked_word(ep.pk, t, unf);
            const bool alive = !(ep.pk.prev_count && *ep.pk.prev_count == 0);
            if (alive) {
                ep.pk.seq[(int64_t)t * ep.pk.T + ep.pk.t_prev] = w;
                if (unf) atomicAdd(ep.pk.count_out, 1);
            }
            ep.pk.unf_out[t] = alive && unf;
        }
    }
    if (elive) {
        int64_t row1 = em;
        if (ep.pk.best) { bool unf; const int w = picked_word(ep.pk, em, unf); row1 = w >= ep.tok_rows ? ep.tok_rows - 1 : w; }
        else if (ep.tok) { const int64_t w = ep.tok[em]; row1 = w < 0 ? 0 : (w >= ep.tok_rows ? ep.tok_rows - 1 : w); }
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
        wq[slot] = WB16 ? subgc_load4_bf(wrow16 + kc) : ld4(wrow + kc);
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
        else if (PICK && !second) {
            // (e == t < 256 here: waves 0..3, all lanes.)  Online (max, arg, sum exp) merge: two shuffles fold the four lanes that
            // hold a decode row inside this wave, the four waves' results meet in LDS -- a short chain instead of a 16-step scan
            const float o = n < N ? sum + (bias ? bias[n] : 0.f) : -INFINITY;
            if (C && m < M && n < N) C[(int64_t)m * ldc + n] = o;
            float mx = o, se = n < N ? 1.f : 0.f;
            int arg = n;
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float om = __shfl_xor(mx, off, 64), os = __shfl_xor(se, off, 64);
                const int oa = __shfl_xor(arg, off, 64);
                const float big = fmaxf(mx, om);
                se = big == -INFINITY ? 0.f : se * __expf(mx - big) + os * __expf(om - big);
                if (om > mx || (om == mx && oa < arg)) arg = oa;
                mx = big;
            }
            if (l < 16) { tile[(v * 16 + l) * 4] = mx; tile[(v * 16 + l) * 4 + 1] = se; tile[(v * 16 + l) * 4 + 2] = __int_as_float(arg); }
        } else if (m < M && n < N) {
            float o = sum + (bias ? bias[n] : 0.f);
            if (add && !second) o += add[(int64_t)m * ldadd + n];
            if (relu) o = fmaxf(o, 0.f);
            const int nc = unperm_R ? ((n & 15) >> 2) * unperm_R + (n >> 4) * 4 + (n & 3) : n;     // permuted gate row -> gate-major column
            C[(int64_t)m * ldc + nc] = o;
        }
    }
    if (PICK && !second) {
        __syncthreads();
        if (t < M) {                                                          // thread = decode row: merge the four waves' results
            float mx = tile[t * 4], se = tile[t * 4 + 1];
            int arg = __float_as_int(tile[t * 4 + 2]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float om = tile[(w * 16 + t) * 4], os = tile[(w * 16 + t) * 4 + 1];
                const int oa = __float_as_int(tile[(w * 16 + t) * 4 + 2]);
                const float big = fmaxf(mx, om);
                se = big == -INFINITY ? 0.f : se * __expf(mx - big) + os * __expf(om - big);
                if (om > mx || (om == mx && oa < arg)) arg = oa;
                mx = big;
            }
            atomicMax(po.best + (t * PICK_SLOTS + (bx & (PICK_SLOTS - 1))) * PICK_LINE,
                      ((unsigned long long)ordered_bits(mx) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)arg));
            float* lp = po.lse_part + ((int64_t)bx * 16 + t) * 2;
            lp[0] = mx; lp[1] = se;
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
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<PLAIN_WAVES, 2, false, 1>), dim3(wgs), dim3(PLAIN_WAVES * 64), 0, s, A, lda, W, ldb, C, ldc, bias, M, N, K, relu, LstmEpi{}, add, ldadd);
    return subgc::check_launch("subgc_gemm_f32(skinny)");
}

}  // namespace subgc

// C[M,N] = act(A[M,K] W[N,K]^T + bias) with bf16-STORED W and fp32 activations / results, M <= 16 (the h2att product of a bf16 decode step)
SUBGC_API int subgc_gemm_skinny_wb16(const float* A, int64_t lda, const uint16_t* W, int64_t ldw, float* C, int64_t ldc, const float* bias, int M,
                                     int N, int K, int relu, void* stream) {
    SUBGC_REQUIRE(M >= 1 && M <= 16 && N > 0 && K > 0 && K % 4 == 0, "gemm_skinny_wb16: need 1 <= M <= 16 and K % 4 == 0");
    SUBGC_REQUIRE(A && W && C && lda >= K && lda % 4 == 0 && ldw >= K && ldw % 4 == 0 && ldc >= N && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 8) == 0,
                  "gemm_skinny_wb16: 16-byte aligned fp32 rows, 8-byte aligned bf16 rows");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * M * (double)N * K);
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<PLAIN_WAVES, 2, false, 1, false, true>), dim3((N + 15) / 16), dim3(PLAIN_WAVES * 64), 0, s, A, lda, W, ldw, C, ldc, bias, M, N, K, relu,
                       LstmEpi{}, nullptr, 0, PickOut{});
    return subgc::check_launch("subgc_gemm_skinny_wb16");
}

// ---- C ABI: one LSTM cell step of a decode batch of <= 16 rows, gate GEMM and cell update in one launch ---------------------
SUBGC_API int subgc_lstm_step_skinny(const float* x, int64_t ldx, const void* w_perm, int64_t ldw, int K, int S, int R, const float* add1,
                                     int64_t ld1, const int64_t* tok, int tok_rows, const float* add2, int64_t ld2, const float* b0,
                                     const float* b1, const float* c_prev, float* c, float* h0, int64_t ldh0, float* h1, int64_t ldh1,
                                     float* h2, int64_t ldh2, int w_bf16, void* stream) {
    SUBGC_REQUIRE(S >= 0 && S <= 32 && R > 0 && R % 4 == 0 && K > 0 && K % 4 == 0, "lstm_step_skinny: need S <= 32, R % 4 == 0, K % 4 == 0");
    if (S == 0) return SUBGC_OK;
    SUBGC_REQUIRE(x && w_perm && c && (h0 || h1 || h2), "lstm_step_skinny: null pointer");
    SUBGC_REQUIRE(ldx >= K && ldx % 4 == 0 && ldw >= K && ldw % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_perm % (w_bf16 ? 8 : 16)) == 0,
                  "lstm_step_skinny: x / w_perm rows must be 16-byte aligned float4 rows (8-byte aligned bf16 rows)");
    SUBGC_REQUIRE(!tok || (add1 && tok_rows > 0), "lstm_step_skinny: tok needs a table in add1");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * S * 4.0 * R * K);
    LstmEpi ep{add1, ld1, tok, tok_rows, add2, ld2, b0, b1, c_prev, c, h0, ldh0, h1, ldh1, h2, ldh2, R, PickIn{}};
    const int N = 4 * R, wgs = N / 16;
    // 4R/16 workgroups (250 at R = 1000) are at most one per CU: 16 waves each instead of 8 (tools/ubench/lstm_step_bench.py, att + lang
    // pair, 10 rows: 21.2 -> 19.9 us with fp32 weights, 22.2 -> 17.4 us with bf16 ones; deeper rings instead: 20.6 / 21.0; 4 waves: 30)
#define SUBGC_LSTM_GO(MT_, W16_) \
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<LSTM_WAVES, 2, true, MT_, false, W16_>), dim3(wgs), dim3(LSTM_WAVES * 64), 0, s, x, ldx, w_perm, ldw, nullptr, 0, nullptr, S, N, K, 0, ep, nullptr, 0, PickOut{})
    if (S > 16) { if (w_bf16) SUBGC_LSTM_GO(2, true); else SUBGC_LSTM_GO(2, false); }      // two activation tiles (beam search: <= 10 sub-graphs x 2-3 beams)
    else { if (w_bf16) SUBGC_LSTM_GO(1, true); else SUBGC_LSTM_GO(1, false); }
#undef SUBGC_LSTM_GO
    return subgc::check_launch("subgc_lstm_step_skinny");
}

// ---- round 6: the token step's independent weight streams share launches ---------------------------------------------------------------
namespace {
// The attention LSTM's cell update of a greedy decode step as its own small launch: the gate PRODUCT no longer waits for the pick (it
// streams beside the logits, subgc_skinny_dual), only this kernel does.  pre [S, 4R] = H1 . Wc1^T in gate-major column order; everything
// else -- the word's x -> gates table row, the fc term, biases, c_prev, the three h destinations, and workgroup 0's filing of the pick --
// as in the LSTM form of gemm_skinny_mfma_kernel, in the same order of additions.
__global__ __launch_bounds__(256) void lstm_cell_pick_kernel(const float* __restrict__ pre, int64_t ldpre, LstmEpi ep, int M) {
    const int t = threadIdx.x;
    if (ep.pk.best && blockIdx.x == 0 && t < 128) {
        if (ep.pk.best_reset) ep.pk.best_reset[t * PICK_LINE] = 0ull;
        if (t < M) {
            bool unf;
            const int w = picked_word(ep.pk, t, unf);
            const bool alive = !(ep.pk.prev_count && *ep.pk.prev_count == 0);
            if (alive) {
                ep.pk.seq[(int64_t)t * ep.pk.T + ep.pk.t_prev] = w;
                if (unf) atomicAdd(ep.pk.count_out, 1);
            }
            ep.pk.unf_out[t] = alive && unf;
        }
    }
    const int idx = blockIdx.x * 256 + t;
    if (idx >= M * ep.R) return;
    const int em = idx / ep.R, ej = idx - em * ep.R;
    // two dependent memory latencies (the pick's slots, then the word's table row) bound this launch: everything that does not depend on
    // the word is requested BEFORE the slots are read
    float pv[4], bs[4], a2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = g * ep.R + ej;
        pv[g] = pre[(int64_t)em * ldpre + col];
        bs[g] = (ep.b0 ? ep.b0[col] : 0.f) + (ep.b1 ? ep.b1[col] : 0.f);              // (0 + b0) + b1, as the LSTM form adds them
        a2[g] = ep.add2 ? ep.add2[(int64_t)em * ep.ld2 + col] : 0.f;
    }
    const float cprev = ep.c_prev ? ep.c_prev[(int64_t)em * ep.R + ej] : 0.f;
    int64_t row1 = em;
    if (ep.pk.best) { bool unf; const int w = picked_word(ep.pk, em, unf); row1 = w >= ep.tok_rows ? ep.tok_rows - 1 : w; }
    else if (ep.tok) { const int64_t w = ep.tok[em]; row1 = w < 0 ? 0 : (w >= ep.tok_rows ? ep.tok_rows - 1 : w); }
    float g4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v = bs[g];
        if (ep.add1) v += ep.add1[row1 * ep.ld1 + g * ep.R + ej];
        if (ep.add2) v += a2[g];
        g4[g] = pv[g] + v;
    }
    const float ig = sigmoidf_(g4[0]), fg = sigmoidf_(g4[1]), gg = tanhf(g4[2]), og = sigmoidf_(g4[3]);
    const float cn = fg * cprev + ig * gg, hn = og * tanhf(cn);
    ep.c[(int64_t)em * ep.R + ej] = cn;
    if (ep.h0) ep.h0[(int64_t)em * ep.ldh0 + ej] = hn;
    if (ep.h1) ep.h1[(int64_t)em * ep.ldh1 + ej] = hn;
    if (ep.h2) ep.h2[(int64_t)em * ep.ldh2 + ej] = hn;
}
}  // namespace

// subgc_lstm_cell_pick: c, h <- LSTMCell pointwise from pre-activations `pre` [S, 4R] (gate-major columns: what subgc_skinny_dual leaves
// with unperm_R) + add1[word] + add2 + b0 + b1.  word of row m = the previous step's fused arg-max when best_prev != NULL (then the pick
// is filed: seq[m, t_prev], unf_out[m], counts[t_prev]; `best_reset`, the other buffer, is cleared for this step's logits launch), else tok[m] (tok may be NULL: add1 row m).  S <= 16.
SUBGC_API int subgc_lstm_cell_pick(const float* pre, int64_t ldpre, int S, int R, const float* add1, int64_t ld1, const int64_t* tok, int tok_rows,
                                   const float* add2, int64_t ld2, const float* b0, const float* b1, const float* c_prev, float* c, float* h0,
                                   int64_t ldh0, float* h1, int64_t ldh1, float* h2, int64_t ldh2, const uint64_t* best_prev, const int32_t* unf_in,
                                   int32_t* unf_out, int64_t* seq, int T, int t_prev, int32_t* count_out, const int32_t* prev_count,
                                   uint64_t* best_reset, void* stream) {
    SUBGC_REQUIRE(S >= 1 && S <= 16 && R > 0 && ldpre >= 4 * R, "lstm_cell_pick: need 1 <= S <= 16 and ldpre >= 4 R");
    SUBGC_REQUIRE(pre && c && (h0 || h1 || h2), "lstm_cell_pick: null pointer");
    SUBGC_REQUIRE(!(tok || best_prev) || (add1 && tok_rows > 0), "lstm_cell_pick: a word needs its table in add1");
    SUBGC_REQUIRE(!best_prev || (unf_out && seq && count_out && t_prev >= 0 && t_prev < T && (t_prev == 0 || unf_in)), "lstm_cell_pick: pick arguments");
    SUBGC_DEBUG_RANGE(tok, 8, S, 1, 1, 0, tok_rows - 1, -1, "lstm_cell_pick: tok (word ids)", stream);
    PickIn pk{};
    if (best_prev)
        pk = PickIn{reinterpret_cast<const unsigned long long*>(best_prev), unf_in, unf_out, seq, T, t_prev, count_out, prev_count,
                    reinterpret_cast<unsigned long long*>(best_reset)};
    LstmEpi ep{add1, ld1, tok, tok_rows, add2, ld2, b0, b1, c_prev, c, h0, ldh0, h1, ldh1, h2, ldh2, R, pk};
    hipLaunchKernelGGL(lstm_cell_pick_kernel, dim3((S * R + 255) / 256), dim3(256), 0, (hipStream_t)stream, pre, ldpre, ep, S);
    return subgc::check_launch("subgc_lstm_cell_pick");
}

// subgc_skinny_dual: TWO weight-streaming products of one decode step in one launch, S <= 16 rows each:
//   problem 1: C1 = x1 W1^T + bias1 (C1 may be NULL with `best`), and with best != NULL the fused arg-max / log-sum-exp partials of
//              the greedy pick (W1 = the logit matrix; `best` slots as PickIn describes, zero before the launch; lse_part[(wg * 16 + m) * 2 + {0, 1}]);
//   problem 2: C2 = x2 W2^T (+ bias2), plain.
// unperm1_R / unperm2_R != 0: that problem's W rows are in the permuted gate order of the LSTM forms and its result columns are written
// gate-major (column g R + j).  Both weight matrices fp32 or both bf16 (w_bf16).
SUBGC_API int subgc_skinny_dual(int S, const float* x1, int64_t ldx1, const void* W1, int64_t ldw1, const float* bias1, int N1, int K1, float* C1,
                                int64_t ldc1, int unperm1_R, uint64_t* best, float* lse_part, const float* x2, int64_t ldx2, const void* W2,
                                int64_t ldw2, const float* bias2, int N2, int K2, float* C2, int64_t ldc2, int unperm2_R, int w_bf16, void* stream) {
    SUBGC_REQUIRE(S >= 1 && S <= 16 && N1 > 0 && N2 > 0 && K1 > 0 && K2 > 0 && K1 % 4 == 0 && K2 % 4 == 0, "skinny_dual: need 1 <= S <= 16 and K %% 4 == 0");
    SUBGC_REQUIRE(x1 && W1 && x2 && W2 && C2 && (C1 || best) && (!best || lse_part), "skinny_dual: null pointer");
    SUBGC_REQUIRE(ldx1 >= K1 && ldx1 % 4 == 0 && ldw1 >= K1 && ldw1 % 4 == 0 && ldx2 >= K2 && ldx2 % 4 == 0 && ldw2 >= K2 && ldw2 % 4 == 0 &&
                      ((uintptr_t)x1 % 16) == 0 && ((uintptr_t)x2 % 16) == 0 && ((uintptr_t)W1 % (w_bf16 ? 8 : 16)) == 0 && ((uintptr_t)W2 % (w_bf16 ? 8 : 16)) == 0,
                  "skinny_dual: x / W rows must be 16-byte aligned float4 rows (8-byte aligned bf16 rows)");
    SUBGC_REQUIRE((!unperm1_R || N1 == 4 * unperm1_R) && (!unperm2_R || N2 == 4 * unperm2_R) && (!C1 || ldc1 >= N1) && ldc2 >= N2, "skinny_dual: destination / permutation sizes");
    SUBGC_REQUIRE(!(best && unperm1_R), "skinny_dual: the pick epilogue belongs to the logit matrix");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GEMM, s, 2.0 * S * ((double)N1 * K1 + (double)N2 * K2));
    const int nb1 = (N1 + 15) / 16, nb2 = (N2 + 15) / 16;
    Second sec{x2, ldx2, W2, ldw2, C2, ldc2, bias2, N2, K2, nb1, unperm2_R};
    PickOut po{reinterpret_cast<unsigned long long*>(best), lse_part};
#define SUBGC_DUAL_GO(PICK_, W16_) \
    hipLaunchKernelGGL((gemm_skinny_mfma_kernel<PLAIN_WAVES, 2, false, 1, PICK_, W16_, true>), dim3(nb1 + nb2), dim3(PLAIN_WAVES * 64), 0, s, x1, ldx1, W1, ldw1, C1, ldc1, \
                       bias1, S, N1, K1, 0, LstmEpi{}, nullptr, 0, po, sec, unperm1_R)
    if (best) { if (w_bf16) SUBGC_DUAL_GO(true, true); else SUBGC_DUAL_GO(true, false); }
    else { if (w_bf16) SUBGC_DUAL_GO(false, true); else SUBGC_DUAL_GO(false, false); }
#undef SUBGC_DUAL_GO
    return subgc::check_launch("subgc_skinny_dual");
}

namespace {
__global__ __launch_bounds__(64) void pick_file_kernel(PickIn pk, int M) {
    const int t = threadIdx.x;
    if (t >= M) return;
    bool unf;
    const int w = picked_word(pk, t, unf);
    const bool alive = !(pk.prev_count && *pk.prev_count == 0);
    if (alive) {
        pk.seq[(int64_t)t * pk.T + pk.t_prev] = w;
        if (unf) atomicAdd(pk.count_out, 1);
    }
    pk.unf_out[t] = alive && unf;
}
// seqlp[m, t] = log_softmax(logits_t[m])[argmax] = -log sum_wg sum_wg * exp(max_wg - max) for every step the loop reached
__global__ __launch_bounds__(256) void pick_lse_finish_kernel(const float* __restrict__ lse_part, int wgs, int S, int T, const int32_t* __restrict__ counts,
                                                              float* __restrict__ seqlp) {
    __shared__ float sm[16];
    const int t = blockIdx.x, m = blockIdx.y;
    if (t > 0 && counts[t - 1] == 0) return;                       // the reference has left its loop (AttModel.py:318-319)
    const float* p = lse_part + (int64_t)t * wgs * 32 + m * 2;
    float mx = -INFINITY;
    for (int w = threadIdx.x; w < wgs; w += 256) mx = fmaxf(mx, p[(int64_t)w * 32]);
    mx = block_max(mx, sm);
    float sum = 0.f;
    for (int w = threadIdx.x; w < wgs; w += 256) sum += p[(int64_t)w * 32 + 1] * expf(p[(int64_t)w * 32] - mx);
    sum = block_sum(sum, sm);
    if (threadIdx.x == 0) seqlp[(int64_t)m * T + t] = -logf(sum);
}
}  // namespace

// subgc_pick_file: the bookkeeping of subgc_lstm_cell_pick alone (the LAST pick of a loop has no following cell launch)
SUBGC_API int subgc_pick_file(const uint64_t* best_prev, const int32_t* unf_in, int32_t* unf_out, int64_t* seq, int S, int T, int t_prev,
                              int32_t* count_out, const int32_t* prev_count, void* stream) {
    SUBGC_REQUIRE(S >= 1 && S <= 16 && t_prev >= 0 && t_prev < T, "pick_file: bad sizes");
    SUBGC_REQUIRE(best_prev && unf_out && seq && count_out && (t_prev == 0 || unf_in), "pick_file: null pointer");
    PickIn pk{reinterpret_cast<const unsigned long long*>(best_prev), unf_in, unf_out, seq, T, t_prev, count_out, prev_count, nullptr};
    hipLaunchKernelGGL(pick_file_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pk, S);
    return subgc::check_launch("subgc_pick_file");
}

// subgc_pick_lse_finish: after the token loop, the log-probabilities of the T picks from the per-step partials
//   lse_part [T][ceil(V/16)][16][2] (one pass instead of a log-softmax reduction inside every step).
SUBGC_API int subgc_pick_lse_finish(const float* lse_part, int V, int S, int T, const int32_t* counts, float* seqlp, void* stream) {
    SUBGC_REQUIRE(V > 0 && S >= 1 && S <= 16 && T >= 1, "pick_lse_finish: bad sizes");
    SUBGC_REQUIRE(lse_part && counts && seqlp, "pick_lse_finish: null pointer");
    hipLaunchKernelGGL(pick_lse_finish_kernel, dim3(T, S), dim3(256), 0, (hipStream_t)stream, lse_part, (V + 15) / 16, S, T, counts, seqlp);
    return subgc::check_launch("subgc_pick_lse_finish");
}
