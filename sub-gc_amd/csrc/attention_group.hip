// Attention kernels for SHARED attention sets: the Full-GC model (AttModel.py:140-149) attends, for every one of an image's
// sentences, over the SAME node rows (all 36 nodes); the reference replicates the node features five times
// (gcn_backbone.py:50-51) and so reads, projects and differentiates five identical copies.  Here the sets exist once per image
// (v = relu(att_embed(X)) [B*Nn, R], u = ctx2att(v) [B*Nn, A]); ONE workgroup per image serves all of the image's live sentences:
// every node row of u / v (and d(u)) is loaded once and used for up to G sentences from registers, so the step's attention traffic
// drops ~G-fold and d(u) needs neither atomics nor a per-sentence copy.
//   rows[b*g + j]  position of the image's j-th sentence in the step's row arrays (the packed decoder's sorted rank, or the sentence
//                  index itself); a sentence takes part in a step iff 0 <= rows[...] < m (m = live rows of the step)
//   lens[row]      its number of valid nodes (<= Nn, the node rows per image)
// Same arithmetic per sentence as attention_vec.hip (softmax over the valid rows == the reference's softmax -> mask -> renormalise).
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

constexpr int GL = 128;     // node rows per image the grouped kernels handle (Full-GC: 37)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <bool UV16>
__device__ __forceinline__ float4 ldx(const void* base, int64_t i) {
    return UV16 ? subgc_load4_bf(static_cast<const uint16_t*>(base) + i) : ld4(static_cast<const float*>(base) + i);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float a, float4 x) { acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w; }

template <int CA, int CR, bool UV16, int G>
__global__ __launch_bounds__(256) void attn_fwd_group_kernel(const void* __restrict__ u, const void* __restrict__ v, const float* __restrict__ ah,
                                                             const float* __restrict__ w_a, const float* __restrict__ b_a,
                                                             const int32_t* __restrict__ rows, const int32_t* __restrict__ lens, int m, int g,
                                                             int Nn, void* __restrict__ ctx, int64_t ldctx, float* __restrict__ alpha,
                                                             int n_stride, int A, int R, int ctx_b16, const subgc::QSrc qs) {
    __shared__ float e_s[G][GL];
    __shared__ int row_s[G], len_s[G];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j0 = blockIdx.y * G;                                     // this workgroup's share of the image's sentences (see the entry point)
    if (t < G) {
        const int r = j0 + t < g ? rows[b * g + j0 + t] : -1;
        const bool live = r >= 0 && r < m;
        row_s[t] = live ? r : -1;
        len_s[t] = live ? min(min(lens[r], Nn), GL) : 0;
    }
    for (int i = t; i < G * GL; i += 256) (&e_s[0][0])[i] = 0.f;
    __syncthreads();
    int lmax = 0;
    int len_r[G], row_r[G];                                            // registers: the unrolled sentence loops index them statically
    bool any_live = false;
#pragma unroll
    for (int j = 0; j < G; ++j) { len_r[j] = len_s[j]; row_r[j] = row_s[j]; lmax = max(lmax, len_r[j]); any_live |= row_r[j] >= 0; }
    if (!any_live) return;                                             // a live sentence WITHOUT valid nodes still gets its (zero) ctx / alpha rows below
    const int64_t m0 = (int64_t)b * Nn;
    const int A4 = A >> 2, R4 = R >> 2;
    // scores: wave per node; the node's u chunk is loaded once and scored against every sentence's query
    {
        float4 q_[G][CA], w[CA];
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int a4 = lane + c * 64;
            const bool ok = a4 < A4;
            w[c] = ok ? ld4(w_a + a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const bool have = ok && row_s[j] >= 0;
                q_[j][c] = have ? subgc_load_q(ah, qs, (int64_t)row_s[j] * A + a4 * 4, a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (qs.out && have && wave == 0) st4(qs.out + (int64_t)row_s[j] * A + a4 * 4, q_[j][c]);   // the summed query, kept for the backward
            }
        }
        const float ba = b_a[0];
        // One wave per SIMD and every node row a compulsory HBM / Infinity-Cache miss (~2 us): nothing hides a load but other loads,
        // so a wave requests the rows of NCH of its nodes at once and scores them when they have landed (9 exposed round trips of
        // a 36-node image become 3; measured 58 -> 40 us came from the tanh alone, the rest is this latency).
        constexpr int NCH = 3;
        for (int i0 = wave; i0 < lmax; i0 += 4 * NCH) {
            float4 x[NCH][CA];
#pragma unroll
            for (int q = 0; q < NCH; ++q)
#pragma unroll
                for (int c = 0; c < CA; ++c)
                    x[q][c] = (i0 + 4 * q < lmax && lane + c * 64 < A4) ? ldx<UV16>(u, (m0 + i0 + 4 * q) * A + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float sc[NCH * G];
#pragma unroll
            for (int q = 0; q < NCH; ++q)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    float acc = 0.f;
                    if (i0 + 4 * q < len_r[j]) {                       // wave-uniform
#pragma unroll
                        for (int c = 0; c < CA; ++c)
                            acc += w[c].x * subgc_tanh(x[q][c].x + q_[j][c].x) + w[c].y * subgc_tanh(x[q][c].y + q_[j][c].y) + w[c].z * subgc_tanh(x[q][c].z + q_[j][c].z) +
                                   w[c].w * subgc_tanh(x[q][c].w + q_[j][c].w);
                    }
                    sc[q * G + j] = acc;
                }
            wave_sum_n<NCH * G>(sc);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NCH; ++q)
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        if (i0 + 4 * q < len_r[j]) e_s[j][i0 + 4 * q] = sc[q * G + j] + ba;
            }
        }
    }
    __syncthreads();
    // softmax over the valid rows: wave w takes sentences w, w+4, ...; a lane holds rows lane and lane + 64
    for (int j = wave; j < G; j += 4) {
        const int l = len_s[j];
        if (l == 0) {                                                  // dead slot, or a live sentence over an empty set: weights (and, e_s being zero, ctx) = 0
            if (alpha && row_s[j] >= 0)
                for (int i = lane; i < n_stride; i += 64) alpha[(int64_t)row_s[j] * n_stride + i] = 0.f;
            continue;
        }
        const float e0 = lane < l ? e_s[j][lane] : -INFINITY, e1 = lane + 64 < l ? e_s[j][lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(e0, e1));
        const float p0 = lane < l ? expf(e0 - mx) : 0.f, p1 = lane + 64 < l ? expf(e1 - mx) : 0.f;
        const float den = wave_sum(p0 + p1);
        e_s[j][lane] = p0 / den;
        e_s[j][lane + 64] = p1 / den;
        if (alpha) {
            float* ar = alpha + (int64_t)row_s[j] * n_stride;
            if (lane < n_stride) ar[lane] = p0 / den;
            if (lane + 64 < n_stride) ar[lane + 64] = p1 / den;
            for (int i = lane + 128; i < n_stride; i += 64) ar[i] = 0.f;
        }
    }
    __syncthreads();
    // contexts: thread = float4 column chunk; a node's v chunk is loaded once for all sentences
#pragma unroll
    for (int c = 0; c < CR; ++c) {
        const int r4 = t + c * 256;
        if (r4 >= R4) continue;
        float4 acc[G];
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t vp = m0 * R + r4 * 4;
        int i = 0;
        for (; i + 12 <= lmax; i += 12) {                          // twelve node rows in flight per thread: 3 round trips for 36 nodes
            float4 x[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) x[q] = ldx<UV16>(v, vp + (int64_t)(i + q) * R);
#pragma unroll
            for (int q = 0; q < 12; ++q)
#pragma unroll
                for (int j = 0; j < G; ++j) fma4(acc[j], e_s[j][i + q], x[q]);
        }
        for (; i + 2 <= lmax; i += 2) {
            const float4 x0 = ldx<UV16>(v, vp + (int64_t)i * R), x1 = ldx<UV16>(v, vp + (int64_t)(i + 1) * R);
#pragma unroll
            for (int j = 0; j < G; ++j) { fma4(acc[j], e_s[j][i], x0); fma4(acc[j], e_s[j][i + 1], x1); }
        }
        for (; i < lmax; ++i) {
            const float4 x0 = ldx<UV16>(v, vp + (int64_t)i * R);
#pragma unroll
            for (int j = 0; j < G; ++j) fma4(acc[j], e_s[j][i], x0);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (row_s[j] < 0) continue;
            const float o[4] = {acc[j].x, acc[j].y, acc[j].z, acc[j].w};
            subgc_store_act<4>(ctx, (int64_t)row_s[j] * ldctx + r4 * 4, o, ctx_b16);
        }
    }
}

template <int CA, int CR64, bool UV16, int G>
__global__ __launch_bounds__(256) void attn_bwd_group_kernel(const void* __restrict__ u, const void* __restrict__ v, const float* __restrict__ ah,
                                                             const float* __restrict__ w_a, const int32_t* __restrict__ rows,
                                                             const int32_t* __restrict__ lens, int m, int g, int Nn,
                                                             const float* __restrict__ alpha, int n_stride, const float* __restrict__ dctx,
                                                             int64_t lddctx, void* __restrict__ dah, float* __restrict__ du,
                                                             float* __restrict__ dw_a, float* __restrict__ db_a, int A, int R, int dah_b16,
                                                             float* __restrict__ dctx_keep, int64_t ldkeep, int n_planes, int64_t plane_stride,
                                                             int64_t du_split_stride) {
    __shared__ float al_s[G][GL];     // alpha, then de
    __shared__ float da_s[G][GL];
    __shared__ float4 part_d[G][128], part_w[G][128];
    __shared__ int row_s[G], len_s[G];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j0 = blockIdx.y * G;                                     // this workgroup's share of the image's sentences; it accumulates
    du += (int64_t)blockIdx.y * du_split_stride;                      // into ITS plane of d(u) (the caller adds the planes)
    if (t < G) {
        const int r = j0 + t < g ? rows[b * g + j0 + t] : -1;
        const bool live = r >= 0 && r < m;
        row_s[t] = live ? r : -1;
        len_s[t] = live ? min(min(lens[r], Nn), GL) : 0;
    }
    __syncthreads();
    int lmax = 0;
    int len_r[G], row_r[G];
    bool any_live = false;
#pragma unroll
    for (int j = 0; j < G; ++j) { len_r[j] = len_s[j]; row_r[j] = row_s[j]; lmax = max(lmax, len_r[j]); any_live |= row_r[j] >= 0; }
    if (!any_live) return;                                             // live sentences over empty sets fall through: zero dah / dw_a / db_a, summed dctx_keep
    for (int i = t; i < G * GL; i += 256) {
        const int j = i / GL, k = i - j * GL;
        al_s[j][k] = (k < len_s[j]) ? alpha[(int64_t)row_s[j] * n_stride + k] : 0.f;
        da_s[j][k] = 0.f;
    }
    __syncthreads();
    const int64_t m0 = (int64_t)b * Nn;
    const int A4 = A >> 2, R4 = R >> 2;
    {
        // d(ctx) of the image's sentences: the split-K planes are added on load by ALL threads at once (one round trip) and the sums
        // staged in LDS (G x R floats); a sentence's kept copy (dctx_keep) is written from there
        extern __shared__ __attribute__((aligned(16))) float dctx_s[];      // [G][R4] float4
        float4* gs = reinterpret_cast<float4*>(dctx_s);
        for (int idx = t; idx < G * R4; idx += 256) {
            const int j = idx / R4, c = idx - j * R4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_r[j] >= 0)
                for (int q = 0; q < n_planes; ++q) {
                    const float4 x = ld4(dctx + q * plane_stride + (int64_t)row_s[j] * lddctx + c * 4);
                    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
                }
            gs[idx] = a;
            if (dctx_keep && row_s[j] >= 0) st4(dctx_keep + (int64_t)row_s[j] * ldkeep + c * 4, a);
        }
        __syncthreads();
        // dalpha_i = <dctx_j, v_i>: wave per node; the rows of NCH of the wave's nodes are requested at once, and the NCH x G dot
        // products are reduced over the lanes TOGETHER
        float4 gd[G][CR64];
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int c = 0; c < CR64; ++c) gd[j][c] = (lane + c * 64 < R4) ? gs[j * R4 + lane + c * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NCH = 3;
        for (int i0 = wave; i0 < lmax; i0 += 4 * NCH) {
            float4 x[NCH][CR64];
#pragma unroll
            for (int q = 0; q < NCH; ++q)
#pragma unroll
                for (int c = 0; c < CR64; ++c)
                    x[q][c] = (i0 + 4 * q < lmax && lane + c * 64 < R4) ? ldx<UV16>(v, (m0 + i0 + 4 * q) * R + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float sc[NCH * G];
#pragma unroll
            for (int q = 0; q < NCH; ++q)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < CR64; ++c) acc += dot4(gd[j][c], x[q][c]);
                    sc[q * G + j] = acc;
                }
            wave_sum_n<NCH * G>(sc);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NCH; ++q)
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        if (i0 + 4 * q < len_r[j]) da_s[j][i0 + 4 * q] = sc[q * G + j];
            }
        }
    }
    __syncthreads();
    // de_i = alpha_i (dalpha_i - sum_k alpha_k dalpha_k): wave w takes sentences w, w+4, ...
    for (int j = wave; j < G; j += 4) {
        if (len_s[j] == 0) {
            if (lane == 0 && db_a && row_s[j] >= 0) db_a[row_s[j]] = 0.f;
            continue;
        }
        const float a0 = al_s[j][lane], a1 = al_s[j][lane + 64], d0 = da_s[j][lane], d1 = da_s[j][lane + 64];
        const float dot = wave_sum(a0 * d0 + a1 * d1);
        const float e0 = a0 * (d0 - dot), e1 = a1 * (d1 - dot);
        al_s[j][lane] = e0; al_s[j][lane + 64] = e1;
        const float desum = wave_sum(e0 + e1);
        if (lane == 0 && db_a) db_a[row_s[j]] = desum;
    }
    __syncthreads();
    // through tanh: 2 node streams x 128 float4 chunks of the hidden dimension; u and d(u) rows are touched once for all sentences
#pragma unroll
    for (int c = 0; c < CA; ++c) {
        const int a4 = (t & 127) + c * 128, grp = t >> 7;
        float4 dsum[G], wsum[G];
#pragma unroll
        for (int j = 0; j < G; ++j) { dsum[j] = make_float4(0.f, 0.f, 0.f, 0.f); wsum[j] = dsum[j]; }
        if (a4 < A4) {
            const float4 wa = ld4(w_a + a4 * 4);
            float4 ha[G];
#pragma unroll
            for (int j = 0; j < G; ++j) ha[j] = row_s[j] >= 0 ? ld4(ah + (int64_t)row_s[j] * A + a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int TCH = 6;                                    // u and d(u) rows of six nodes of this stream in flight at once
            for (int i0 = grp; i0 < lmax; i0 += 2 * TCH) {
                float4 x[TCH], d[TCH];
#pragma unroll
                for (int q = 0; q < TCH; ++q) {
                    const int i = i0 + 2 * q;
                    const int64_t o = (m0 + min(i, lmax - 1)) * A + a4 * 4;
                    x[q] = ldx<UV16>(u, o);
                    d[q] = ld4(du + o);
                }
#pragma unroll
                for (int q = 0; q < TCH; ++q) {
                    const int i = i0 + 2 * q;
                    if (i >= lmax) break;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        if (i >= len_r[j]) continue;
                        const float de = al_s[j][i];
                        const float t0 = subgc_tanh(x[q].x + ha[j].x), t1 = subgc_tanh(x[q].y + ha[j].y), t2 = subgc_tanh(x[q].z + ha[j].z), t3 = subgc_tanh(x[q].w + ha[j].w);
                        const float p0 = de * wa.x * (1.f - t0 * t0), p1 = de * wa.y * (1.f - t1 * t1);
                        const float p2 = de * wa.z * (1.f - t2 * t2), p3 = de * wa.w * (1.f - t3 * t3);
                        d[q].x += p0; d[q].y += p1; d[q].z += p2; d[q].w += p3;
                        dsum[j].x += p0; dsum[j].y += p1; dsum[j].z += p2; dsum[j].w += p3;
                        wsum[j].x += de * t0; wsum[j].y += de * t1; wsum[j].z += de * t2; wsum[j].w += de * t3;
                    }
                    st4(du + (m0 + i) * A + a4 * 4, d[q]);
                }
            }
        }
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int j = 0; j < G; ++j) { part_d[j][t & 127] = dsum[j]; part_w[j][t & 127] = wsum[j]; }
        }
        __syncthreads();
        if (grp == 0 && a4 < A4) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (row_s[j] < 0) continue;
                const float4 od = part_d[j][t & 127], ow = part_w[j][t & 127];
                const float o[4] = {dsum[j].x + od.x, dsum[j].y + od.y, dsum[j].z + od.z, dsum[j].w + od.w};
                subgc_store_act<4>(dah, (int64_t)row_s[j] * A + a4 * 4, o, dah_b16);
                st4(dw_a + (int64_t)row_s[j] * A + a4 * 4, make_float4(wsum[j].x + ow.x, wsum[j].y + ow.y, wsum[j].z + ow.z, wsum[j].w + ow.w));
            }
        }
    }
}

// d(v) of ALL time steps and ALL of the image's sentences in one pass: dv[b*Nn + i, :] = sum over (step t, sentence j live at t) of
// alpha_t[row_j, i] * dctx_t[row_j, :].  grid (B, column slices of 64 float4): a workgroup stages its 1 KB column slice of the d(ctx)
// rows and the attention weights of up to `tg` live (t, j) pairs in LDS (all T*g = 85 pairs of Full-GC in one go), then wave = node,
// lane = float4 column.  Every one of the image's Nn node rows is written (zeros where nothing attends), so dv needs no zero fill.
// step_off: the packed decoder's layout (step t's rows start at step_off[t], step_off[t+1] - step_off[t] of them live); the
// unpacked one is step_off[t] = t * S.
__global__ __launch_bounds__(256) void attn_dv_accum_group_kernel(const float* __restrict__ alpha, int n_stride, const float* __restrict__ dctx,
                                                                  int64_t lddctx, const int32_t* __restrict__ step_off, int T,
                                                                  const int32_t* __restrict__ rows, int g, int Nn, float* __restrict__ dv, int R,
                                                                  int tg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dv_lds[];
    __shared__ int so_s[66], rr_s[8], flat_s[1024], nl_s;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int R4 = R >> 2, l = min(Nn, n_stride), c4 = blockIdx.y * 64 + lane;
    const bool col_ok = c4 < R4;
    float4* g_s = reinterpret_cast<float4*>(dv_lds);                      // [tg][64]
    float* a_s = reinterpret_cast<float*>(g_s + (size_t)tg * 64);         // [tg][n_stride]
    // the step table and the image's row positions once (a dependent global read per (step, sentence) pair serialised the staging:
    // 85 round trips, 220 us)
    for (int i = t; i <= T; i += 256) so_s[i] = step_off[i];
    if (t < g) rr_s[t] = rows[b * g + t];
    bool first = true;
    const int pairs = T * g;
    for (int p0 = 0; p0 < pairs || first; p0 += tg) {
        __syncthreads();
        if (t == 0) {                                                      // flat row of every live pair of this group
            int nl = 0;
            for (int p = p0; p < min(pairs, p0 + tg); ++p) {
                const int tt = p / g, j = p - tt * g;
                const int r = rr_s[j], o = so_s[tt], cnt = so_s[tt + 1] - o;
                if (r >= 0 && r < cnt) flat_s[nl++] = o + r;
            }
            nl_s = nl;
        }
        __syncthreads();
        const int nl = nl_s;
        // stage: all loads of the group in flight at once -- thread = (pair, float4 column) for d(ctx), (pair, node) for alpha
        for (int idx = t; idx < nl * 64; idx += 256) {
            const int k = idx >> 6, c = blockIdx.y * 64 + (idx & 63);
            g_s[idx] = c < R4 ? ld4(dctx + (int64_t)flat_s[k] * lddctx + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int idx = t; idx < nl * l; idx += 256) {
            const int k = idx / l, i = idx - k * l;
            a_s[k * n_stride + i] = alpha[(int64_t)flat_s[k] * n_stride + i];
        }
        __syncthreads();
        if (nl == 0 && !first) continue;
        for (int i = wave; i < Nn; i += 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < l) {
                int k = 0;
                for (; k + 8 <= nl; k += 8) {                          // eight LDS read pairs in flight (the serial form was LDS-latency bound)
                    float a[8];
                    float4 x[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { a[q] = a_s[(k + q) * n_stride + i]; x[q] = g_s[(size_t)(k + q) * 64 + lane]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) fma4(acc, a[q], x[q]);
                }
                for (; k < nl; ++k) fma4(acc, a_s[k * n_stride + i], g_s[(size_t)k * 64 + lane]);
            }
            if (col_ok) {
                float* dvr = dv + ((int64_t)b * Nn + i) * R + c4 * 4;
                if (!first) { const float4 o = ld4(dvr); acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
                st4(dvr, acc);
            }
        }
        first = false;
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// An image's sentences are served by `splits` workgroups of ceil(g / splits) sentences each (blockIdx.y): with one workgroup per image
// the chip holds ONE wave per SIMD and every dependent instruction (the tanh chains) exposes its latency; two per image re-read
// the image's u / v rows from L2 but give every SIMD a second wave (fwd 40 -> 2x-wave figure in DESIGN 3.4).  The backward's two
// workgroups accumulate into separate d(u) planes.
inline int group_splits(int g) { return g >= 4 ? 2 : 1; }       // three per image measured the same as two (44.6 / 23.7 vs 45.2 / 25.3 us)
#define SUBGC_G_DISPATCH(CALL)                                   \
    do {                                                         \
        const int per = (g + splits - 1) / splits;               \
        if (per <= 2) { CALL(2); } else if (per <= 3) { CALL(3); } else if (per <= 5) { CALL(5); } else { CALL(8); } \
    } while (0)

namespace {
int attn_fwd_group_any(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* rows,
                       const int32_t* lens, int m, int B, int g, int Nn, void* ctx, int64_t ldctx, float* alpha, int n_stride, int A,
                       int R, int bf16_bits, void* stream, subgc::QSrc qs) {
    const int ctx_b16 = bf16_bits & 1, uv16 = (bf16_bits >> 1) & 1;
    SUBGC_REQUIRE(qs.n_planes >= 1 && qs.stride % 4 == 0 && al16(qs.bias) && al16(qs.out), "attn_fwd_group: query planes / bias / output must be 16-byte aligned");
    SUBGC_REQUIRE(B >= 0 && g >= 1 && g <= 8 && Nn >= 1 && Nn <= GL && m >= 0 && A > 0 && R > 0 && n_stride >= 0, "attn_fwd_group: bad sizes (g <= 8, Nn <= %d)", GL);
    if (B == 0 || m == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && v && ah && w_a && b_a && rows && lens && ctx, "attn_fwd_group: null pointer");
    SUBGC_REQUIRE(A % 4 == 0 && R % 4 == 0 && ldctx % 4 == 0 && al16(u) && al16(v) && al16(ah) && al16(w_a) && al16(ctx), "attn_fwd_group: A, R %% 4 == 0 and 16-byte aligned rows");
    const int ca = (A / 4 + 63) / 64, cr = (R / 4 + 255) / 256;
    SUBGC_REQUIRE(ca <= 2 && cr <= 2, "attn_fwd_group: att_hid_size <= 512 and rnn_size <= 2048");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    const int splits = group_splits(g);
#define SUBGC_FWD_G(G_)                                                                                                                              \
    do {                                                                                                                                               \
        if (uv16) { if (ca == 1 && cr == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<1, 1, true, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else if (ca == 2 && cr == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<2, 1, true, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else if (ca == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<1, 2, true, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else hipLaunchKernelGGL((attn_fwd_group_kernel<2, 2, true, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); } \
        else { if (ca == 1 && cr == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<1, 1, false, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else if (ca == 2 && cr == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<2, 1, false, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else if (ca == 1) hipLaunchKernelGGL((attn_fwd_group_kernel<1, 2, false, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); \
            else hipLaunchKernelGGL((attn_fwd_group_kernel<2, 2, false, G_>), dim3(B, splits), dim3(256), 0, s, u, v, ah, w_a, b_a, rows, lens, m, g, Nn, ctx, ldctx, alpha, n_stride, A, R, ctx_b16, qs); } \
    } while (0)
    SUBGC_G_DISPATCH(SUBGC_FWD_G);
#undef SUBGC_FWD_G
    return subgc::check_launch("subgc_attn_fwd_group");
}
}  // namespace

SUBGC_API int subgc_attn_fwd_group(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* rows,
                                   const int32_t* lens, int m, int B, int g, int Nn, void* ctx, int64_t ldctx, float* alpha, int n_stride, int A,
                                   int R, int bf16_bits, void* stream) {
    return attn_fwd_group_any(u, v, ah, w_a, b_a, rows, lens, m, B, g, Nn, ctx, ldctx, alpha, n_stride, A, R, bf16_bits, stream,
                              subgc::QSrc{nullptr, nullptr, 1, 0});
}
// the query of row r = q_bias + sum of n_planes planes (q_planes + p * plane_stride)[r, :] (the h2att product left as split-K partial planes,
// subgc_gemm_*_planes); the summed rows are written to q_out [m, A] for the backward
SUBGC_API int subgc_attn_fwd_group_q(const void* u, const void* v, const float* q_planes, int n_planes, int64_t plane_stride, const float* q_bias,
                                     float* q_out, const float* w_a, const float* b_a, const int32_t* rows, const int32_t* lens, int m, int B, int g,
                                     int Nn, void* ctx, int64_t ldctx, float* alpha, int n_stride, int A, int R, int bf16_bits, void* stream) {
    SUBGC_REQUIRE(n_planes >= 1 && n_planes <= 16 && q_out, "attn_fwd_group_q: 1 <= n_planes <= 16 and an output for the summed query");
    return attn_fwd_group_any(u, v, q_planes, w_a, b_a, rows, lens, m, B, g, Nn, ctx, ldctx, alpha, n_stride, A, R, bf16_bits, stream,
                              subgc::QSrc{q_bias, q_out, n_planes, plane_stride});
}

SUBGC_API int subgc_attn_bwd_group(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* rows, const int32_t* lens, int m,
                                   int B, int g, int Nn, const float* alpha, int n_stride, const float* dctx, int64_t lddctx, void* dah, float* du,
                                   float* dw_a, float* db_a, int A, int R, int bf16_bits, float* dctx_keep, int64_t ldkeep, int dctx_planes,
                                   int64_t plane_stride, int du_planes, int64_t du_plane_stride, void* stream) {
    const int dah_b16 = bf16_bits & 1, uv16 = (bf16_bits >> 1) & 1;
    const int n_planes = dctx_planes;
    SUBGC_REQUIRE(B >= 0 && g >= 1 && g <= 8 && Nn >= 1 && Nn <= GL && m >= 0 && A > 0 && R > 0 && n_stride > 0 && dctx_planes >= 1 && plane_stride % 4 == 0, "attn_bwd_group: bad sizes (g <= 8, Nn <= %d)", GL);
    if (B == 0 || m == 0) return SUBGC_OK;
    SUBGC_REQUIRE(u && v && ah && w_a && rows && lens && alpha && dctx && dah && du && dw_a, "attn_bwd_group: null pointer");
    SUBGC_REQUIRE(A % 4 == 0 && R % 4 == 0 && lddctx % 4 == 0 && ldkeep % 4 == 0 && al16(u) && al16(v) && al16(ah) && al16(w_a) && al16(dctx) && al16(dah) &&
                      al16(du) && al16(dw_a) && al16(dctx_keep), "attn_bwd_group: A, R %% 4 == 0 and 16-byte aligned rows");
    SUBGC_REQUIRE(!dctx_keep || ldkeep >= R, "attn_bwd_group: dctx_keep rows too short");
    const int ca = (A / 4 + 127) / 128, cr = (R / 4 + 63) / 64;
    SUBGC_REQUIRE(ca <= 2 && cr <= 4, "attn_bwd_group: att_hid_size <= 1024 and rnn_size <= 1024");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    const size_t bwd_lds = (size_t)8 * R * sizeof(float);                 // d(ctx) rows of up to 8 sentences
    SUBGC_REQUIRE(bwd_lds <= 32 * 1024, "attn_bwd_group: rnn_size <= 1024");
    const int splits = group_splits(g);
    SUBGC_REQUIRE(du_planes >= splits && (splits == 1 || du_plane_stride >= (int64_t)B * Nn * A), "attn_bwd_group: d(u) needs %d planes of B * Nn * A floats", splits);
#define SUBGC_BWD_ARGS u, v, ah, w_a, rows, lens, m, g, Nn, alpha, n_stride, dctx, lddctx, dah, du, dw_a, db_a, A, R, dah_b16, dctx_keep, ldkeep, n_planes, plane_stride, du_plane_stride
#define SUBGC_BWD_G(G_)                                                                                                                               \
    do {                                                                                                                                                \
        if (uv16) { if (ca == 1 && cr <= 2) hipLaunchKernelGGL((attn_bwd_group_kernel<1, 2, true, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);     \
            else if (ca == 1) hipLaunchKernelGGL((attn_bwd_group_kernel<1, 4, true, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);                  \
            else if (cr <= 2) hipLaunchKernelGGL((attn_bwd_group_kernel<2, 2, true, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);                  \
            else hipLaunchKernelGGL((attn_bwd_group_kernel<2, 4, true, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS); }                             \
        else { if (ca == 1 && cr <= 2) hipLaunchKernelGGL((attn_bwd_group_kernel<1, 2, false, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);         \
            else if (ca == 1) hipLaunchKernelGGL((attn_bwd_group_kernel<1, 4, false, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);                 \
            else if (cr <= 2) hipLaunchKernelGGL((attn_bwd_group_kernel<2, 2, false, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS);                 \
            else hipLaunchKernelGGL((attn_bwd_group_kernel<2, 4, false, G_>), dim3(B, splits), dim3(256), bwd_lds, s, SUBGC_BWD_ARGS); }                            \
    } while (0)
    SUBGC_G_DISPATCH(SUBGC_BWD_G);
#undef SUBGC_BWD_G
#undef SUBGC_BWD_ARGS
    return subgc::check_launch("subgc_attn_bwd_group");
}

SUBGC_API int subgc_attn_dv_accum_group(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off, int T,
                                        const int32_t* rows, int B, int g, int Nn, float* dv, int R, void* stream) {
    SUBGC_REQUIRE(B >= 0 && g >= 1 && g <= 8 && Nn >= 1 && R > 0 && T >= 1 && n_stride > 0 && lddctx >= R, "attn_dv_accum_group: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(alpha && dctx && step_off && rows && dv, "attn_dv_accum_group: null pointer");
    SUBGC_REQUIRE(R % 4 == 0 && lddctx % 4 == 0 && al16(dctx) && al16(dv), "attn_dv_accum_group: R %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_ATTN, s, 0.0);
    const size_t per = (size_t)64 * 16 + (size_t)n_stride * 4;
    SUBGC_REQUIRE(T <= 64, "attn_dv_accum_group: at most 64 steps");
    const int tg = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)T * g, 1024), (size_t)(140 * 1024) / per));
    const size_t lds = (size_t)tg * per;
    if (int rc = subgc::raise_lds_cached((const void*)attn_dv_accum_group_kernel, lds, "attn_dv_accum_group")) return rc;
    hipLaunchKernelGGL(attn_dv_accum_group_kernel, dim3(B, (R / 4 + 63) / 64), dim3(256), lds, s, alpha, n_stride, dctx, lddctx, step_off, T, rows, g, Nn,
                       dv, R, tg);
    return subgc::check_launch("subgc_attn_dv_accum_group");
}

// how many d(u) planes subgc_attn_bwd_group accumulates into for groups of g sentences (the caller zeroes and finally adds them)
SUBGC_API int subgc_attn_group_du_planes(int g) { return group_splits(g); }
