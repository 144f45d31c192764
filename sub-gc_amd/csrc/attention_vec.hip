// Vectorised (float4) forms of the per-step attention kernels, used when A % 4 == 0, R % 4 == 0 and all
// row starts are 16-byte aligned (every Sub-GC preset; other widths are rejected at model construction).
//
// One workgroup (256 threads = 4 waves) per sentence.  The sets are tiny (<= 11 nodes on Sub_GC_Kar), so
// the kernels are latency-bound: everything is arranged so that each thread issues ALL of its loads for a
// phase before it consumes any of them (full unrolling over float4 chunks), instead of one dependent
// global round trip per loop iteration.
// Reference: AttModel.py:453-466 (forward), its autograd backward.
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

constexpr int MAXLEN = 512;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// four consecutive elements at index i of a node-feature array that is fp32 or (UV16: compute_dtype = bf16, where u and v are
// kept only as the bf16 tensors the GEMMs wrote) bf16
template <bool UV16>
__device__ __forceinline__ float4 ldx(const void* base, int64_t i) {
    return UV16 ? subgc_load4_bf(static_cast<const uint16_t*>(base) + i) : ld4(static_cast<const float*>(base) + i);
}

// the same four elements as they lie in memory (a uint2 of bf16 pairs, or the float4 itself): what a thread holds while several rows are
// in flight -- the conversion happens when a row is used, so a bf16 row in flight costs two registers, not four
template <bool UV16> struct RawOf { using type = float4; };
template <> struct RawOf<true> { using type = uint2; };
template <bool UV16>
__device__ __forceinline__ typename RawOf<UV16>::type ldraw(const void* base, int64_t i) {
    if constexpr (UV16) return *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + i);
    else return ld4(static_cast<const float*>(base) + i);
}
__device__ __forceinline__ float4 cvt4(const float4& r) { return r; }
__device__ __forceinline__ float4 cvt4(const uint2& q) {
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
}

// CA = ceil(A/4 / 64): float4 chunks of a score row per lane;  CR = ceil(R/4 / 256): float4 chunks of a value row per thread
template <int CA, int CR, bool UV16>
__global__ __launch_bounds__(256) void attn_fwd_vec_kernel(const void* __restrict__ u, const void* __restrict__ v,
                                                           const float* __restrict__ ah, const float* __restrict__ w_a,
                                                           const float* __restrict__ b_a, const int32_t* __restrict__ off,
                                                           const int32_t* __restrict__ len, void* __restrict__ ctx, int64_t ldctx,
                                                           float* __restrict__ alpha, int n_stride, int A, int R, int ctx_b16, const subgc::QSrc qs) {
    __shared__ float e_s[MAXLEN];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l = min(len[s], MAXLEN), m0 = off[s];
    const int A4 = A >> 2, R4 = R >> 2;
    // scores: one wave per node; the query slice and w_a of this lane are loaded once
    float4 q[CA], w[CA];
#pragma unroll
    for (int c = 0; c < CA; ++c) {
        const int a4 = lane + c * 64;
        const bool ok = a4 < A4;
        q[c] = ok ? subgc_load_q(ah, qs, (int64_t)s * A + a4 * 4, a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (qs.out && ok && wave == 0) st4(qs.out + (int64_t)s * A + a4 * 4, q[c]);     // the summed query, kept for the backward
        w[c] = ok ? ld4(w_a + a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // four of this wave's nodes at a time: their rows are requested together and their scores reduced over the lanes TOGETHER (four
    // overlapping exchange chains instead of four dependent ones)
    constexpr int NCH = 4;
    const float ba = b_a[0];
    for (int i0 = wave; i0 < l; i0 += 4 * NCH) {
        float4 x[NCH][CA];
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int c = 0; c < CA; ++c)
                x[k][c] = (i0 + 4 * k < l && lane + c * 64 < A4) ? ldx<UV16>(u, (int64_t)(m0 + i0 + 4 * k) * A + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float sc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            float acc = 0.f;
            if (i0 + 4 * k < l) {
#pragma unroll
                for (int c = 0; c < CA; ++c)
                    acc += w[c].x * subgc_tanh(x[k][c].x + q[c].x) + w[c].y * subgc_tanh(x[k][c].y + q[c].y) + w[c].z * subgc_tanh(x[k][c].z + q[c].z) +
                           w[c].w * subgc_tanh(x[k][c].w + q[c].w);
            }
            sc[k] = acc;
        }
        wave_sum_n<NCH>(sc);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (i0 + 4 * k < l) e_s[i0 + 4 * k] = sc[k] + ba;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = 0; i < l; ++i) mx = fmaxf(mx, e_s[i]);
    __syncthreads();
    // every exponential ONCE (it used to be evaluated by all 256 threads inside the sum: a quarter of the kernel's VALU instructions on
    // 37-node sets); the sum below adds the same values in the same order, so the weights are bit for bit what they were
    for (int i = t; i < l; i += 256) e_s[i] = expf(e_s[i] - mx);
    __syncthreads();
    float den = 0.f;
    for (int i = 0; i < l; ++i) den += e_s[i];
    __syncthreads();
    for (int i = t; i < l; i += 256) e_s[i] = e_s[i] / den;
    __syncthreads();
    if (alpha)
        for (int i = t; i < n_stride; i += 256) alpha[(int64_t)s * n_stride + i] = i < l ? e_s[i] : 0.f;
    // context: thread = float4 column chunk; 4 node rows in flight at a time
#pragma unroll
    for (int c = 0; c < CR; ++c) {
        const int r4 = t + c * 256;
        if (r4 >= R4) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t vp = (int64_t)m0 * R + r4 * 4;
        int i = 0;
        for (; i + 4 <= l; i += 4) {
            const float4 x0 = ldx<UV16>(v, vp + (int64_t)(i + 0) * R), x1 = ldx<UV16>(v, vp + (int64_t)(i + 1) * R);
            const float4 x2 = ldx<UV16>(v, vp + (int64_t)(i + 2) * R), x3 = ldx<UV16>(v, vp + (int64_t)(i + 3) * R);
            const float a0 = e_s[i], a1 = e_s[i + 1], a2 = e_s[i + 2], a3 = e_s[i + 3];
            acc.x += a0 * x0.x; acc.y += a0 * x0.y; acc.z += a0 * x0.z; acc.w += a0 * x0.w;
            acc.x += a1 * x1.x; acc.y += a1 * x1.y; acc.z += a1 * x1.z; acc.w += a1 * x1.w;
            acc.x += a2 * x2.x; acc.y += a2 * x2.y; acc.z += a2 * x2.z; acc.w += a2 * x2.w;
            acc.x += a3 * x3.x; acc.y += a3 * x3.y; acc.z += a3 * x3.z; acc.w += a3 * x3.w;
        }
        for (; i < l; ++i) {
            const float4 x0 = ldx<UV16>(v, vp + (int64_t)i * R);
            const float a0 = e_s[i];
            acc.x += a0 * x0.x; acc.y += a0 * x0.y; acc.z += a0 * x0.z; acc.w += a0 * x0.w;
        }
        const float o[4] = {acc.x, acc.y, acc.z, acc.w};        // the context row is the lang-LSTM GEMM's operand: bf16 when asked
        subgc_store_act<4>(ctx, (int64_t)s * ldctx + r4 * 4, o, ctx_b16);
    }
}

// LEAN: d(u) and d(v) are both deferred (du == dv == NULL: every launch of the product) -- their read-modify-write paths are compiled out
// and their registers with them
template <int CA, int CR64, bool UV16, bool LEAN>
__device__ __forceinline__ void attn_bwd_vec_body(const void* __restrict__ u, const void* __restrict__ v,
                                                           const float* __restrict__ ah, const float* __restrict__ w_a,
                                                           const int32_t* __restrict__ off, const int32_t* __restrict__ len,
                                                           const float* __restrict__ alpha, int n_stride,
                                                           const float* __restrict__ dctx, int64_t lddctx, void* __restrict__ dah,
                                                           float* __restrict__ du, float* __restrict__ dv, float* __restrict__ dw_a,
                                                           float* __restrict__ db_a, int A, int R, int dah_b16,
                                                           float* __restrict__ dctx_keep, int64_t ldkeep, int n_planes, int64_t plane_stride,
                                                           float* __restrict__ de_keep) {
    // de_keep != NULL (then du == NULL): d(u) is deferred too -- this step only files its d(e) row (de_keep [S, n_stride]); after the time
    // loop subgc_attn_du_accum forms d(u) from the kept d(e) and query rows of all steps.  Read-modify-writing all of d(u) at every step
    // is the largest memory stream of this kernel when the sets are long (Full-GC: 36 rows x 512 x 8 bytes per sentence and step).
    // n_planes > 1: d(ctx) arrives as the split-K partial planes of the data-gradient GEMM (dctx + q * plane_stride), summed on load
    // dv == NULL: d(v) is deferred -- the caller keeps every step's d(ctx) rows (dctx_keep, written here by wave 0) and alpha
    // and calls subgc_attn_dv_accum once after the time loop instead of read-modify-writing all of d(v) at every step
    __shared__ float al_s[MAXLEN];    // alpha, then de
    __shared__ float da_s[MAXLEN];    // dalpha
    __shared__ float4 part_d[128], part_w[128];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l = min(len[s], MAXLEN), m0 = off[s];
    const int A4 = A >> 2, R4 = R >> 2;
    for (int i = t; i < l; i += 256) al_s[i] = alpha[(int64_t)s * n_stride + i];
    // dalpha_i = <dctx, v_i>,  dv_i += alpha_i dctx : wave per node; this lane's dctx chunks are loaded once
    float4 g[CR64];
#pragma unroll
    for (int c = 0; c < CR64; ++c) {
        g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane + c * 64 < R4) {
            // the planes of a chunk are requested together (see subgc_load_q); added in plane order
            constexpr int MAXQ = 4;
            float4 x[MAXQ];
            const float* base = dctx + (int64_t)s * lddctx + (lane + c * 64) * 4;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) x[q] = ld4(base + (q < n_planes ? q : n_planes - 1) * plane_stride);
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
                if (q < n_planes) { g[c].x += x[q].x; g[c].y += x[q].y; g[c].z += x[q].z; g[c].w += x[q].w; }
            for (int q = MAXQ; q < n_planes; ++q) {
                const float4 y = ld4(base + q * plane_stride);
                g[c].x += y.x; g[c].y += y.y; g[c].z += y.z; g[c].w += y.w;
            }
        }
    }
    if (dctx_keep && wave == 0) {
#pragma unroll
        for (int c = 0; c < CR64; ++c)
            if (lane + c * 64 < R4) st4(dctx_keep + (int64_t)s * ldkeep + (lane + c * 64) * 4, g[c]);
    }
    __syncthreads();
    // NCH of this wave's nodes at a time: rows requested together, the NCH dot products reduced over the lanes together
    constexpr int NCH = CR64 <= 2 ? 4 : (CR64 <= 4 ? (UV16 ? 3 : 2) : 1);
    for (int i0 = wave; i0 < l; i0 += 4 * NCH) {
        typename RawOf<UV16>::type x[NCH][CR64];
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int c = 0; c < CR64; ++c)
                x[k][c] = ldraw<UV16>(v, (int64_t)(m0 + min(i0 + 4 * k, l - 1)) * R + min(lane + c * 64, R4 - 1) * 4);
        float sc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int i = i0 + 4 * k;
            float acc = 0.f;
            if (i < l) {
                if (!LEAN && dv) {
                    const float a_i = al_s[i];
                    float* dvr = dv + (int64_t)(m0 + i) * R;
                    float4 y[CR64];
#pragma unroll
                    for (int c = 0; c < CR64; ++c) y[c] = (lane + c * 64 < R4) ? ld4(dvr + (lane + c * 64) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < CR64; ++c) {
                        y[c].x += a_i * g[c].x; y[c].y += a_i * g[c].y; y[c].z += a_i * g[c].z; y[c].w += a_i * g[c].w;
                        if (lane + c * 64 < R4) st4(dvr + (lane + c * 64) * 4, y[c]);
                    }
                }
#pragma unroll
                for (int c = 0; c < CR64; ++c) {
                    const float4 xv = cvt4(x[k][c]);            // g is zero in the chunks past R: a clamped load there adds nothing
                    acc += g[c].x * xv.x + g[c].y * xv.y + g[c].z * xv.z + g[c].w * xv.w;
                }
            }
            sc[k] = acc;
        }
        wave_sum_n<NCH>(sc);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (i0 + 4 * k < l) da_s[i0 + 4 * k] = sc[k];
        }
    }
    constexpr int UN = LEAN ? (UV16 ? 8 : 6) : 4;
    const int grp = t >> 7;
    typename RawOf<UV16>::type xu[UN] = {};
    if (l > 0) {                                            // (an empty set has no row to request)
        const int a40 = min(t & 127, A4 - 1);
#pragma unroll
        for (int k = 0; k < UN; ++k) xu[k] = ldraw<UV16>(u, (int64_t)(m0 + min(grp + 2 * k, l - 1)) * A + a40 * 4);
    }
    __syncthreads();
    float dot = 0.f;
    for (int i = 0; i < l; ++i) dot += al_s[i] * da_s[i];
    __syncthreads();
    for (int i = t; i < l; i += 256) {
        const float de = al_s[i] * (da_s[i] - dot);                            // de_i
        al_s[i] = de;
        if (de_keep) de_keep[(int64_t)s * n_stride + i] = de;
    }
    __syncthreads();
    if (t == 0 && db_a) {
        float desum = 0.f;
        for (int i = 0; i < l; ++i) desum += al_s[i];
        db_a[s] = desum;
    }
    // through tanh: 2 thread groups x 128 float4 chunks of the hidden dimension; group g takes nodes i = g, g+2, ...
    // UN of a group's nodes at a time: their u (and d(u)) rows are requested together, the sums keep the node order.  LEAN (nothing of
    // d(u) is touched here): the rows stay in memory format until used, so eight bf16 / six fp32 rows are in flight -- and the first batch
    // was requested before the softmax backward above (it depends on nothing that section computes)
#pragma unroll
    for (int c = 0; c < CA; ++c) {
        const int a4 = (t & 127) + c * 128;
        float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f), wsum = dsum;
        if (a4 < A4) {
            const float4 wa = ld4(w_a + a4 * 4), ha = ld4(ah + (int64_t)s * A + a4 * 4);
            for (int i0 = grp; i0 < l; i0 += 2 * UN) {
                float4 d[LEAN ? 1 : UN];
                if (c > 0 || i0 > grp) {
#pragma unroll
                    for (int k = 0; k < UN; ++k) xu[k] = ldraw<UV16>(u, (int64_t)(m0 + min(i0 + 2 * k, l - 1)) * A + a4 * 4);
                }
                if constexpr (!LEAN) {
#pragma unroll
                    for (int k = 0; k < UN; ++k)
                        d[k] = du ? ld4(du + (int64_t)(m0 + min(i0 + 2 * k, l - 1)) * A + a4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int i = i0 + 2 * k;
                    if (i >= l) break;
                    const float4 x = cvt4(xu[k]);
                    const float de = al_s[i];
                    const float t0 = subgc_tanh(x.x + ha.x), t1 = subgc_tanh(x.y + ha.y), t2 = subgc_tanh(x.z + ha.z), t3 = subgc_tanh(x.w + ha.w);
                    const float p0 = de * wa.x * (1.f - t0 * t0), p1 = de * wa.y * (1.f - t1 * t1);
                    const float p2 = de * wa.z * (1.f - t2 * t2), p3 = de * wa.w * (1.f - t3 * t3);
                    if constexpr (!LEAN) {
                        if (du) {
                            d[k].x += p0; d[k].y += p1; d[k].z += p2; d[k].w += p3;
                            st4(du + (int64_t)(m0 + i) * A + a4 * 4, d[k]);
                        }
                    }
                    dsum.x += p0; dsum.y += p1; dsum.z += p2; dsum.w += p3;
                    wsum.x += de * t0; wsum.y += de * t1; wsum.z += de * t2; wsum.w += de * t3;
                }
            }
        }
        __syncthreads();
        if (grp == 1 && (t & 127) < 128) { part_d[t & 127] = dsum; part_w[t & 127] = wsum; }
        __syncthreads();
        if (grp == 0 && a4 < A4) {
            const float4 od = part_d[t & 127], ow = part_w[t & 127];
            dsum.x += od.x; dsum.y += od.y; dsum.z += od.z; dsum.w += od.w;
            wsum.x += ow.x; wsum.y += ow.y; wsum.z += ow.z; wsum.w += ow.w;
            const float o[4] = {dsum.x, dsum.y, dsum.z, dsum.w};
            subgc_store_act<4>(dah, (int64_t)s * A + a4 * 4, o, dah_b16);
            st4(dw_a + (int64_t)s * A + a4 * 4, wsum);      // per-sentence partial; the caller column-sums once over all steps
        }
    }
}

// The general form, as the compiler allots registers (114-124 with 1000-wide value rows and the read-modify-write paths: four workgroups
// per CU); the deferred form over bf16 sets is additionally held to 96 registers so that FIVE workgroups share a CU -- Full-GC's 1280
// sentence rows are then resident at once instead of 1024 + a second round (fp32, deferred: 73 registers without being asked).
template <int CA, int CR64, bool UV16, bool LEAN>
__global__ __launch_bounds__(256) void attn_bwd_vec_kernel(const void* __restrict__ u, const void* __restrict__ v,
                                                           const float* __restrict__ ah, const float* __restrict__ w_a,
                                                           const int32_t* __restrict__ off, const int32_t* __restrict__ len,
                                                           const float* __restrict__ alpha, int n_stride,
                                                           const float* __restrict__ dctx, int64_t lddctx, void* __restrict__ dah,
                                                           float* __restrict__ du, float* __restrict__ dv, float* __restrict__ dw_a,
                                                           float* __restrict__ db_a, int A, int R, int dah_b16,
                                                           float* __restrict__ dctx_keep, int64_t ldkeep, int n_planes, int64_t plane_stride,
                                                           float* __restrict__ de_keep) {
    attn_bwd_vec_body<CA, CR64, UV16, LEAN>(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dah, du, dv, dw_a, db_a, A, R, dah_b16, dctx_keep, ldkeep, n_planes, plane_stride, de_keep);
}
template <int CA, int CR64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void attn_bwd_vec_b16_kernel(const void* __restrict__ u, const void* __restrict__ v,
                                                           const float* __restrict__ ah, const float* __restrict__ w_a,
                                                           const int32_t* __restrict__ off, const int32_t* __restrict__ len,
                                                           const float* __restrict__ alpha, int n_stride,
                                                           const float* __restrict__ dctx, int64_t lddctx, void* __restrict__ dah,
                                                           float* __restrict__ du, float* __restrict__ dv, float* __restrict__ dw_a,
                                                           float* __restrict__ db_a, int A, int R, int dah_b16,
                                                           float* __restrict__ dctx_keep, int64_t ldkeep, int n_planes, int64_t plane_stride,
                                                           float* __restrict__ de_keep) {
    attn_bwd_vec_body<CA, CR64, true, true>(u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dah, du, dv, dw_a, db_a, A, R, dah_b16, dctx_keep, ldkeep, n_planes, plane_stride, de_keep);
}

// d(v) of ALL time steps in one pass: dv[m0 + i, :] = sum over the steps t at which sentence s is live of alpha_t[s, i] *
// dctx_t[s, :].  One workgroup per sentence; the sentence's d(ctx) rows and attention weights of up to `tg` steps are staged
// in LDS once (17 steps x 4 KB on Sub-GC) and every node row is then written exactly once -- instead of reading and writing all
// of d(v) at every step (Full-GC, 36 nodes per sentence: 380 of the 830 MB a step's attention backward moved).
// Step t holds its live sentences as rows step_off[t] .. step_off[t+1]-1 of alpha / dctx (sentence s live iff s < the count):
// the packed decoder's layout; the unpacked one is step_off[t] = t * S.
template <int CR64>
__global__ __launch_bounds__(256) void attn_dv_accum_kernel(const float* __restrict__ alpha, int n_stride, const float* __restrict__ dctx,
                                                            int64_t lddctx, const int32_t* __restrict__ step_off, int T,
                                                            const int32_t* __restrict__ off, const int32_t* __restrict__ len,
                                                            float* __restrict__ dv, int R, int tg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dv_lds[];
    const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l = min(len[s], MAXLEN), m0 = off[s], R4 = R >> 2;
    float4* g_s = reinterpret_cast<float4*>(dv_lds);                      // [tg][R4]
    float* a_s = reinterpret_cast<float*>(g_s + (size_t)tg * R4);         // [tg][n_stride]
    bool first = true;
    for (int t0 = 0; t0 < T; t0 += tg) {
        __syncthreads();                                                   // the previous group has been consumed
        int nl = 0;                                                        // live steps of this group (uniform over the workgroup)
        for (int tt = t0; tt < min(T, t0 + tg); ++tt) {
            const int o = step_off[tt], m = step_off[tt + 1] - o;
            if (s >= m) continue;
            const int64_t flat = (int64_t)o + s;
            for (int c = t; c < R4; c += 256) g_s[(size_t)nl * R4 + c] = ld4(dctx + flat * lddctx + c * 4);
            for (int i = t; i < l; i += 256) a_s[nl * n_stride + i] = alpha[flat * n_stride + i];
            ++nl;
        }
        __syncthreads();
        if (nl == 0 && !first) continue;
        for (int i = wave; i < l; i += 4) {
            float4 acc[CR64];
#pragma unroll
            for (int c = 0; c < CR64; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= nl; k += 4) {                              // four steps' LDS reads in flight (the serial form waited for each)
                float a[4];
                float4 g[4][CR64];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = a_s[(k + q) * n_stride + i];
#pragma unroll
                    for (int c = 0; c < CR64; ++c) g[q][c] = (lane + c * 64 < R4) ? g_s[(size_t)(k + q) * R4 + lane + c * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < CR64; ++c) { acc[c].x += a[q] * g[q][c].x; acc[c].y += a[q] * g[q][c].y; acc[c].z += a[q] * g[q][c].z; acc[c].w += a[q] * g[q][c].w; }
            }
            for (; k < nl; ++k) {
                const float a = a_s[k * n_stride + i];
#pragma unroll
                for (int c = 0; c < CR64; ++c) {
                    if (lane + c * 64 >= R4) continue;
                    const float4 g = g_s[(size_t)k * R4 + lane + c * 64];
                    acc[c].x += a * g.x; acc[c].y += a * g.y; acc[c].z += a * g.z; acc[c].w += a * g.w;
                }
            }
            float* dvr = dv + (int64_t)(m0 + i) * R;
#pragma unroll
            for (int c = 0; c < CR64; ++c) {
                if (lane + c * 64 >= R4) continue;
                if (!first) { const float4 o = ld4(dvr + (lane + c * 64) * 4); acc[c].x += o.x; acc[c].y += o.y; acc[c].z += o.z; acc[c].w += o.w; }
                st4(dvr + (lane + c * 64) * 4, acc[c]);
            }
        }
        first = false;
    }
}

// d(u) of ALL time steps in one pass (deferred form of attn_bwd's accumulation):
//   du[m0 + i, a] = w_a[a] * sum over the steps t at which sentence s is live of de_t[s, i] * (1 - tanh^2(u[m0 + i, a] + ah_t[s, a]))
// One workgroup per sentence; the live steps' query rows ah_t[s, :] and d(e) rows are staged in LDS `tg` steps at a time; thread
// (a4 = t % 128 (+128 c), grp = t / 128) owns four hidden columns and every second node.  Each d(u) row is written once (per step
// group) instead of read and written at every step; the price is one more tanh per (step, node, column) -- VALU work the chip has to spare.
template <int CA, bool UV16>
__global__ __launch_bounds__(256) void attn_du_accum_kernel(const void* __restrict__ u, const float* __restrict__ ah, const float* __restrict__ de,
                                                            int n_stride, const int32_t* __restrict__ step_off, int T,
                                                            const int32_t* __restrict__ off, const int32_t* __restrict__ len,
                                                            const float* __restrict__ w_a, float* __restrict__ du, int A, int tg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char du_lds[];
    const int s = blockIdx.x, t = threadIdx.x;
    const int l = min(len[s], MAXLEN), m0 = off[s], A4 = A >> 2;
    float4* h_s = reinterpret_cast<float4*>(du_lds);                      // [tg][A4]
    float* e_s = reinterpret_cast<float*>(h_s + (size_t)tg * A4);         // [tg][n_stride]
    bool first = true;
    for (int t0 = 0; t0 < T; t0 += tg) {
        __syncthreads();
        int nl = 0;
        for (int tt = t0; tt < min(T, t0 + tg); ++tt) {
            const int o = step_off[tt], m = step_off[tt + 1] - o;
            if (s >= m) continue;
            const int64_t flat = (int64_t)o + s;
            for (int c = t; c < A4; c += 256) h_s[(size_t)nl * A4 + c] = ld4(ah + flat * A + c * 4);
            for (int i = t; i < l; i += 256) e_s[nl * n_stride + i] = de[flat * n_stride + i];
            ++nl;
        }
        __syncthreads();
        if (nl == 0 && !first) continue;
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int a4 = (t & 127) + c * 128, grp = t >> 7;
            if (a4 >= A4) continue;
            const float4 wa = ld4(w_a + a4 * 4);
            for (int i = grp; i < l; i += 2) {
                const int64_t o = (int64_t)(m0 + i) * A + a4 * 4;
                const float4 x = ldx<UV16>(u, o);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < nl; ++k) {
                    const float4 ha = h_s[(size_t)k * A4 + a4];
                    const float e = e_s[k * n_stride + i];
                    const float t0_ = subgc_tanh(x.x + ha.x), t1 = subgc_tanh(x.y + ha.y), t2 = subgc_tanh(x.z + ha.z), t3 = subgc_tanh(x.w + ha.w);
                    acc.x += e * (1.f - t0_ * t0_); acc.y += e * (1.f - t1 * t1); acc.z += e * (1.f - t2 * t2); acc.w += e * (1.f - t3 * t3);
                }
                acc.x *= wa.x; acc.y *= wa.y; acc.z *= wa.z; acc.w *= wa.w;
                if (!first) { const float4 p = ld4(du + o); acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
                st4(du + o, acc);
            }
        }
        first = false;
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace subgc {

// return -100 when the vector form does not apply
int attn_fwd_vec(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* off,
                 const int32_t* len, void* ctx, int64_t ldctx, float* alpha, int n_stride, int S, int A, int R, int ctx_b16, int uv_b16, hipStream_t s,
                 QSrc qs) {
    if (A % 4 || R % 4 || ldctx % 4 || !al16(u) || !al16(v) || !al16(ah) || !al16(w_a) || !al16(ctx) || qs.stride % 4 || !al16(qs.bias) || !al16(qs.out))
        return -100;
    const int ca = (A / 4 + 63) / 64, cr = (R / 4 + 255) / 256;
    if (ca > 2 || cr > 2) return -100;
#define SUBGC_ATT_FWD(CA_, CR_)                                                                                                          \
    do {                                                                                                                                   \
        if (uv_b16) hipLaunchKernelGGL((attn_fwd_vec_kernel<CA_, CR_, true>), dim3(S), dim3(256), 0, s, u, v, ah, w_a, b_a, off, len, ctx, ldctx, \
                                       alpha, n_stride, A, R, ctx_b16, qs);                                                               \
        else hipLaunchKernelGGL((attn_fwd_vec_kernel<CA_, CR_, false>), dim3(S), dim3(256), 0, s, u, v, ah, w_a, b_a, off, len, ctx, ldctx, \
                                alpha, n_stride, A, R, ctx_b16, qs);                                                                      \
    } while (0)
    if (ca == 1 && cr == 1) SUBGC_ATT_FWD(1, 1);
    else if (ca == 2 && cr == 1) SUBGC_ATT_FWD(2, 1);
    else if (ca == 1 && cr == 2) SUBGC_ATT_FWD(1, 2);
    else SUBGC_ATT_FWD(2, 2);
#undef SUBGC_ATT_FWD
    return check_launch("subgc_attn_fwd(vec)");
}

int attn_bwd_vec(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                 const float* alpha, int n_stride, const float* dctx, int64_t lddctx, void* dah, float* du, float* dv, float* dw_a,
                 float* db_a, int S, int A, int R, int dah_b16, int uv_b16, float* dctx_keep, int64_t ldkeep, hipStream_t s, int n_planes,
                 int64_t plane_stride, float* de_keep) {
    if (A % 4 || R % 4 || lddctx % 4 || ldkeep % 4 || plane_stride % 4 || !al16(u) || !al16(v) || !al16(ah) || !al16(w_a) || !al16(dctx) || !al16(dah) || !al16(du) ||
        !al16(dv) || !al16(dw_a) || !al16(dctx_keep))
        return -100;
    const int ca = (A / 4 + 127) / 128, cr = (R / 4 + 63) / 64;
    if (ca > 2 || cr > 8) return -100;
    const bool lean = !du && !dv;
#define SUBGC_ATT_BWD2(K_, ...)                                                                                                           \
    hipLaunchKernelGGL((K_<__VA_ARGS__>), dim3(S), dim3(256), 0, s, u, v, ah, w_a, off, len, alpha, n_stride, dctx, lddctx, dah, du, dv, dw_a, db_a, A, \
                       R, dah_b16, dctx_keep, ldkeep, n_planes, plane_stride, de_keep)
#define SUBGC_ATT_BWD(CA_, CR_)                                                                                                          \
    do {                                                                                                                                   \
        if (uv_b16) { if (lean) SUBGC_ATT_BWD2(attn_bwd_vec_b16_kernel, CA_, CR_); else SUBGC_ATT_BWD2(attn_bwd_vec_kernel, CA_, CR_, true, false); } \
        else { if (lean) SUBGC_ATT_BWD2(attn_bwd_vec_kernel, CA_, CR_, false, true); else SUBGC_ATT_BWD2(attn_bwd_vec_kernel, CA_, CR_, false, false); } \
    } while (0)
    if (ca == 1) {
        if (cr <= 1) SUBGC_ATT_BWD(1, 1); else if (cr <= 2) SUBGC_ATT_BWD(1, 2); else if (cr <= 4) SUBGC_ATT_BWD(1, 4); else SUBGC_ATT_BWD(1, 8);
    } else {
        if (cr <= 1) SUBGC_ATT_BWD(2, 1); else if (cr <= 2) SUBGC_ATT_BWD(2, 2); else if (cr <= 4) SUBGC_ATT_BWD(2, 4); else SUBGC_ATT_BWD(2, 8);
    }
#undef SUBGC_ATT_BWD
#undef SUBGC_ATT_BWD2
    return check_launch("subgc_attn_bwd(vec)");
}

// -100 when the float4 form does not apply (the caller then has to accumulate d(v) inside subgc_attn_bwd)
int attn_dv_accum_vec(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off, int T, const int32_t* off,
                      const int32_t* len, float* dv, int S, int R, hipStream_t s) {
    if (R % 4 || lddctx % 4 || !al16(dctx) || !al16(dv)) return -100;
    const int cr = (R / 4 + 63) / 64;
    if (cr > 8) return -100;
    const size_t per_step = (size_t)R * 4 + (size_t)n_stride * 4;
    const int tg = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, (size_t)(72 * 1024) / per_step));   // <= 72 KB: two workgroups per CU
    const size_t lds = (size_t)tg * per_step;
    if (lds > 150 * 1024) return -100;
#define SUBGC_ATT_DV(CR_)                                                                                                                 \
    do {                                                                                                                                  \
        if (int rc = raise_lds_cached((const void*)attn_dv_accum_kernel<CR_>, lds, "attn_dv_accum")) return rc;                              \
        hipLaunchKernelGGL((attn_dv_accum_kernel<CR_>), dim3(S), dim3(256), lds, s, alpha, n_stride, dctx, lddctx, step_off, T, off, len, dv, R, tg); \
    } while (0)
    if (cr <= 1) SUBGC_ATT_DV(1); else if (cr <= 2) SUBGC_ATT_DV(2); else if (cr <= 4) SUBGC_ATT_DV(4); else SUBGC_ATT_DV(8);
#undef SUBGC_ATT_DV
    return check_launch("subgc_attn_dv_accum");
}

// -100 when the float4 form does not apply
int attn_du_accum_vec(const void* u, int uv_b16, const float* ah, const float* de, int n_stride, const int32_t* step_off, int T, const int32_t* off,
                      const int32_t* len, const float* w_a, float* du, int S, int A, hipStream_t s) {
    if (A % 4 || !al16(u) || !al16(ah) || !al16(w_a) || !al16(du)) return -100;
    const int ca = (A / 4 + 127) / 128;
    if (ca > 2) return -100;
    const size_t per_step = (size_t)A * 4 + (size_t)n_stride * 4;
    const int tg = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, (size_t)(72 * 1024) / per_step));   // <= 72 KB: two workgroups per CU
    const size_t lds = (size_t)tg * per_step;
    if (lds > 150 * 1024) return -100;
#define SUBGC_ATT_DU(CA_, B16_)                                                                                                            \
    do {                                                                                                                                  \
        if (int rc = raise_lds_cached((const void*)attn_du_accum_kernel<CA_, B16_>, lds, "attn_du_accum")) return rc;                       \
        hipLaunchKernelGGL((attn_du_accum_kernel<CA_, B16_>), dim3(S), dim3(256), lds, s, u, ah, de, n_stride, step_off, T, off, len, w_a, du, A, tg); \
    } while (0)
    if (ca == 1) { if (uv_b16) SUBGC_ATT_DU(1, true); else SUBGC_ATT_DU(1, false); }
    else { if (uv_b16) SUBGC_ATT_DU(2, true); else SUBGC_ATT_DU(2, false); }
#undef SUBGC_ATT_DU
    return check_launch("subgc_attn_du_accum");
}

}  // namespace subgc
