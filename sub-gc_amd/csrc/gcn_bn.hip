// BatchNorm1d of the Full-GC collection units (graph_conv_unit.py:31-32, nn.BatchNorm1d(dim) on the [rows, L] unit outputs)
// FUSED into the GCN aggregation kernels that consume it (graph_conv_unit.py:34-36, graph_conv.py:26,33):
//   forward   ONE pass over the raw unit output y computes the batch statistics (per-slab shifted sums, merged with Chan's
//             formula in double), a finishing launch turns them into the per-column triple {mean, gamma * rstd, beta} and updates
//             the running statistics; the aggregation kernels apply (y - mean) * scale + beta ON LOAD -- the normalised tensor is
//             never written (was: two statistics passes + normalise = 3 reads + 1 write of y and 5 launches);
//   backward  the aggregation backward yields d(normalised); ONE reduce launch leaves per-slab {sum dy * xhat, sum dy}, ONE apply
//             launch adds the slabs in a fixed order and writes d(y) in y's storage type (bf16 when y is a bf16 GEMM result, so the
//             producing Linear's backward needs no cast pass) and d(gamma), d(beta) straight into their gradient slots.
// y may be fp32 or bf16 (SRC16): under compute_dtype = bf16 the unit's second GEMM writes bf16 only.
// All kernels are HBM-bound streams over [rows, L] with L % 4 == 0: a lane owns four adjacent columns (16-byte / 8-byte accesses).
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

constexpr int TC = 128;

template <bool SRC16>
__device__ __forceinline__ float4 ldsrc(const void* base, int64_t i) {
    return SRC16 ? subgc_load4_bf(static_cast<const uint16_t*>(base) + i) : *reinterpret_cast<const float4*>(static_cast<const float*>(base) + i);
}
struct Aff4 { float4 mu, sc, be; bool on; };
__device__ __forceinline__ Aff4 load_aff(const float* __restrict__ aff, int L, int col) {
    Aff4 a;
    a.on = aff != nullptr;
    if (a.on) {
        a.mu = *reinterpret_cast<const float4*>(aff + col);
        a.sc = *reinterpret_cast<const float4*>(aff + L + col);
        a.be = *reinterpret_cast<const float4*>(aff + 2 * L + col);
    }
    return a;
}
__device__ __forceinline__ float4 apply_aff(const Aff4& a, float4 x) {
    if (!a.on) return x;
    return make_float4((x.x - a.mu.x) * a.sc.x + a.be.x, (x.y - a.mu.y) * a.sc.y + a.be.y, (x.z - a.mu.z) * a.sc.z + a.be.z,
                       (x.w - a.mu.w) * a.sc.w + a.be.w);
}

// ------------------------------------------------------------------ statistics
// grid (C/256, slabs): part[slab][{sum (x - p), sum (x - p)^2, p}][C], p = the slab's first row (shifted-data sums)
template <bool SRC16>
__device__ __forceinline__ void bn_stats_body(const void* __restrict__ X, int M, int C, int rpb, float* __restrict__ part) {
    __shared__ float4 ss[4][64], sq[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s, p = s;
    if (col < C) {
        p = ldsrc<SRC16>(X, (int64_t)r0 * C + col);
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = ldsrc<SRC16>(X, (int64_t)(r + 4 * k) * C + col);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = x[k].x - p.x, d1 = x[k].y - p.y, d2 = x[k].z - p.z, d3 = x[k].w - p.w;
                s.x += d0; s.y += d1; s.z += d2; s.w += d3;
                q.x += d0 * d0; q.y += d1 * d1; q.z += d2 * d2; q.w += d3 * d3;
            }
        }
        for (; r < r1; r += 4) {
            const float4 x = ldsrc<SRC16>(X, (int64_t)r * C + col);
            const float d0 = x.x - p.x, d1 = x.y - p.y, d2 = x.z - p.z, d3 = x.w - p.w;
            s.x += d0; s.y += d1; s.z += d2; s.w += d3;
            q.x += d0 * d0; q.y += d1 * d1; q.z += d2 * d2; q.w += d3 * d3;
        }
    }
    ss[w][lane] = s; sq[w][lane] = q;
    __syncthreads();
    if (w == 0 && col < C) {
        const float4 a0 = ss[0][lane], a1 = ss[1][lane], a2 = ss[2][lane], a3 = ss[3][lane];
        const float4 b0 = sq[0][lane], b1 = sq[1][lane], b2 = sq[2][lane], b3 = sq[3][lane];
        float* o = part + (int64_t)blockIdx.y * 3 * C + col;
        *reinterpret_cast<float4*>(o) = make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
        *reinterpret_cast<float4*>(o + C) = make_float4(b0.x + b1.x + b2.x + b3.x, b0.y + b1.y + b2.y + b3.y, b0.z + b1.z + b2.z + b3.z, b0.w + b1.w + b2.w + b3.w);
        *reinterpret_cast<float4*>(o + 2 * C) = p;
    }
}
template <bool SRC16>
__global__ __launch_bounds__(256) void bn_stats_kernel(const void* __restrict__ X, int M, int C, int rpb, float* __restrict__ part) {
    bn_stats_body<SRC16>(X, M, C, rpb, part);
}
// the two units of a GCN pair in one launch (blockIdx.z = unit; its slab partials at part + z * slabs * 3 C)
struct BnPair { const void* X[2]; const float* gamma[2]; const float* beta[2]; float* rmean[2]; float* rvar[2]; float* aff[2]; float* rstd[2]; };
template <bool SRC16>
__global__ __launch_bounds__(256) void bn_stats_pair_kernel(BnPair a, int M, int C, int rpb, float* __restrict__ part, int slabs) {
    bn_stats_body<SRC16>(a.X[blockIdx.z], M, C, rpb, part + (size_t)blockIdx.z * slabs * 3 * C);
}
// grid C/64 x 1024 threads: wave w takes slabs w, w+16, ...  Two sweeps over the (L2-resident) partials, no division in the loops:
// the batch mean from the slab means, then M2 = sum_k [M2_k + n_k (mean_k - mean)^2] (Chan's formula for many groups), in double.
// A sweep requests FIN_U slabs' partials before it consumes any: with one load in flight per wave the kernel was a chain of L2
// latencies (128 slabs, 4 waves: 23.7 us -- 3.5 x the pass over the data it finishes).
constexpr int FIN_WAVES = 16, FIN_U = 8;
__device__ __forceinline__ void bn_stats_finish_body(const float* __restrict__ part, int slabs, int rpb, int M, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ aff, float* __restrict__ rstd_out, float momentum, float eps) {
    __shared__ double sacc[FIN_WAVES][64];
    __shared__ double smean[64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = min(blockIdx.x * 64 + lane, C - 1);
    const int n_last = M - (slabs - 1) * rpb;
    const double inv_full = 1.0 / (double)rpb, inv_last = 1.0 / (double)n_last;
    double acc = 0.0;
    for (int k0 = w; k0 < slabs; k0 += FIN_WAVES * FIN_U) {                // sum_k n_k mean_k = sum_k (n_k p_k + s_k)
        float s_[FIN_U], p_[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int k = min(k0 + u * FIN_WAVES, slabs - 1);
            const float* o = part + (int64_t)k * 3 * C + c;
            s_[u] = o[0]; p_[u] = o[2 * C];
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int k = k0 + u * FIN_WAVES;
            if (k < slabs) acc += (double)(k == slabs - 1 ? n_last : rpb) * (double)p_[u] + (double)s_[u];
        }
    }
    sacc[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < FIN_WAVES; ++i) t += sacc[i][lane];
        smean[lane] = t / (double)M;
    }
    __syncthreads();
    const double mean = smean[lane];
    acc = 0.0;
    for (int k0 = w; k0 < slabs; k0 += FIN_WAVES * FIN_U) {
        float s_[FIN_U], q_[FIN_U], p_[FIN_U];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int k = min(k0 + u * FIN_WAVES, slabs - 1);
            const float* o = part + (int64_t)k * 3 * C + c;
            s_[u] = o[0]; q_[u] = o[C]; p_[u] = o[2 * C];
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int k = k0 + u * FIN_WAVES;
            if (k < slabs) {
                const bool last = k == slabs - 1;
                const double n = last ? (double)n_last : (double)rpb, inv = last ? inv_last : inv_full;
                const double s = s_[u], q = q_[u], p = p_[u];
                const double d = p + s * inv - mean;
                acc += (q - s * s * inv) + n * d * d;
            }
        }
    }
    __syncthreads();
    sacc[w][lane] = acc;
    __syncthreads();
    if (w != 0 || blockIdx.x * 64 + lane >= C) return;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < FIN_WAVES; ++i) m2 += sacc[i][lane];
    const float meanf = (float)mean, var_b = (float)(m2 / (double)M);
    const float rs = 1.f / sqrtf(var_b + eps);
    aff[c] = meanf; aff[C + c] = gamma[c] * rs; aff[2 * C + c] = beta[c];
    rstd_out[c] = rs;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (M > 1 ? (float)(m2 / (double)(M - 1)) : var_b);
}
__global__ __launch_bounds__(FIN_WAVES * 64) void bn_stats_finish_kernel(const float* __restrict__ part, int slabs, int rpb, int M, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ aff, float* __restrict__ rstd_out, float momentum, float eps) {
    bn_stats_finish_body(part, slabs, rpb, M, C, gamma, beta, running_mean, running_var, aff, rstd_out, momentum, eps);
}
__global__ __launch_bounds__(FIN_WAVES * 64) void bn_stats_finish_pair_kernel(const float* __restrict__ part, int slabs, int rpb, int M, int C, BnPair a,
                                                                   float momentum, float eps) {
    const int z = blockIdx.y;
    bn_stats_finish_body(part + (size_t)z * slabs * 3 * C, slabs, rpb, M, C, a.gamma[z], a.beta[z], a.rmean[z], a.rvar[z], a.aff[z], a.rstd[z], momentum, eps);
}
__global__ __launch_bounds__(256) void bn_eval_aff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                          float* __restrict__ aff, float* __restrict__ rstd_out, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.f / sqrtf(running_var[c] + eps);
    aff[c] = running_mean[c]; aff[C + c] = gamma[c] * rs; aff[2 * C + c] = beta[c];
    if (rstd_out) rstd_out[c] = rs;
}

// ------------------------------------------------------------------ nodes <- relations with the unit outputs normalised on load
template <bool SRC16>
__global__ __launch_bounds__(256) void gcn_nodes_fwd_bn_kernel(const void* __restrict__ F0, const void* __restrict__ F1,
                                                               const float* __restrict__ aff0, const float* __restrict__ aff1,
                                                               const int32_t* __restrict__ ptr, const int32_t* __restrict__ edges,
                                                               const float* __restrict__ skip, float* __restrict__ Xout,
                                                               uint16_t* __restrict__ Xout16, uint8_t* __restrict__ act, int B, int N, int K, int L) {
    extern __shared__ int sm_i[];
    int* ps = sm_i; int* po = ps + (N + 1); int* es = po + (N + 1); int* eo = es + K;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
        ps[i] = ptr[((int64_t)0 * B + b) * (N + 1) + i];
        po[i] = ptr[((int64_t)1 * B + b) * (N + 1) + i];
    }
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        es[i] = edges[((int64_t)0 * B + b) * K + i];
        eo[i] = edges[((int64_t)1 * B + b) * K + i];
    }
    __syncthreads();
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= L) return;
    const Aff4 A0 = load_aff(aff0, L, col), A1 = load_aff(aff1, L, col);
    const int64_t base = (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        const int s0 = ps[n], s1 = ps[n + 1], o0 = po[n], o1 = po[n + 1];
        gather_pair(es, s0, s1, [&](int e) { return apply_aff(A0, ldsrc<SRC16>(F0, base + (int64_t)e * L)); },
                    eo, o0, o1, [&](int e) { return apply_aff(A1, ldsrc<SRC16>(F1, base + (int64_t)e * L)); }, a, c);
        const float da = (float)(s1 - s0) + 1e-7f, dc = (float)(o1 - o0) + 1e-7f;
        const float av[4] = {a.x / da, a.y / da, a.z / da, a.w / da}, cv[4] = {c.x / dc, c.y / dc, c.z / dc, c.w / dc};
        const int64_t o = ((int64_t)b * N + n) * L + col;
        float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
        if (skip) sk = *reinterpret_cast<const float4*>(skip + o);
        const float skv[4] = {sk.x, sk.y, sk.z, sk.w};
        float v[4];
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bits |= (uint32_t)((av[e] > 0.f ? 1 : 0) | (cv[e] > 0.f ? 2 : 0)) << (8 * e);
            v[e] = (fmaxf(av[e], 0.f) + fmaxf(cv[e], 0.f)) / 2.f;
            if (skip) v[e] += skv[e];
        }
        *reinterpret_cast<float4*>(Xout + o) = make_float4(v[0], v[1], v[2], v[3]);
        if (Xout16) *reinterpret_cast<uint2*>(Xout16 + o) = subgc_pack4(v[0], v[1], v[2], v[3]);
        if (act) *reinterpret_cast<uint32_t*>(act + o) = bits;
    }
}

// ------------------------------------------------------------------ relations <- nodes (node tiles staged in LDS, normalised while staging)
template <bool SRC16>
__global__ __launch_bounds__(256) void gcn_edges_fwd_bn_kernel(const void* __restrict__ F2, const void* __restrict__ F3,
                                                               const float* __restrict__ aff2, const float* __restrict__ aff3,
                                                               const int64_t* __restrict__ rel_ind, const float* __restrict__ skip,
                                                               float* __restrict__ Pout, uint16_t* __restrict__ Pout16, int B, int N, int K, int L, int vec) {
    extern __shared__ __attribute__((aligned(16))) float sm_f[];
    float* t2 = sm_f;
    float* t3 = sm_f + (size_t)N * TC;
    int* ns = reinterpret_cast<int*>(t3 + (size_t)N * TC);
    int* no = ns + K;
    const int b = blockIdx.y, c0 = blockIdx.x * TC;
    const float cdiv1 = 1.f + 1e-7f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int64_t s = rel_ind[((int64_t)b * K + k) * 2 + 0], o = rel_ind[((int64_t)b * K + k) * 2 + 1];
        ns[k] = (int)(s < 0 ? 0 : (s >= N ? N - 1 : s));
        no[k] = (int)(o < 0 ? 0 : (o >= N ? N - 1 : o));
    }
    const int items = N * (TC / 4);
    for (int i0 = threadIdx.x; i0 < items; i0 += 4 * blockDim.x) {         // four node-row quads requested before the first is staged
        float4 a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x;
            a[u] = make_float4(0.f, 0.f, 0.f, 0.f); c[u] = a[u];
            if (i < items) {
                const int n = i / (TC / 4), c4 = (i % (TC / 4)) * 4;
                if (c0 + c4 < L) {                                        // L % 4 == 0: a float4 is all in or all out
                    const int64_t g = ((int64_t)b * N + n) * L + c0 + c4;
                    a[u] = ldsrc<SRC16>(F2, g);
                    c[u] = ldsrc<SRC16>(F3, g);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < items) {
                const int n = i / (TC / 4), c4 = (i % (TC / 4)) * 4;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
                if (c0 + c4 < L) {
                    x = apply_aff(load_aff(aff2, L, c0 + c4), a[u]);
                    y = apply_aff(load_aff(aff3, L, c0 + c4), c[u]);
                }
                x.x = fmaxf(x.x / cdiv1, 0.f); x.y = fmaxf(x.y / cdiv1, 0.f); x.z = fmaxf(x.z / cdiv1, 0.f); x.w = fmaxf(x.w / cdiv1, 0.f);
                y.x = fmaxf(y.x / cdiv1, 0.f); y.y = fmaxf(y.y / cdiv1, 0.f); y.z = fmaxf(y.z / cdiv1, 0.f); y.w = fmaxf(y.w / cdiv1, 0.f);
                *reinterpret_cast<float4*>(t2 + n * TC + c4) = x;
                *reinterpret_cast<float4*>(t3 + n * TC + c4) = y;
            }
        }
    }
    __syncthreads();
    if (vec) {                                                             // 8 relation streams x 32 lanes x 4 columns: 16-byte LDS reads and stores
        const int cl4 = (threadIdx.x & 31) * 4, st = threadIdx.x >> 5;
        if (c0 + cl4 >= L) return;
        for (int k0 = st; k0 < K; k0 += 32) {
            float4 v[4], sk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 8 * u;
                if (k < K) {
                    const float4 x = *reinterpret_cast<const float4*>(t2 + ns[k] * TC + cl4), y = *reinterpret_cast<const float4*>(t3 + no[k] * TC + cl4);
                    v[u] = make_float4((x.x + y.x) / 2.f, (x.y + y.y) / 2.f, (x.z + y.z) / 2.f, (x.w + y.w) / 2.f);
                    if (skip) sk[u] = *reinterpret_cast<const float4*>(skip + ((int64_t)b * K + k) * L + c0 + cl4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 8 * u;
                if (k < K) {
                    float4 r = v[u];
                    if (skip) { r.x += sk[u].x; r.y += sk[u].y; r.z += sk[u].z; r.w += sk[u].w; }
                    const int64_t o = ((int64_t)b * K + k) * L + c0 + cl4;
                    *reinterpret_cast<float4*>(Pout + o) = r;
                    if (Pout16) *reinterpret_cast<uint2*>(Pout16 + o) = subgc_pack4(r.x, r.y, r.z, r.w);
                }
            }
        }
        return;
    }
    const int cl = threadIdx.x & (TC - 1), half = threadIdx.x >> 7;
    const int col = c0 + cl;
    if (col >= L) return;
    for (int k0 = half; k0 < K; k0 += 8) {                                 // four relations per round: LDS reads and skip loads in flight together
        float v[4], sk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 2 * u;
            if (k < K) {
                v[u] = (t2[ns[k] * TC + cl] + t3[no[k] * TC + cl]) / 2.f;
                sk[u] = skip ? skip[((int64_t)b * K + k) * L + col] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 2 * u;
            if (k < K) {
                const int64_t o = ((int64_t)b * K + k) * L + col;
                const float r = skip ? v[u] + sk[u] : v[u];
                Pout[o] = r;
                if (Pout16) Pout16[o] = (uint16_t)subgc_f2bf(r);
            }
        }
    }
}

template <bool SRC16>
__global__ __launch_bounds__(256) void gcn_edges_bwd_bn_kernel(const float* __restrict__ dP, const void* __restrict__ F2,
                                                               const void* __restrict__ F3, const float* __restrict__ aff2,
                                                               const float* __restrict__ aff3, const int32_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ edges, void* __restrict__ dF2,
                                                               void* __restrict__ dF3, int B, int N, int K, int L, int o16) {
    extern __shared__ int sm_i[];
    int* ps = sm_i; int* po = ps + (N + 1); int* es = po + (N + 1); int* eo = es + K;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
        ps[i] = ptr[((int64_t)0 * B + b) * (N + 1) + i];
        po[i] = ptr[((int64_t)1 * B + b) * (N + 1) + i];
    }
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        es[i] = edges[((int64_t)0 * B + b) * K + i];
        eo[i] = edges[((int64_t)1 * B + b) * K + i];
    }
    __syncthreads();
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= L) return;
    const Aff4 A2 = load_aff(aff2, L, col), A3 = load_aff(aff3, L, col);
    const float cdiv1 = 1.f + 1e-7f;
    const float* dp = dP + (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        const int64_t o = ((int64_t)b * N + n) * L + col;
        const float4 f2 = apply_aff(A2, ldsrc<SRC16>(F2, o)), f3 = apply_aff(A3, ldsrc<SRC16>(F3, o));
        auto row = [&](int e) { return *reinterpret_cast<const float4*>(dp + (int64_t)e * L); };
        gather_pair(es, ps[n], ps[n + 1], row, eo, po[n], po[n + 1], row, a, c);
        float4 r2, r3;
        r2.x = (f2.x / cdiv1 > 0.f) ? a.x * 0.5f / cdiv1 : 0.f; r2.y = (f2.y / cdiv1 > 0.f) ? a.y * 0.5f / cdiv1 : 0.f;
        r2.z = (f2.z / cdiv1 > 0.f) ? a.z * 0.5f / cdiv1 : 0.f; r2.w = (f2.w / cdiv1 > 0.f) ? a.w * 0.5f / cdiv1 : 0.f;
        r3.x = (f3.x / cdiv1 > 0.f) ? c.x * 0.5f / cdiv1 : 0.f; r3.y = (f3.y / cdiv1 > 0.f) ? c.y * 0.5f / cdiv1 : 0.f;
        r3.z = (f3.z / cdiv1 > 0.f) ? c.z * 0.5f / cdiv1 : 0.f; r3.w = (f3.w / cdiv1 > 0.f) ? c.w * 0.5f / cdiv1 : 0.f;
        const float o2[4] = {r2.x, r2.y, r2.z, r2.w}, o3[4] = {r3.x, r3.y, r3.z, r3.w};
        subgc_store_act<4>(dF2, o, o2, o16);
        subgc_store_act<4>(dF3, o, o3, o16);
    }
}

// ------------------------------------------------------------------ BatchNorm backward
// grid (C/256, slabs): part[slab][{sum dy * xhat, sum dy}][C]
template <bool SRC16>
__device__ __forceinline__ void bn_bwd_reduce2_body(const float* __restrict__ dY, const void* __restrict__ X, int M, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int rpb,
                                                             float* __restrict__ part) {
    __shared__ float4 sg[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    if (col < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + col), rs = *reinterpret_cast<const float4*>(rstd + col);
        int r = r0 + w;
        for (; r + 4 < r1; r += 8) {
            const float4 d0 = *reinterpret_cast<const float4*>(dY + (int64_t)r * C + col), d1 = *reinterpret_cast<const float4*>(dY + (int64_t)(r + 4) * C + col);
            const float4 x0 = ldsrc<SRC16>(X, (int64_t)r * C + col), x1 = ldsrc<SRC16>(X, (int64_t)(r + 4) * C + col);
            ab.x += d0.x + d1.x; ab.y += d0.y + d1.y; ab.z += d0.z + d1.z; ab.w += d0.w + d1.w;
            ag.x += d0.x * (x0.x - mu.x) * rs.x + d1.x * (x1.x - mu.x) * rs.x; ag.y += d0.y * (x0.y - mu.y) * rs.y + d1.y * (x1.y - mu.y) * rs.y;
            ag.z += d0.z * (x0.z - mu.z) * rs.z + d1.z * (x1.z - mu.z) * rs.z; ag.w += d0.w * (x0.w - mu.w) * rs.w + d1.w * (x1.w - mu.w) * rs.w;
        }
        for (; r < r1; r += 4) {
            const float4 dy = *reinterpret_cast<const float4*>(dY + (int64_t)r * C + col);
            const float4 x = ldsrc<SRC16>(X, (int64_t)r * C + col);
            ab.x += dy.x; ab.y += dy.y; ab.z += dy.z; ab.w += dy.w;
            ag.x += dy.x * (x.x - mu.x) * rs.x; ag.y += dy.y * (x.y - mu.y) * rs.y;
            ag.z += dy.z * (x.z - mu.z) * rs.z; ag.w += dy.w * (x.w - mu.w) * rs.w;
        }
    }
    sg[w][lane] = ag; sb[w][lane] = ab;
    __syncthreads();
    if (w == 0 && col < C) {
        const float4 g0 = sg[0][lane], g1 = sg[1][lane], g2 = sg[2][lane], g3 = sg[3][lane];
        const float4 b0 = sb[0][lane], b1 = sb[1][lane], b2 = sb[2][lane], b3 = sb[3][lane];
        float* o = part + (int64_t)blockIdx.y * 2 * C + col;
        *reinterpret_cast<float4*>(o) = make_float4(g0.x + g1.x + g2.x + g3.x, g0.y + g1.y + g2.y + g3.y, g0.z + g1.z + g2.z + g3.z, g0.w + g1.w + g2.w + g3.w);
        *reinterpret_cast<float4*>(o + C) = make_float4(b0.x + b1.x + b2.x + b3.x, b0.y + b1.y + b2.y + b3.y, b0.z + b1.z + b2.z + b3.z, b0.w + b1.w + b2.w + b3.w);
    }
}
// grid (C/256, row slabs): every workgroup first adds the reduce pass's slab partials of ITS columns (wave w: slabs w, w+4, ...;
// the four waves combined in a fixed order, so every workgroup gets the same totals), then streams its rows
template <bool SRC16, bool DX16>
__device__ __forceinline__ void bn_bwd_apply2_body(const float* __restrict__ dY, const void* __restrict__ X, void* __restrict__ dX,
                                                            int M, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ part, int slabs,
                                                            int rpb, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    __shared__ float4 sg[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = (blockIdx.x * 64 + lane) * 4;
    float4 tg = make_float4(0.f, 0.f, 0.f, 0.f), tb = tg;
    if (col < C)
        for (int k0 = w; k0 < slabs; k0 += 4 * 8) {                        // eight slabs requested before any is added (else: a chain of L2 latencies)
            float4 g[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + 4 * u, slabs - 1);
                g[u] = *reinterpret_cast<const float4*>(part + (int64_t)k * 2 * C + col);
                b[u] = *reinterpret_cast<const float4*>(part + ((int64_t)k * 2 + 1) * C + col);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 4 * u < slabs) {
                    tg.x += g[u].x; tg.y += g[u].y; tg.z += g[u].z; tg.w += g[u].w;
                    tb.x += b[u].x; tb.y += b[u].y; tb.z += b[u].z; tb.w += b[u].w;
                }
        }
    sg[w][lane] = tg; sb[w][lane] = tb;
    __syncthreads();
    if (col >= C) return;
    {
        const float4 g0 = sg[0][lane], g1 = sg[1][lane], g2 = sg[2][lane], g3 = sg[3][lane];
        const float4 b0 = sb[0][lane], b1 = sb[1][lane], b2 = sb[2][lane], b3 = sb[3][lane];
        tg = make_float4(g0.x + g1.x + g2.x + g3.x, g0.y + g1.y + g2.y + g3.y, g0.z + g1.z + g2.z + g3.z, g0.w + g1.w + g2.w + g3.w);
        tb = make_float4(b0.x + b1.x + b2.x + b3.x, b0.y + b1.y + b2.y + b3.y, b0.z + b1.z + b2.z + b3.z, b0.w + b1.w + b2.w + b3.w);
    }
    if (blockIdx.y == 0 && w == 0) {
        float4 og = tg, ob = tb;
        if (accumulate) {
            const float4 pg = *reinterpret_cast<const float4*>(dgamma + col), pb = *reinterpret_cast<const float4*>(dbeta + col);
            og.x += pg.x; og.y += pg.y; og.z += pg.z; og.w += pg.w;
            ob.x += pb.x; ob.y += pb.y; ob.z += pb.z; ob.w += pb.w;
        }
        *reinterpret_cast<float4*>(dgamma + col) = og;
        *reinterpret_cast<float4*>(dbeta + col) = ob;
    }
    const float invM = 1.f / (float)M;
    const float4 mu = *reinterpret_cast<const float4*>(mean + col), rs = *reinterpret_cast<const float4*>(rstd + col);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + col);
    const float kx = ga.x * rs.x, ky = ga.y * rs.y, kz = ga.z * rs.z, kw = ga.w * rs.w;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    auto one = [&](int64_t o, const float4& dy, const float4& x) {
        const float out[4] = {kx * (dy.x - tb.x * invM - (x.x - mu.x) * rs.x * tg.x * invM), ky * (dy.y - tb.y * invM - (x.y - mu.y) * rs.y * tg.y * invM),
                              kz * (dy.z - tb.z * invM - (x.z - mu.z) * rs.z * tg.z * invM), kw * (dy.w - tb.w * invM - (x.w - mu.w) * rs.w * tg.w * invM)};
        subgc_store_act<4>(dX, o, out, DX16 ? 1 : 0);
    };
    int r = r0 + w;
    for (; r + 12 < r1; r += 16) {                                         // four rows requested before the first is used
        float4 dy[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t o = (int64_t)(r + 4 * u) * C + col;
            dy[u] = *reinterpret_cast<const float4*>(dY + o);
            x[u] = ldsrc<SRC16>(X, o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one((int64_t)(r + 4 * u) * C + col, dy[u], x[u]);
    }
    for (; r < r1; r += 4) {
        const int64_t o = (int64_t)r * C + col;
        one(o, *reinterpret_cast<const float4*>(dY + o), ldsrc<SRC16>(X, o));
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
inline int stat_rows_per_block(int M, int C) {
    const int col_groups = (C + 255) / 256;
    const int slabs = std::max(1, 512 / col_groups);
    return std::max(16, (M + slabs - 1) / slabs);
}
inline int raise_lds(const void* fn, size_t bytes, const char* what) { return subgc::raise_lds_cached(fn, bytes, what); }

template <bool SRC16>
__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const float* __restrict__ dY, const void* __restrict__ X, int M, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, int rpb,
                                                             float* __restrict__ part) {
    bn_bwd_reduce2_body<SRC16>(dY, X, M, C, mean, rstd, rpb, part);
}
template <bool SRC16, bool DX16>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const float* __restrict__ dY, const void* __restrict__ X, void* __restrict__ dX,
                                                            int M, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ part, int slabs,
                                                            int rpb, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    bn_bwd_apply2_body<SRC16, DX16>(dY, X, dX, M, C, mean, rstd, gamma, part, slabs, rpb, dgamma, dbeta, accumulate);
}
// both units of a GCN pair (blockIdx.z = unit; partials at part + z * slabs * 2 C)
struct BnPairBwd { const float* dY[2]; const void* X[2]; void* dX[2]; const float* mean[2]; const float* rstd[2]; const float* gamma[2]; float* dgamma[2]; float* dbeta[2]; };
template <bool SRC16>
__global__ __launch_bounds__(256) void bn_bwd_reduce2_pair_kernel(BnPairBwd a, int M, int C, int rpb, float* __restrict__ part, int slabs) {
    const int z = blockIdx.z;
    bn_bwd_reduce2_body<SRC16>(a.dY[z], a.X[z], M, C, a.mean[z], a.rstd[z], rpb, part + (size_t)z * slabs * 2 * C);
}
template <bool SRC16, bool DX16>
__global__ __launch_bounds__(256) void bn_bwd_apply2_pair_kernel(BnPairBwd a, int M, int C, const float* __restrict__ part, int slabs, int rpb, int accumulate) {
    const int z = blockIdx.z;
    bn_bwd_apply2_body<SRC16, DX16>(a.dY[z], a.X[z], a.dX[z], M, C, a.mean[z], a.rstd[z], a.gamma[z], part + (size_t)z * slabs * 2 * C, slabs, rpb, a.dgamma[z],
                                    a.dbeta[z], accumulate);
}

}  // namespace

SUBGC_API int subgc_bn_stats_workspace_bytes(int M, int C, size_t* bytes) {
    SUBGC_REQUIRE(M > 0 && C > 0 && bytes, "bn_stats_workspace_bytes: bad arguments");
    const int rpb = stat_rows_per_block(M, C);
    *bytes = (size_t)((M + rpb - 1) / rpb) * 3 * C * sizeof(float);
    return SUBGC_OK;
}

SUBGC_API int subgc_bn_stats(const void* X, int x_bf16, int M, int C, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float* aff, float* rstd, int training, float momentum, float eps, void* workspace,
                             size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_stats: M > 0 and C a positive multiple of 4 (got M=%d C=%d)", M, C);
    SUBGC_REQUIRE(gamma && beta && aff, "bn_stats: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!training) {
        SUBGC_REQUIRE(running_mean && running_var, "bn_stats(eval): running statistics required");
        hipLaunchKernelGGL(bn_eval_aff_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, gamma, beta, (const float*)running_mean,
                           (const float*)running_var, aff, rstd, eps);
        return subgc::check_launch("subgc_bn_stats");
    }
    SUBGC_REQUIRE(X && rstd, "bn_stats(train): null pointer");
    SUBGC_REQUIRE(x_bf16 ? al8(X) : al16(X), "bn_stats: X must be 16-byte (fp32) / 8-byte (bf16) aligned");
    const int rpb = stat_rows_per_block(M, C), slabs = (M + rpb - 1) / rpb;
    SUBGC_REQUIRE(workspace && al16(workspace) && ws_bytes >= (size_t)slabs * 3 * C * sizeof(float),
                  "bn_stats: workspace of %zu bytes required", (size_t)slabs * 3 * C * sizeof(float));
    float* part = static_cast<float*>(workspace);
    const dim3 g((C + 255) / 256, slabs);
    if (x_bf16) hipLaunchKernelGGL((bn_stats_kernel<true>), g, dim3(256), 0, s, X, M, C, rpb, part);
    else hipLaunchKernelGGL((bn_stats_kernel<false>), g, dim3(256), 0, s, X, M, C, rpb, part);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 63) / 64), dim3(FIN_WAVES * 64), 0, s, (const float*)part, slabs, rpb, M, C, gamma, beta, running_mean,
                       running_var, aff, rstd, momentum, eps);
    return subgc::check_launch("subgc_bn_stats");
}

SUBGC_API int subgc_bn_bwd_fused(const float* dY, const void* X, int x_bf16, int M, int C, const float* gamma, const float* mean,
                                 const float* rstd, void* dX, int dx_bf16, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                 size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_fused: M > 0 and C a positive multiple of 4");
    SUBGC_REQUIRE(dY && X && gamma && mean && rstd && dX && dgamma && dbeta, "bn_bwd_fused: null pointer");
    SUBGC_REQUIRE(al16(dY) && (x_bf16 ? al8(X) : al16(X)) && (dx_bf16 ? al8(dX) : al16(dX)) && al16(dgamma) && al16(dbeta) && al16(mean) && al16(rstd) &&
                      al16(gamma), "bn_bwd_fused: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = stat_rows_per_block(M, C), slabs = (M + rpb - 1) / rpb;
    SUBGC_REQUIRE(workspace && al16(workspace) && ws_bytes >= (size_t)slabs * 2 * C * sizeof(float), "bn_bwd_fused: workspace of %zu bytes required",
                  (size_t)slabs * 2 * C * sizeof(float));
    float* part = static_cast<float*>(workspace);
    const dim3 g((C + 255) / 256, slabs);
    if (x_bf16) hipLaunchKernelGGL((bn_bwd_reduce2_kernel<true>), g, dim3(256), 0, s, dY, X, M, C, mean, rstd, rpb, part);
    else hipLaunchKernelGGL((bn_bwd_reduce2_kernel<false>), g, dim3(256), 0, s, dY, X, M, C, mean, rstd, rpb, part);
    const int col_groups = (C + 255) / 256;
    const int rpb2 = std::max(8, (int)(((int64_t)M * col_groups + 1023) / 1024));
    const dim3 g2(col_groups, (M + rpb2 - 1) / rpb2);
#define SUBGC_BN_APPLY(S16, D16) \
    hipLaunchKernelGGL((bn_bwd_apply2_kernel<S16, D16>), g2, dim3(256), 0, s, dY, X, dX, M, C, mean, rstd, gamma, (const float*)part, slabs, rpb2, dgamma, dbeta, accumulate)
    if (x_bf16) { if (dx_bf16) SUBGC_BN_APPLY(true, true); else SUBGC_BN_APPLY(true, false); }
    else { if (dx_bf16) SUBGC_BN_APPLY(false, true); else SUBGC_BN_APPLY(false, false); }
#undef SUBGC_BN_APPLY
    return subgc::check_launch("subgc_bn_bwd_fused");
}

// The two units of a GCN pair (same M, C, storage type) through subgc_bn_stats in two launches instead of four
SUBGC_API int subgc_bn_stats_pair(const void* X0, const void* X1, int x_bf16, int M, int C, const float* gamma0, const float* gamma1, const float* beta0,
                                  const float* beta1, float* rmean0, float* rmean1, float* rvar0, float* rvar1, float* aff0, float* aff1, float* rstd0,
                                  float* rstd1, float momentum, float eps, void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_stats_pair: M > 0 and C a positive multiple of 4 (got M=%d C=%d)", M, C);
    SUBGC_REQUIRE(X0 && X1 && gamma0 && gamma1 && beta0 && beta1 && aff0 && aff1 && rstd0 && rstd1, "bn_stats_pair: null pointer");
    SUBGC_REQUIRE(x_bf16 ? (al8(X0) && al8(X1)) : (al16(X0) && al16(X1)), "bn_stats_pair: X must be 16-byte (fp32) / 8-byte (bf16) aligned");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = stat_rows_per_block(M, C), slabs = (M + rpb - 1) / rpb;
    SUBGC_REQUIRE(workspace && al16(workspace) && ws_bytes >= (size_t)2 * slabs * 3 * C * sizeof(float), "bn_stats_pair: workspace of %zu bytes required",
                  (size_t)2 * slabs * 3 * C * sizeof(float));
    float* part = static_cast<float*>(workspace);
    BnPair a{{X0, X1}, {gamma0, gamma1}, {beta0, beta1}, {rmean0, rmean1}, {rvar0, rvar1}, {aff0, aff1}, {rstd0, rstd1}};
    const dim3 g((C + 255) / 256, slabs, 2);
    if (x_bf16) hipLaunchKernelGGL((bn_stats_pair_kernel<true>), g, dim3(256), 0, s, a, M, C, rpb, part, slabs);
    else hipLaunchKernelGGL((bn_stats_pair_kernel<false>), g, dim3(256), 0, s, a, M, C, rpb, part, slabs);
    hipLaunchKernelGGL(bn_stats_finish_pair_kernel, dim3((C + 63) / 64, 2), dim3(FIN_WAVES * 64), 0, s, (const float*)part, slabs, rpb, M, C, a, momentum, eps);
    return subgc::check_launch("subgc_bn_stats_pair");
}

// ... and through subgc_bn_bwd_fused in two launches instead of four
SUBGC_API int subgc_bn_bwd_fused_pair(const float* dY0, const float* dY1, const void* X0, const void* X1, int x_bf16, int M, int C, const float* gamma0,
                                      const float* gamma1, const float* mean0, const float* mean1, const float* rstd0, const float* rstd1, void* dX0,
                                      void* dX1, int dx_bf16, float* dgamma0, float* dgamma1, float* dbeta0, float* dbeta1, int accumulate,
                                      void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd_fused_pair: M > 0 and C a positive multiple of 4");
    SUBGC_REQUIRE(dY0 && dY1 && X0 && X1 && gamma0 && gamma1 && mean0 && mean1 && rstd0 && rstd1 && dX0 && dX1 && dgamma0 && dgamma1 && dbeta0 && dbeta1,
                  "bn_bwd_fused_pair: null pointer");
    SUBGC_REQUIRE(al16(dY0) && al16(dY1) && (x_bf16 ? (al8(X0) && al8(X1)) : (al16(X0) && al16(X1))) && (dx_bf16 ? (al8(dX0) && al8(dX1)) : (al16(dX0) && al16(dX1))) &&
                      al16(dgamma0) && al16(dgamma1) && al16(dbeta0) && al16(dbeta1) && al16(mean0) && al16(mean1) && al16(rstd0) && al16(rstd1) && al16(gamma0) &&
                      al16(gamma1), "bn_bwd_fused_pair: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = stat_rows_per_block(M, C), slabs = (M + rpb - 1) / rpb;
    SUBGC_REQUIRE(workspace && al16(workspace) && ws_bytes >= (size_t)2 * slabs * 2 * C * sizeof(float), "bn_bwd_fused_pair: workspace of %zu bytes required",
                  (size_t)2 * slabs * 2 * C * sizeof(float));
    float* part = static_cast<float*>(workspace);
    BnPairBwd a{{dY0, dY1}, {X0, X1}, {dX0, dX1}, {mean0, mean1}, {rstd0, rstd1}, {gamma0, gamma1}, {dgamma0, dgamma1}, {dbeta0, dbeta1}};
    const dim3 g((C + 255) / 256, slabs, 2);
    if (x_bf16) hipLaunchKernelGGL((bn_bwd_reduce2_pair_kernel<true>), g, dim3(256), 0, s, a, M, C, rpb, part, slabs);
    else hipLaunchKernelGGL((bn_bwd_reduce2_pair_kernel<false>), g, dim3(256), 0, s, a, M, C, rpb, part, slabs);
    const int col_groups = (C + 255) / 256;
    const int rpb2 = std::max(8, (int)(((int64_t)M * col_groups + 1023) / 1024));
    const dim3 g2(col_groups, (M + rpb2 - 1) / rpb2, 2);
#define SUBGC_BN_APPLY(S16, D16) hipLaunchKernelGGL((bn_bwd_apply2_pair_kernel<S16, D16>), g2, dim3(256), 0, s, a, M, C, (const float*)part, slabs, rpb2, accumulate)
    if (x_bf16) { if (dx_bf16) SUBGC_BN_APPLY(true, true); else SUBGC_BN_APPLY(true, false); }
    else { if (dx_bf16) SUBGC_BN_APPLY(false, true); else SUBGC_BN_APPLY(false, false); }
#undef SUBGC_BN_APPLY
    return subgc::check_launch("subgc_bn_bwd_fused_pair");
}

SUBGC_API int subgc_gcn_nodes_fwd_bn(const void* F0, const void* F1, int f_bf16, const float* aff0, const float* aff1, const int32_t* ptr,
                                     const int32_t* edges, const float* skip, float* Xout, uint16_t* Xout16, uint8_t* act, int B, int N, int K,
                                     int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0 && L % 4 == 0, "gcn_nodes_fwd_bn: bad sizes (L must be a multiple of 4)");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(F0 && F1 && ptr && edges && Xout, "gcn_nodes_fwd_bn: null pointer");
    SUBGC_REQUIRE((f_bf16 ? (al8(F0) && al8(F1)) : (al16(F0) && al16(F1))) && al16(skip) && al16(Xout) && al8(Xout16) && al16(aff0) && al16(aff1) &&
                      (reinterpret_cast<uintptr_t>(act) & 3) == 0, "gcn_nodes_fwd_bn: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, (f_bf16 ? 2.0 : 4.0) * B * L * 2.0 * K + 4.0 * B * L * (skip ? 2.0 : 1.0) * N);
    const size_t lds = sizeof(int) * (2 * (N + 1) + 2 * K);
    const dim3 g((L / 4 + 255) / 256, B, subgc::gcn_zsplit((L / 4 + 255) / 256, B, N));
    if (f_bf16) hipLaunchKernelGGL((gcn_nodes_fwd_bn_kernel<true>), g, dim3(256), lds, s, F0, F1, aff0, aff1, ptr, edges, skip, Xout, Xout16, act, B, N, K, L);
    else hipLaunchKernelGGL((gcn_nodes_fwd_bn_kernel<false>), g, dim3(256), lds, s, F0, F1, aff0, aff1, ptr, edges, skip, Xout, Xout16, act, B, N, K, L);
    return subgc::check_launch("subgc_gcn_nodes_fwd_bn");
}

SUBGC_API int subgc_gcn_edges_fwd_bn(const void* F2, const void* F3, int f_bf16, const float* aff2, const float* aff3, const int64_t* rel_ind,
                                     const float* skip, float* Pout, uint16_t* Pout16, int B, int N, int K, int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0 && L % 4 == 0, "gcn_edges_fwd_bn: bad sizes (L must be a multiple of 4)");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(F2 && F3 && rel_ind && Pout, "gcn_edges_fwd_bn: null pointer");
    SUBGC_DEBUG_RANGE(rel_ind, 8, (int64_t)B * K, 2, 2, 0, N - 1, -1, "gcn_edges_fwd_bn: rel_ind", stream);
    SUBGC_REQUIRE((f_bf16 ? (al8(F2) && al8(F3)) : (al16(F2) && al16(F3))) && al16(aff2) && al16(aff3), "gcn_edges_fwd_bn: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * 2 * (size_t)N * TC + sizeof(int) * 2 * K;
    int rc = f_bf16 ? raise_lds((const void*)gcn_edges_fwd_bn_kernel<true>, lds, "gcn_edges_fwd_bn")
                    : raise_lds((const void*)gcn_edges_fwd_bn_kernel<false>, lds, "gcn_edges_fwd_bn");
    if (rc) return rc;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, (f_bf16 ? 2.0 : 4.0) * B * L * 2.0 * N + 4.0 * B * L * (skip ? 2.0 : 1.0) * K);
    const dim3 g((L + TC - 1) / TC, B);
    const int vec = al16(Pout) && al16(skip) && al8(Pout16);
    if (f_bf16) hipLaunchKernelGGL((gcn_edges_fwd_bn_kernel<true>), g, dim3(256), lds, s, F2, F3, aff2, aff3, rel_ind, skip, Pout, Pout16, B, N, K, L, vec);
    else hipLaunchKernelGGL((gcn_edges_fwd_bn_kernel<false>), g, dim3(256), lds, s, F2, F3, aff2, aff3, rel_ind, skip, Pout, Pout16, B, N, K, L, vec);
    return subgc::check_launch("subgc_gcn_edges_fwd_bn");
}

SUBGC_API int subgc_gcn_edges_bwd_bn(const float* dP, const void* F2, const void* F3, int f_bf16, const float* aff2, const float* aff3,
                                     const int32_t* ptr, const int32_t* edges, void* dF2, void* dF3, int out_bf16, int B, int N, int K, int L,
                                     void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0 && L % 4 == 0, "gcn_edges_bwd_bn: bad sizes (L must be a multiple of 4)");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(dP && F2 && F3 && ptr && edges && dF2 && dF3, "gcn_edges_bwd_bn: null pointer");
    SUBGC_REQUIRE(al16(dP) && (f_bf16 ? (al8(F2) && al8(F3)) : (al16(F2) && al16(F3))) && (out_bf16 ? (al8(dF2) && al8(dF3)) : (al16(dF2) && al16(dF3))) &&
                      al16(aff2) && al16(aff3), "gcn_edges_bwd_bn: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, 4.0 * B * L * (2.0 * K + 2.0 * N) + (f_bf16 ? 2.0 : 4.0) * B * L * 2.0 * N);
    const size_t lds = sizeof(int) * (2 * (N + 1) + 2 * K);
    const dim3 g((L / 4 + 255) / 256, B, subgc::gcn_zsplit((L / 4 + 255) / 256, B, N));
    if (f_bf16) hipLaunchKernelGGL((gcn_edges_bwd_bn_kernel<true>), g, dim3(256), lds, s, dP, F2, F3, aff2, aff3, ptr, edges, dF2, dF3, B, N, K, L, out_bf16);
    else hipLaunchKernelGGL((gcn_edges_bwd_bn_kernel<false>), g, dim3(256), lds, s, dP, F2, F3, aff2, aff3, ptr, edges, dF2, dF3, B, N, K, L, out_bf16);
    return subgc::check_launch("subgc_gcn_edges_bwd_bn");
}
