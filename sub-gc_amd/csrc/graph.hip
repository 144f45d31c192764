// Scene-graph kernels: class arg-max, CSR build, GCN neighbour aggregation (fwd + bwd), BatchNorm.
// All are HBM-bound streaming kernels: threads run along the feature dimension L (coalesced
// 256-float rows), indices live in LDS, node tiles are LDS-staged where rows are re-used.
//
// Reference op sites: AttModel.py:376,383,385 (argmax); gcn_backbone.py:55-67 (make_map, replaced
// by CSR); graph_conv_unit.py:28-36 (bmm + normalise + ReLU, BatchNorm); graph_conv.py:24-33
// (average of the two roles); gcn_backbone.py:43-47 (residual).
#include "common.h"
#include "bf16_util.h"

#include <algorithm>

namespace {

// ------------------------------------------------------------------ row arg-max (first max)
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ X, int64_t ldx, int rows,
                                                         int cols, int skip, int64_t* __restrict__ idx,
                                                         float* __restrict__ val, int32_t* __restrict__ idx32) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (int64_t)row * ldx;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = skip + lane; c < cols; c += 64) {
        const float v = x[c];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }   // strict >: first max inside a lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        if (idx) idx[row] = bi;
        if (idx32) idx32[row] = bi;
        if (val) val[row] = best;
    }
}

// ------------------------------------------------------------------ CSR by subject / by object
// grid (B, 2 roles); thread n owns node n: counts its relations, then lists them in ascending k.
__global__ __launch_bounds__(256) void csr_build_kernel(const int64_t* __restrict__ rel_ind, int B, int K, int N,
                                                        int32_t* __restrict__ ptr, int32_t* __restrict__ edges) {
    extern __shared__ int sm_i[];
    int* node_of = sm_i;          // [K]
    int* start = sm_i + K;        // [N+1]
    const int b = blockIdx.x, role = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int64_t v = rel_ind[((int64_t)b * K + k) * 2 + role];
        node_of[k] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int c = 0;
        for (int k = 0; k < K; ++k) c += node_of[k] == n;
        start[n + 1] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        start[0] = 0;
        for (int n = 0; n < N; ++n) start[n + 1] += start[n];
    }
    __syncthreads();
    int32_t* p = ptr + ((int64_t)role * B + b) * (N + 1);
    int32_t* e = edges + ((int64_t)role * B + b) * K;
    for (int n = threadIdx.x; n <= N; n += blockDim.x) p[n] = start[n];
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int pos = start[n];
        for (int k = 0; k < K; ++k)
            if (node_of[k] == n) e[pos++] = k;
    }
}

// ------------------------------------------------------------------ nodes <- relations
// grid (L/256, B); thread = one feature column; CSR lists of the image in LDS.
__global__ __launch_bounds__(256) void gcn_nodes_fwd_kernel(const float* __restrict__ F0, const float* __restrict__ F1,
                                                            const int32_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ edges,
                                                            const float* __restrict__ skip, float* __restrict__ Xout,
                                                            uint8_t* __restrict__ act, int B, int N, int K, int L) {
    extern __shared__ int sm_i[];
    int* ps = sm_i;                 // [N+1] subject ptr
    int* po = ps + (N + 1);         // [N+1] object ptr
    int* es = po + (N + 1);         // [K]
    int* eo = es + K;               // [K]
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
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= L) return;
    const float* f0 = F0 + (int64_t)b * K * L + col;
    const float* f1 = F1 + (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {           // node chunks over gridDim.z: shorter dependent chains, 4x the loads in flight
        float a = 0.f, c = 0.f;
        const int s0 = ps[n], s1 = ps[n + 1], o0 = po[n], o1 = po[n + 1];
        for (int j = s0; j < s1; ++j) a += f0[(int64_t)es[j] * L];
        for (int j = o0; j < o1; ++j) c += f1[(int64_t)eo[j] * L];
        a = a / ((float)(s1 - s0) + 1e-7f);
        c = c / ((float)(o1 - o0) + 1e-7f);
        const uint8_t bits = (a > 0.f ? 1 : 0) | (c > 0.f ? 2 : 0);
        float v = (fmaxf(a, 0.f) + fmaxf(c, 0.f)) / 2.f;
        const int64_t o = ((int64_t)b * N + n) * L + col;
        if (skip) v += skip[o];
        Xout[o] = v;
        if (act) act[o] = bits;
    }
}

// float4 form of the same kernel (L % 4 == 0, 16-byte aligned rows): a thread owns four adjacent columns, so a wave reads 1 KB
// of a relation row per instruction instead of 256 B (the scalar form ran at 2.3 TB/s on Full-GC's 65 x 1024 relation rows);
// every element is summed in the same order as before (bit-identical results)
__global__ __launch_bounds__(256) void gcn_nodes_fwd_vec_kernel(const float* __restrict__ F0, const float* __restrict__ F1,
                                                                const int32_t* __restrict__ ptr, const int32_t* __restrict__ edges,
                                                                const float* __restrict__ skip, float* __restrict__ Xout,
                                                                uint8_t* __restrict__ act, int B, int N, int K, int L) {
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
    const float* f0 = F0 + (int64_t)b * K * L + col;
    const float* f1 = F1 + (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        const int s0 = ps[n], s1 = ps[n + 1], o0 = po[n], o1 = po[n + 1];
        gather_pair(es, s0, s1, [&](int e) { return *reinterpret_cast<const float4*>(f0 + (int64_t)e * L); },
                    eo, o0, o1, [&](int e) { return *reinterpret_cast<const float4*>(f1 + (int64_t)e * L); }, a, c);
        const float da = (float)(s1 - s0) + 1e-7f, dc = (float)(o1 - o0) + 1e-7f;
        float av[4] = {a.x / da, a.y / da, a.z / da, a.w / da}, cv[4] = {c.x / dc, c.y / dc, c.z / dc, c.w / dc};
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
        if (act) *reinterpret_cast<uint32_t*>(act + o) = bits;
    }
}

// dF0[b,k,:] = 1/2 [bit0](b,s_k,:) dX[b,s_k,:] / (cnt_s[s_k] + 1e-7);  dF1 likewise with o_k / bit1
__global__ __launch_bounds__(256) void gcn_nodes_bwd_kernel(const float* __restrict__ dX, const uint8_t* __restrict__ act,
                                                            const int64_t* __restrict__ rel_ind,
                                                            const int32_t* __restrict__ ptr, float* __restrict__ dF0,
                                                            float* __restrict__ dF1, int B, int N, int K, int L) {
    extern __shared__ int sm_i[];
    int* ns = sm_i;            // [K] subject node of k
    int* no = ns + K;          // [K] object node of k
    float* ds = reinterpret_cast<float*>(no + K);   // [N] cnt_s + 1e-7 denominators
    float* dn_o = ds + N;
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int64_t s = rel_ind[((int64_t)b * K + k) * 2 + 0], o = rel_ind[((int64_t)b * K + k) * 2 + 1];
        ns[k] = (int)(s < 0 ? 0 : (s >= N ? N - 1 : s));
        no[k] = (int)(o < 0 ? 0 : (o >= N ? N - 1 : o));
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int32_t* p0 = ptr + ((int64_t)0 * B + b) * (N + 1);
        const int32_t* p1 = ptr + ((int64_t)1 * B + b) * (N + 1);
        ds[n] = (float)(p0[n + 1] - p0[n]) + 1e-7f;
        dn_o[n] = (float)(p1[n + 1] - p1[n]) + 1e-7f;
    }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= L) return;
    for (int k = blockIdx.z; k < K; k += gridDim.z) {
        const int s = ns[k], o = no[k];
        const int64_t is = ((int64_t)b * N + s) * L + col, io = ((int64_t)b * N + o) * L + col;
        const float gs = (act[is] & 1) ? dX[is] * 0.5f / ds[s] : 0.f;
        const float go = (act[io] & 2) ? dX[io] * 0.5f / dn_o[o] : 0.f;
        const int64_t ok = ((int64_t)b * K + k) * L + col;
        dF0[ok] = gs;
        dF1[ok] = go;
    }
}

// o16: dF0 / dF1 are bf16 (the unit outputs they are gradients of were bf16 GEMM results: compute_dtype = bf16 without BatchNorm)
__global__ __launch_bounds__(256) void gcn_nodes_bwd_vec_kernel(const float* __restrict__ dX, const uint8_t* __restrict__ act,
                                                                const int64_t* __restrict__ rel_ind, const int32_t* __restrict__ ptr,
                                                                void* __restrict__ dF0, void* __restrict__ dF1, int B, int N, int K, int L, int o16) {
    extern __shared__ int sm_i[];
    int* ns = sm_i; int* no = ns + K;
    float* ds = reinterpret_cast<float*>(no + K); float* dn_o = ds + N;
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int64_t s = rel_ind[((int64_t)b * K + k) * 2 + 0], o = rel_ind[((int64_t)b * K + k) * 2 + 1];
        ns[k] = (int)(s < 0 ? 0 : (s >= N ? N - 1 : s));
        no[k] = (int)(o < 0 ? 0 : (o >= N ? N - 1 : o));
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int32_t* p0 = ptr + ((int64_t)0 * B + b) * (N + 1);
        const int32_t* p1 = ptr + ((int64_t)1 * B + b) * (N + 1);
        ds[n] = (float)(p0[n + 1] - p0[n]) + 1e-7f;
        dn_o[n] = (float)(p1[n + 1] - p1[n]) + 1e-7f;
    }
    __syncthreads();
    const int col = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (col >= L) return;
    const int kz = gridDim.z;
    for (int k0 = blockIdx.z; k0 < K; k0 += 4 * kz) {                      // four relations' node rows requested before the first is used
        int sn[4], on[4];
        uint32_t as[4], ao[4];
        float4 xs[4], xo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * kz;
            if (k < K) {
                sn[u] = ns[k]; on[u] = no[k];
                const int64_t is = ((int64_t)b * N + sn[u]) * L + col, io = ((int64_t)b * N + on[u]) * L + col;
                as[u] = *reinterpret_cast<const uint32_t*>(act + is); ao[u] = *reinterpret_cast<const uint32_t*>(act + io);
                xs[u] = *reinterpret_cast<const float4*>(dX + is); xo[u] = *reinterpret_cast<const float4*>(dX + io);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * kz;
            if (k < K) {
                const float xsv[4] = {xs[u].x, xs[u].y, xs[u].z, xs[u].w}, xov[4] = {xo[u].x, xo[u].y, xo[u].z, xo[u].w};
                float gs[4], go[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gs[e] = ((as[u] >> (8 * e)) & 1u) ? xsv[e] * 0.5f / ds[sn[u]] : 0.f;
                    go[e] = ((ao[u] >> (8 * e)) & 2u) ? xov[e] * 0.5f / dn_o[on[u]] : 0.f;
                }
                const int64_t ok = ((int64_t)b * K + k) * L + col;
                subgc_store_act<4>(dF0, ok, gs, o16);
                subgc_store_act<4>(dF1, ok, go, o16);
            }
        }
    }
}

// ------------------------------------------------------------------ relations <- nodes
// grid (L/TC, B), TC = 128 columns; the image's F2/F3 node tiles [N x TC] are staged in LDS
// (each node row is consumed by ~K/N relations per role), then every relation gathers from LDS.
constexpr int TC = 128;
template <bool VEC>
__global__ __launch_bounds__(256) void gcn_edges_fwd_kernel(const float* __restrict__ F2, const float* __restrict__ F3,
                                                            const int64_t* __restrict__ rel_ind,
                                                            const float* __restrict__ skip, float* __restrict__ Pout,
                                                            int B, int N, int K, int L) {
    extern __shared__ __attribute__((aligned(16))) float sm_f[];
    float* t2 = sm_f;                    // [N][TC]
    float* t3 = sm_f + (size_t)N * TC;   // [N][TC]
    int* ns = reinterpret_cast<int*>(t3 + (size_t)N * TC);   // [K]
    int* no = ns + K;
    const int b = blockIdx.y, c0 = blockIdx.x * TC;
    const float cdiv1 = 1.f + 1e-7f;     // fp32(1 + 1e-7) = 1.00000012: rowsum of a 0/1 incidence row + 1e-7
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
                const int64_t g = ((int64_t)b * N + n) * L + c0 + c4;
                if (c0 + c4 + 3 < L) {
                    a[u] = *reinterpret_cast<const float4*>(F2 + g);
                    c[u] = *reinterpret_cast<const float4*>(F3 + g);
                } else {
                    float* pa = &a[u].x; float* pc = &c[u].x;
                    for (int j = 0; j < 4; ++j)
                        if (c0 + c4 + j < L) { pa[j] = F2[g + j]; pc[j] = F3[g + j]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < items) {
                const int n = i / (TC / 4), c4 = (i % (TC / 4)) * 4;
                // relu(x / c) is what every consumer needs: do it once per node element
                float4 x = a[u], y = c[u];
                x.x = fmaxf(x.x / cdiv1, 0.f); x.y = fmaxf(x.y / cdiv1, 0.f); x.z = fmaxf(x.z / cdiv1, 0.f); x.w = fmaxf(x.w / cdiv1, 0.f);
                y.x = fmaxf(y.x / cdiv1, 0.f); y.y = fmaxf(y.y / cdiv1, 0.f); y.z = fmaxf(y.z / cdiv1, 0.f); y.w = fmaxf(y.w / cdiv1, 0.f);
                *reinterpret_cast<float4*>(t2 + n * TC + c4) = x;
                *reinterpret_cast<float4*>(t3 + n * TC + c4) = y;
            }
        }
    }
    __syncthreads();
    if (VEC) {                                                             // 8 relation streams x 32 lanes x 4 columns: 16-byte LDS reads and stores
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
                    *reinterpret_cast<float4*>(Pout + ((int64_t)b * K + k) * L + c0 + cl4) = r;
                }
            }
        }
        return;
    }
    const int cl = threadIdx.x & (TC - 1), half = threadIdx.x >> 7;   // 2 relation streams x 128 columns
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
            if (k < K) Pout[((int64_t)b * K + k) * L + col] = skip ? v[u] + sk[u] : v[u];
        }
    }
}

// dF2[b,n,:] = 1/2 [F2>0] / c * sum_{k in CSR_s(n)} dP[b,k,:]   (and F3 / CSR_o)
__global__ __launch_bounds__(256) void gcn_edges_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ F2,
                                                            const float* __restrict__ F3, const int32_t* __restrict__ ptr,
                                                            const int32_t* __restrict__ edges, float* __restrict__ dF2,
                                                            float* __restrict__ dF3, int B, int N, int K, int L) {
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
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= L) return;
    const float cdiv1 = 1.f + 1e-7f;
    const float* dp = dP + (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        float a = 0.f, c = 0.f;
        for (int j = ps[n]; j < ps[n + 1]; ++j) a += dp[(int64_t)es[j] * L];
        for (int j = po[n]; j < po[n + 1]; ++j) c += dp[(int64_t)eo[j] * L];
        const int64_t o = ((int64_t)b * N + n) * L + col;
        dF2[o] = (F2[o] / cdiv1 > 0.f) ? a * 0.5f / cdiv1 : 0.f;
        dF3[o] = (F3[o] / cdiv1 > 0.f) ? c * 0.5f / cdiv1 : 0.f;
    }
}

__global__ __launch_bounds__(256) void gcn_edges_bwd_vec_kernel(const float* __restrict__ dP, const float* __restrict__ F2,
                                                                const float* __restrict__ F3, const int32_t* __restrict__ ptr,
                                                                const int32_t* __restrict__ edges, float* __restrict__ dF2,
                                                                float* __restrict__ dF3, int B, int N, int K, int L) {
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
    const float cdiv1 = 1.f + 1e-7f;
    const float* dp = dP + (int64_t)b * K * L + col;
    for (int n = blockIdx.z; n < N; n += gridDim.z) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        const int64_t o = ((int64_t)b * N + n) * L + col;
        const float4 f2 = *reinterpret_cast<const float4*>(F2 + o), f3 = *reinterpret_cast<const float4*>(F3 + o);
        auto row = [&](int e) { return *reinterpret_cast<const float4*>(dp + (int64_t)e * L); };
        gather_pair(es, ps[n], ps[n + 1], row, eo, po[n], po[n + 1], row, a, c);
        float4 r2, r3;
        r2.x = (f2.x / cdiv1 > 0.f) ? a.x * 0.5f / cdiv1 : 0.f; r2.y = (f2.y / cdiv1 > 0.f) ? a.y * 0.5f / cdiv1 : 0.f;
        r2.z = (f2.z / cdiv1 > 0.f) ? a.z * 0.5f / cdiv1 : 0.f; r2.w = (f2.w / cdiv1 > 0.f) ? a.w * 0.5f / cdiv1 : 0.f;
        r3.x = (f3.x / cdiv1 > 0.f) ? c.x * 0.5f / cdiv1 : 0.f; r3.y = (f3.y / cdiv1 > 0.f) ? c.y * 0.5f / cdiv1 : 0.f;
        r3.z = (f3.z / cdiv1 > 0.f) ? c.z * 0.5f / cdiv1 : 0.f; r3.w = (f3.w / cdiv1 > 0.f) ? c.w * 0.5f / cdiv1 : 0.f;
        *reinterpret_cast<float4*>(dF2 + o) = r2;
        *reinterpret_cast<float4*>(dF3 + o) = r3;
    }
}

// ------------------------------------------------------------------ BatchNorm1d over rows
// column statistics: grid (C/64, slabs); 4 waves stride rows; atomics merge slabs.
__global__ __launch_bounds__(256) void bn_colsum_kernel(const float* __restrict__ X, int M, int C, const float* __restrict__ center,
                                                        float* __restrict__ sum, int square, int rows_per_block) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const float mu = (center && col < C) ? center[col] : 0.f;
    float acc = 0.f;
    if (col < C)
        for (int r = r0 + w; r < r1; r += 4) {
            const float d = X[(int64_t)r * C + col] - mu;
            acc += square ? d * d : d;
        }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < C) atomicAdd(sum + col, sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
}
// float4 form (C % 4 == 0, 16-byte aligned rows): a wave covers 256 columns of a row per load instruction (1 KB) instead of 64
// (the scalar form moved 68 MB in 39.5 us = 1.7 TB/s on the [16640, 1024] relation features of Full-GC); grid (C/256, slabs)
// part != NULL: the slab's column sums go to part[blockIdx.y][C] with plain stores and bn_finalize_kernel adds the slabs in a
// fixed order -- no zero-fill launches, no atomics (256 same-address float atomics per column from 8 XCDs were most of the
// 32 us the pass still took), and bit-reproducible statistics
__global__ __launch_bounds__(256) void bn_colsum_vec_kernel(const float* __restrict__ X, int M, int C, const float* __restrict__ center,
                                                            float* __restrict__ sum, int square, int rows_per_block, float* __restrict__ part) {
    __shared__ float4 sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < C) {
        const float4 mu = center ? *reinterpret_cast<const float4*>(center + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {                                  // four rows in flight per wave
            float4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const float4*>(X + (int64_t)(r + 4 * q) * C + col);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d0 = x[q].x - mu.x, d1 = x[q].y - mu.y, d2 = x[q].z - mu.z, d3 = x[q].w - mu.w;
                if (square) { acc.x += d0 * d0; acc.y += d1 * d1; acc.z += d2 * d2; acc.w += d3 * d3; }
                else { acc.x += d0; acc.y += d1; acc.z += d2; acc.w += d3; }
            }
        }
        for (; r < r1; r += 4) {
            const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)r * C + col);
            const float d0 = x.x - mu.x, d1 = x.y - mu.y, d2 = x.z - mu.z, d3 = x.w - mu.w;
            if (square) { acc.x += d0 * d0; acc.y += d1 * d1; acc.z += d2 * d2; acc.w += d3 * d3; }
            else { acc.x += d0; acc.y += d1; acc.z += d2; acc.w += d3; }
        }
    }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < C) {
        const float4 a = sm[0][lane], b = sm[1][lane], c = sm[2][lane], d = sm[3][lane];
        const float4 t = make_float4(a.x + b.x + c.x + d.x, a.y + b.y + c.y + d.y, a.z + b.z + c.z + d.z, a.w + b.w + c.w + d.w);
        if (part) *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * C + col) = t;
        else { atomicAdd(sum + col, t.x); atomicAdd(sum + col + 1, t.y); atomicAdd(sum + col + 2, t.z); atomicAdd(sum + col + 3, t.w); }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const float* __restrict__ dY, const float* __restrict__ X, int M, int C,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int rows_per_block,
                                                                float* __restrict__ part) {
    // part != NULL: slab sums to part[blockIdx.y][2][C] (dgamma row, dbeta row); bn_sum_slabs_kernel adds them
    __shared__ float4 sg[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    if (col < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + col), rs = *reinterpret_cast<const float4*>(rstd + col);
        for (int r = r0 + w; r < r1; r += 4) {
            const float4 dy = *reinterpret_cast<const float4*>(dY + (int64_t)r * C + col);
            const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)r * C + col);
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
        const float4 tg = make_float4(g0.x + g1.x + g2.x + g3.x, g0.y + g1.y + g2.y + g3.y, g0.z + g1.z + g2.z + g3.z, g0.w + g1.w + g2.w + g3.w);
        const float4 tb = make_float4(b0.x + b1.x + b2.x + b3.x, b0.y + b1.y + b2.y + b3.y, b0.z + b1.z + b2.z + b3.z, b0.w + b1.w + b2.w + b3.w);
        if (part) {
            *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2 + 0) * C + col) = tg;
            *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * 2 + 1) * C + col) = tb;
        } else {
            atomicAdd(dgamma + col, tg.x); atomicAdd(dgamma + col + 1, tg.y); atomicAdd(dgamma + col + 2, tg.z); atomicAdd(dgamma + col + 3, tg.w);
            atomicAdd(dbeta + col, tb.x); atomicAdd(dbeta + col + 1, tb.y); atomicAdd(dbeta + col + 2, tb.z); atomicAdd(dbeta + col + 3, tb.w);
        }
    }
}
// column c of vector j: sum over slabs of part[slab][j][c] in a fixed order.  256 threads = 64 columns x 4 waves, wave w adds
// slabs w, w+4, ... eight loads at a time, LDS combines the four; every thread of the column's lane gets the total.
__device__ __forceinline__ float bn_slab_sum(const float* __restrict__ part, int slabs, int nvec, int j, int C, int c, float (*sm)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C) {
        int k = w;
        for (; k + 28 < slabs; k += 32) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = part[((int64_t)(k + 4 * q) * nvec + j) * C + c];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += x[q];
        }
        for (; k < slabs; k += 4) s += part[((int64_t)k * nvec + j) * C + c];
    }
    __syncthreads();
    sm[w][lane] = s;
    __syncthreads();
    return sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
}
// grid C/64 x 256 threads: out0 / out1 [C] = the two summed vectors of the backward (dgamma, dbeta)
__global__ __launch_bounds__(256) void bn_sum_slabs_kernel(const float* __restrict__ part, int slabs, int C, float* __restrict__ out0,
                                                           float* __restrict__ out1) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float a = bn_slab_sum(part, slabs, 2, 0, C, c, sm), b = bn_slab_sum(part, slabs, 2, 1, C, c, sm);
    if (threadIdx.x < 64 && c < C) { out0[c] = a; out1[c] = b; }
}
// finalise: mean = s/M (pass 0)  |  var -> rstd, running stats (pass 1)
__global__ __launch_bounds__(256) void bn_finalize_kernel(float* __restrict__ mean_or_var, int M, int C, int pass, float* __restrict__ save_mean,
                                                          float* __restrict__ save_rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum, float eps,
                                                          const float* __restrict__ part, int slabs) {
    // part == NULL: grid C/256, thread per column, the sums are in mean_or_var (atomic path)
    // part != NULL: grid C/64, 64 columns x 4 waves: the column sums arrive as per-slab partials (bn_slab_sum)
    __shared__ float sm[4][64];
    int c;
    if (part) {
        c = blockIdx.x * 64 + (threadIdx.x & 63);
        const float tot = bn_slab_sum(part, slabs, 1, 0, C, c, sm);
        if (threadIdx.x >= 64 || c >= C) return;
        mean_or_var[c] = tot;
    } else {
        c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= C) return;
    }
    if (pass == 0) {
        const float m = mean_or_var[c] / (float)M;
        save_mean[c] = m;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    } else {
        const float ss = mean_or_var[c];
        const float var_b = ss / (float)M;
        save_rstd[c] = 1.f / sqrtf(var_b + eps);
        if (running_var) {
            const float var_u = M > 1 ? ss / (float)(M - 1) : var_b;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var_u;
        }
    }
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t total, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ rvar, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float rs = rstd ? rstd[c] : 1.f / sqrtf(rvar[c] + eps);
        Y[i] = (X[i] - mean[c]) * rs * gamma[c] + beta[c];
    }
}
// backward pass 1: dbeta = sum dy, dgamma = sum dy * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dY, const float* __restrict__ X, int M, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows_per_block) {
    __shared__ float sg[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float ag = 0.f, ab = 0.f;
    if (col < C) {
        const float mu = mean[col], rs = rstd[col];
        for (int r = r0 + w; r < r1; r += 4) {
            const float dy = dY[(int64_t)r * C + col];
            ab += dy;
            ag += dy * (X[(int64_t)r * C + col] - mu) * rs;
        }
    }
    sg[w][lane] = ag; sb[w][lane] = ab;
    __syncthreads();
    if (w == 0 && col < C) {
        atomicAdd(dgamma + col, sg[0][lane] + sg[1][lane] + sg[2][lane] + sg[3][lane]);
        atomicAdd(dbeta + col, sb[0][lane] + sb[1][lane] + sb[2][lane] + sb[3][lane]);
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ dX,
                                                           int64_t total, int M, int C, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    const float invM = 1.f / (float)M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xh = (X[i] - mean[c]) * rstd[c];
        dX[i] = gamma[c] * rstd[c] * (dY[i] - dbeta[c] * invM - xh * dgamma[c] * invM);
    }
}
__global__ void zero_f32_kernel(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

inline bool gcn_vec_ok(int L, const void* a, const void* b, const void* c, const void* d) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return L % 4 == 0 && al(a) && al(b) && al(c) && al(d);
}
inline bool bn_vec_ok(int C, const void* a, const void* b, const void* c) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return C % 4 == 0 && al(a) && al(b) && al(c);
}
// rows per workgroup of the float4 column reductions: ~512 workgroups (2 per CU) however tall the matrix is
inline int bn_rows_per_block(int M, int C) {
    const int col_groups = (C + 255) / 256;
    const int slabs = std::max(1, 512 / col_groups);
    return std::max(16, (M + slabs - 1) / slabs);
}

inline int raise_lds(const void* fn, size_t bytes, const char* what) { return subgc::raise_lds_cached(fn, bytes, what); }

}  // namespace

SUBGC_API int subgc_row_argmax_f32(const float* X, int64_t ldx, int rows, int cols, int skip, int64_t* idx, float* val,
                                   void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols > skip && skip >= 0 && ldx >= cols, "row_argmax: bad sizes rows=%d cols=%d skip=%d", rows, cols, skip);
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && idx, "row_argmax: null pointer");
    hipLaunchKernelGGL(row_argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, cols, skip, idx, val, (int32_t*)nullptr);
    return subgc::check_launch("subgc_row_argmax_f32");
}
SUBGC_API int subgc_row_argmax_i32(const float* X, int64_t ldx, int rows, int cols, int skip, int32_t* idx, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && cols > skip && skip >= 0 && ldx >= cols, "row_argmax: bad sizes rows=%d cols=%d skip=%d", rows, cols, skip);
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(X && idx, "row_argmax: null pointer");
    hipLaunchKernelGGL(row_argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, cols, skip, (int64_t*)nullptr,
                       (float*)nullptr, idx);
    return subgc::check_launch("subgc_row_argmax_i32");
}

SUBGC_API int subgc_csr_build(const int64_t* rel_ind, int B, int K, int N, int32_t* ptr, int32_t* edges, void* stream) {
    SUBGC_REQUIRE(B >= 0 && K > 0 && N > 0, "csr_build: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(rel_ind && ptr && edges, "csr_build: null pointer");
    SUBGC_DEBUG_RANGE(rel_ind, 8, (int64_t)B * K, 2, 2, 0, N - 1, -1, "csr_build: rel_ind (node ids of the relations)", stream);
    const size_t lds = sizeof(int) * (K + N + 1);
    hipLaunchKernelGGL(csr_build_kernel, dim3(B, 2), dim3(128), lds, (hipStream_t)stream, rel_ind, B, K, N, ptr, edges);
    return subgc::check_launch("subgc_csr_build");
}

SUBGC_API int subgc_gcn_nodes_fwd(const float* F0, const float* F1, const int32_t* ptr, const int32_t* edges, const float* skip,
                                  float* Xout, uint8_t* act, int B, int N, int K, int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0, "gcn_nodes_fwd: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(F0 && F1 && ptr && edges && Xout, "gcn_nodes_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, 4.0 * B * L * (2.0 * K + (skip ? 2.0 : 1.0) * N));
    const size_t lds = sizeof(int) * (2 * (N + 1) + 2 * K);
    if (gcn_vec_ok(L, F0, F1, skip, Xout) && (reinterpret_cast<uintptr_t>(act) & 3) == 0)
        hipLaunchKernelGGL(gcn_nodes_fwd_vec_kernel, dim3((L / 4 + 255) / 256, B, subgc::gcn_zsplit((L / 4 + 255) / 256, B, N)), dim3(256), lds, s, F0, F1, ptr, edges, skip, Xout, act, B, N, K, L);
    else
        hipLaunchKernelGGL(gcn_nodes_fwd_kernel, dim3((L + 255) / 256, B, subgc::gcn_zsplit((L + 255) / 256, B, N)), dim3(256), lds, s, F0, F1, ptr, edges, skip, Xout, act, B, N, K, L);
    return subgc::check_launch("subgc_gcn_nodes_fwd");
}

SUBGC_API int subgc_gcn_nodes_bwd(const float* dX, const uint8_t* act, const int64_t* rel_ind, const int32_t* ptr, void* dF0,
                                  void* dF1, int out_bf16, int B, int N, int K, int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0, "gcn_nodes_bwd: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(dX && act && rel_ind && ptr && dF0 && dF1, "gcn_nodes_bwd: null pointer");
    SUBGC_DEBUG_RANGE(rel_ind, 8, (int64_t)B * K, 2, 2, 0, N - 1, -1, "gcn_nodes_bwd: rel_ind", stream);
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, 4.0 * B * L * (2.0 * K + 2.0 * K));
    const size_t lds = sizeof(int) * (2 * K) + sizeof(float) * 2 * N;
    const bool vec = gcn_vec_ok(L, dX, nullptr, nullptr, nullptr) && (reinterpret_cast<uintptr_t>(act) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(dF0) | reinterpret_cast<uintptr_t>(dF1)) & (out_bf16 ? 7 : 15)) == 0;
    SUBGC_REQUIRE(vec || !out_bf16, "gcn_nodes_bwd: bf16 gradients need L %% 4 == 0 and aligned rows");
    if (vec)
        hipLaunchKernelGGL(gcn_nodes_bwd_vec_kernel, dim3((L / 4 + 255) / 256, B, subgc::gcn_zsplit((L / 4 + 255) / 256, B, (K + 3) / 4)), dim3(256), lds, s, dX, act, rel_ind, ptr, dF0, dF1, B, N, K, L, out_bf16);
    else
        hipLaunchKernelGGL(gcn_nodes_bwd_kernel, dim3((L + 255) / 256, B, subgc::gcn_zsplit((L + 255) / 256, B, K)), dim3(256), lds, s, dX, act, rel_ind, ptr, static_cast<float*>(dF0), static_cast<float*>(dF1), B, N, K, L);
    return subgc::check_launch("subgc_gcn_nodes_bwd");
}

SUBGC_API int subgc_gcn_edges_fwd(const float* F2, const float* F3, const int64_t* rel_ind, const float* skip, float* Pout, int B,
                                  int N, int K, int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0, "gcn_edges_fwd: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(F2 && F3 && rel_ind && Pout, "gcn_edges_fwd: null pointer");
    SUBGC_DEBUG_RANGE(rel_ind, 8, (int64_t)B * K, 2, 2, 0, N - 1, -1, "gcn_edges_fwd: rel_ind", stream);
    SUBGC_REQUIRE(L % 4 == 0, "gcn_edges_fwd: L must be a multiple of 4 (got %d)", L);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * 2 * (size_t)N * TC + sizeof(int) * 2 * K;
    const bool vec = gcn_vec_ok(L, Pout, skip, nullptr, nullptr);
    int rc = vec ? raise_lds((const void*)gcn_edges_fwd_kernel<true>, lds, "gcn_edges_fwd") : raise_lds((const void*)gcn_edges_fwd_kernel<false>, lds, "gcn_edges_fwd");
    if (rc) return rc;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, 4.0 * B * L * (2.0 * N + (skip ? 2.0 : 1.0) * K));
    if (vec) hipLaunchKernelGGL(gcn_edges_fwd_kernel<true>, dim3((L + TC - 1) / TC, B), dim3(256), lds, s, F2, F3, rel_ind, skip, Pout, B, N, K, L);
    else hipLaunchKernelGGL(gcn_edges_fwd_kernel<false>, dim3((L + TC - 1) / TC, B), dim3(256), lds, s, F2, F3, rel_ind, skip, Pout, B, N, K, L);
    return subgc::check_launch("subgc_gcn_edges_fwd");
}

SUBGC_API int subgc_gcn_edges_bwd(const float* dP, const float* F2, const float* F3, const int32_t* ptr, const int32_t* edges,
                                  float* dF2, float* dF3, int B, int N, int K, int L, void* stream) {
    SUBGC_REQUIRE(B >= 0 && N > 0 && K > 0 && L > 0, "gcn_edges_bwd: bad sizes");
    if (B == 0) return SUBGC_OK;
    SUBGC_REQUIRE(dP && F2 && F3 && ptr && edges && dF2 && dF3, "gcn_edges_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    subgc::ProfScope prof(SUBGC_FAM_GCN, s, 4.0 * B * L * (2.0 * K + 4.0 * N));
    const size_t lds = sizeof(int) * (2 * (N + 1) + 2 * K);
    if (gcn_vec_ok(L, dP, F2, F3, dF2) && (reinterpret_cast<uintptr_t>(dF3) & 15) == 0)
        hipLaunchKernelGGL(gcn_edges_bwd_vec_kernel, dim3((L / 4 + 255) / 256, B, subgc::gcn_zsplit((L / 4 + 255) / 256, B, N)), dim3(256), lds, s, dP, F2, F3, ptr, edges, dF2, dF3, B, N, K, L);
    else
        hipLaunchKernelGGL(gcn_edges_bwd_kernel, dim3((L + 255) / 256, B, subgc::gcn_zsplit((L + 255) / 256, B, N)), dim3(256), lds, s, dP, F2, F3, ptr, edges, dF2, dF3, B, N, K, L);
    return subgc::check_launch("subgc_gcn_edges_bwd");
}

SUBGC_API int subgc_bn_fwd(const float* X, float* Y, int M, int C, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_rstd, int training, float momentum, float eps,
                           void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0, "bn_fwd: bad sizes");
    SUBGC_REQUIRE(X && Y && gamma && beta, "bn_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)M * C;
    const int ew_blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    if (!training) {
        SUBGC_REQUIRE(running_mean && running_var, "bn_fwd(eval): running stats required");
        hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks), dim3(256), 0, s, X, Y, total, C, running_mean, (const float*)nullptr,
                           running_var, eps, gamma, beta);
        return subgc::check_launch("subgc_bn_fwd");
    }
    SUBGC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "bn_fwd: workspace must be 16-byte aligned");
    SUBGC_REQUIRE(save_mean && save_rstd, "bn_fwd(train): save buffers required");
    const bool vec = bn_vec_ok(C, X, save_mean, save_rstd);
    const int rpb = vec ? bn_rows_per_block(M, C) : 512;
    dim3 g(vec ? (C + 255) / 256 : (C + 63) / 64, (M + rpb - 1) / rpb);
    float* part = vec && workspace && ws_bytes >= (size_t)g.y * C * sizeof(float) ? static_cast<float*>(workspace) : nullptr;
    if (!part) {
        hipLaunchKernelGGL(zero_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, save_mean, (int64_t)C);
        hipLaunchKernelGGL(zero_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, save_rstd, (int64_t)C);
    }
    if (vec) hipLaunchKernelGGL(bn_colsum_vec_kernel, g, dim3(256), 0, s, X, M, C, (const float*)nullptr, save_mean, 0, rpb, part);
    else hipLaunchKernelGGL(bn_colsum_kernel, g, dim3(256), 0, s, X, M, C, (const float*)nullptr, save_mean, 0, rpb);
    const dim3 fg(part ? (C + 63) / 64 : (C + 255) / 256);
    hipLaunchKernelGGL(bn_finalize_kernel, fg, dim3(256), 0, s, save_mean, M, C, 0, save_mean, save_rstd,
                       running_mean, running_var, momentum, eps, (const float*)part, (int)g.y);
    if (vec) hipLaunchKernelGGL(bn_colsum_vec_kernel, g, dim3(256), 0, s, X, M, C, (const float*)save_mean, save_rstd, 1, rpb, part);
    else hipLaunchKernelGGL(bn_colsum_kernel, g, dim3(256), 0, s, X, M, C, (const float*)save_mean, save_rstd, 1, rpb);
    hipLaunchKernelGGL(bn_finalize_kernel, fg, dim3(256), 0, s, save_rstd, M, C, 1, save_mean, save_rstd,
                       running_mean, running_var, momentum, eps, (const float*)part, (int)g.y);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks), dim3(256), 0, s, X, Y, total, C, (const float*)save_mean,
                       (const float*)save_rstd, (const float*)nullptr, eps, gamma, beta);
    return subgc::check_launch("subgc_bn_fwd");
}

SUBGC_API int subgc_bn_bwd(const float* dY, const float* X, const float* gamma, const float* save_mean, const float* save_rstd,
                           float* dX, float* dgamma, float* dbeta, int M, int C, void* workspace, size_t ws_bytes, void* stream) {
    SUBGC_REQUIRE(M > 0 && C > 0, "bn_bwd: bad sizes");
    SUBGC_REQUIRE(dY && X && gamma && save_mean && save_rstd && dX && dgamma && dbeta, "bn_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)M * C;
    const bool vec = bn_vec_ok(C, X, dY, save_mean) && bn_vec_ok(C, save_rstd, dgamma, dbeta);
    const int rpb = vec ? bn_rows_per_block(M, C) : 512;
    dim3 g(vec ? (C + 255) / 256 : (C + 63) / 64, (M + rpb - 1) / rpb);
    float* part = vec && workspace && ws_bytes >= (size_t)g.y * 2 * C * sizeof(float) ? static_cast<float*>(workspace) : nullptr;
    if (!part) {
        hipLaunchKernelGGL(zero_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dgamma, (int64_t)C);
        hipLaunchKernelGGL(zero_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dbeta, (int64_t)C);
    }
    if (vec) hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, g, dim3(256), 0, s, dY, X, M, C, save_mean, save_rstd, dgamma, dbeta, rpb, part);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel, g, dim3(256), 0, s, dY, X, M, C, save_mean, save_rstd, dgamma, dbeta, rpb);
    if (part) hipLaunchKernelGGL(bn_sum_slabs_kernel, dim3((C + 63) / 64), dim3(256), 0, s, (const float*)part, (int)g.y, C, dgamma, dbeta);
    const int ew_blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks), dim3(256), 0, s, dY, X, dX, total, M, C, save_mean, save_rstd, gamma,
                       (const float*)dgamma, (const float*)dbeta);
    return subgc::check_launch("subgc_bn_bwd");
}
