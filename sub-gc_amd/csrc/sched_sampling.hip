// Scheduled sampling (reference AttModel.py:157-167): at training step i >= 1 a row's input word is, with probability
// ss_prob, a draw from the model's own previous-step distribution instead of the ground-truth word.
//   subgc_uniform_f32      counter-based uniforms in [0, 1) (the same Philox-4x32-10 stream family as the dropout masks)
//   subgc_multinomial_rows rows whose selector uniform is below `prob` get tok = inverse-CDF draw from softmax(logits[row])
//                          (index order; the reference's torch.multinomial stream cannot be reproduced elsewhere, the
//                          distribution is the same); other rows keep their word.
#include "common.h"

namespace {

__device__ __forceinline__ void philox_round_(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}

__global__ __launch_bounds__(256) void uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset / 4 + (uint64_t)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round_(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = (c[j] >> 8) * (1.0f / 16777216.0f);
    }
}

// One workgroup per row; thread t owns the contiguous chunk [t*CH, (t+1)*CH) so that the prefix order is the index order.  The row is read
// from memory ONCE, coalesced (thread t takes columns t, t+256, ...), and exp(x - max) is staged in LDS; the chunk sums and the final scan
// then run on LDS in exactly the order of additions the first version used (it read each thread's 152-byte chunk straight from memory,
// three times, 64 cache lines per wave load: 28 us for 160 rows of 9488) -- same picks, bit for bit.
__global__ __launch_bounds__(256) void multinomial_kernel(const float* __restrict__ logits, int64_t ld, int V, const float* __restrict__ u,
                                                          const float* __restrict__ sel_u, float prob, int64_t* __restrict__ tok, int64_t tok_stride) {
    extern __shared__ float ex[];                                  // [V] exp(x - max)
    __shared__ float part[256];
    __shared__ float smf[16];
    __shared__ int pick_s;
    const int r = blockIdx.x;
    if (!(sel_u[r] < prob)) return;                                // workgroup-uniform: this row keeps its ground-truth word
    const float* p = logits + (int64_t)r * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) { const float x = p[c]; ex[c] = x; mx = fmaxf(mx, x); }
    mx = block_max(mx, smf);                                       // (barriers inside: ex[] is complete afterwards)
    for (int c = threadIdx.x; c < V; c += 256) ex[c] = expf(ex[c] - mx);
    __syncthreads();
    const int CH = (V + 255) / 256;
    const int lo = threadIdx.x * CH, hi = min(V, lo + CH);
    float s = 0.f;
    for (int c = lo; c < hi; ++c) s += ex[c];
    part[threadIdx.x] = s;
    if (threadIdx.x == 0) pick_s = V - 1;
    __syncthreads();
    if (threadIdx.x == 0) {                                       // 256 partials: a serial exclusive scan is cheaper than it looks
        float run = 0.f;
        for (int i = 0; i < 256; ++i) { const float v = part[i]; part[i] = run; run += v; }
        smf[0] = run;
    }
    __syncthreads();
    const float target = u[r] * smf[0];
    const float before = part[threadIdx.x];
    const float after = threadIdx.x == 255 ? INFINITY : part[threadIdx.x + 1];
    if (lo < hi && target >= before && target < after) {          // exactly one thread owns the target
        float run = before;
        int pick = hi - 1;
        for (int c = lo; c < hi; ++c) {
            run += ex[c];
            if (target < run) { pick = c; break; }
        }
        pick_s = pick;
    }
    __syncthreads();
    if (threadIdx.x == 0) tok[(int64_t)r * tok_stride] = pick_s;
}

}  // namespace

SUBGC_API int subgc_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    SUBGC_REQUIRE(n >= 0 && offset % 4 == 0, "uniform: bad arguments");
    if (n == 0) return SUBGC_OK;
    SUBGC_REQUIRE(out, "uniform: null pointer");
    const int64_t q = (n + 3) / 4;
    hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)std::min<int64_t>((q + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                       offset);
    return subgc::check_launch("subgc_uniform_f32");
}

SUBGC_API int subgc_multinomial_rows(const float* logits, int64_t ld, int rows, int V, const float* u, const float* sel_u, float prob,
                                     int64_t* tok, int64_t tok_stride, void* stream) {
    SUBGC_REQUIRE(rows >= 0 && V > 0 && ld >= V && tok_stride >= 1, "multinomial_rows: bad sizes");
    if (rows == 0) return SUBGC_OK;
    SUBGC_REQUIRE(logits && u && sel_u && tok, "multinomial_rows: null pointer");
    SUBGC_REQUIRE(V <= 36000, "multinomial_rows: at most 36000 columns (the row is staged in LDS)");
    if (int rc = subgc::raise_lds_cached((const void*)multinomial_kernel, (size_t)V * sizeof(float), "multinomial_rows")) return rc;
    hipLaunchKernelGGL(multinomial_kernel, dim3(rows), dim3(256), (size_t)V * sizeof(float), (hipStream_t)stream, logits, ld, V, u,
                       sel_u, prob, tok, tok_stride);
    return subgc::check_launch("subgc_multinomial_rows");
}
