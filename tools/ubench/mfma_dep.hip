// micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 with NACC independent accumulators per wave (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* d) {
    const int iters = 20000 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, 256>>>(d, 10);
    hipEventRecord(e0);
    k<NACC><<<256, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 4 * NACC;
    printf("NACC=%d  %.1f ns/MFMA/wave  (%.1f cycles @2.4GHz)  %.0f TFLOP/s\n", NACC, ms * 1e6 / n, ms * 1e6 / n * 2.4,
           n * 32768.0 * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<1>(d); run<2>(d); run<4>(d); run<8>(d);
    return 0;
}
