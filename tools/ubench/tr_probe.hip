// Probe of ds_read_b64_tr_b16 on gfx950: every lane reads 8 bytes at its own address; which 16-bit elements land where?
// LDS holds lds[i] = i (uint16).  Pattern A: lane l -> byte address 8*l (elements 4l..4l+3 without the transpose).
// Pattern B: a [4 k][16 n] block layout with row stride `stride` elements: lane l (within its 16-lane group g = l/16) points at
// row (l%16)/4 ... see host print.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void probe(uint16_t* out, int pattern, int stride) {
    __shared__ uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;
    if (pattern == 0) elem = 4 * l;
    else { const int g = l >> 4, i = l & 15; elem = g * 1024 + (i >> 2) * stride + (i & 3) * 4; }   // group g: block [4 rows][16 cols], row stride `stride`
    uint32_t addr = (uint32_t)(uintptr_t)(lds) + elem * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int pat = 0; pat < 3; ++pat) {
        const int stride = pat == 2 ? 128 : 16;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat == 0 ? 0 : 1, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d (stride %d)\n", pat, stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
