// micro-benchmark: sustained global-store rate of G workgroups (one per CU when G <= 256), two store patterns:
//   "row"  : a wave instruction writes 1 KB contiguous (64 lanes x 16 B) -- a fill kernel's pattern
//   "tile" : a wave instruction writes 32 rows x 2 x 16 B, rows `pitch` bytes apart -- the pattern of an MFMA C^T accumulator tile
//            (lane = row, register quad = 4 consecutive columns) as gemm_bf16.hip's epilogue issues it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(1024) void k_row(float4* out, size_t per_wg_f4, int reps) {
    float4* base = out + (size_t)blockIdx.x * per_wg_f4;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i < per_wg_f4; i += 1024) base[i] = v;
}
// MODE 0: accumulator pattern (lane = row, quad = 4 consecutive columns: 32 rows x 2 x 16 B per instruction)
// MODE 1: row pattern on the same tile (a wave writes one 1 KB row segment per instruction, 16 rows per wave)
// every pass writes a DIFFERENT 256 x 256 tile (column block r of the workgroup's 256 rows), so nothing is absorbed by a cache
template <int MODE>
__global__ __launch_bounds__(1024) void k_tile(float* out, size_t per_wg_floats, int pitch_floats, int reps) {
    float* wg = out + (size_t)blockIdx.x * per_wg_floats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int r = 0; r < reps; ++r) {
        float* base = wg + (size_t)r * 256;
        if (MODE == 0) {
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int q = 0; q < 4; ++q) {
                        const int m = wm + a * 32 + (lane & 31), n = wn + b * 32 + 8 * q + 4 * (lane >> 5);
                        *reinterpret_cast<float4*>(base + (size_t)m * pitch_floats + n) = v;
                    }
        } else {
            for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(base + (size_t)(wave * 16 + i) * pitch_floats + lane * 4) = v;
        }
    }
}
int main(int argc, char** argv) {
    const size_t total = (size_t)6 << 30;          // 512 workgroups x 256 rows x 9488 floats = 4.97 GB for the tile pattern
    float* d; hipMalloc(&d, total + (64 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {1, 8, 32, 64, 128, 256, 512};
    for (int g : grids) {
        // row pattern: each workgroup streams 4 MB x reps
        {
            const size_t per = (4 << 20) / 16;
            const int reps = 8;
            k_row<<<g, 1024>>>(reinterpret_cast<float4*>(d), per, 1);
            hipEventRecord(e0);
            k_row<<<g, 1024>>>(reinterpret_cast<float4*>(d), per, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)g * per * 16 * reps;
            printf("row   G=%3d  %8.1f GB/s total  %7.2f GB/s per workgroup\n", g, bytes / ms * 1e-6, bytes / ms * 1e-6 / g);
        }
        for (int mode = 0; mode < 2; ++mode) {
            const int pitch = 9472;                      // ~ the logit product's row pitch (floats), 37 column blocks of 256
            const size_t per = (size_t)256 * pitch;      // 256 rows of the output per workgroup: disjoint row blocks
            const int reps = 37;                         // one pass per column block: the workgroup writes its 256 x 9472 slab once
            (mode ? k_tile<1> : k_tile<0>)<<<g, 1024>>>(d, per, pitch, 1);
            hipEventRecord(e0);
            (mode ? k_tile<1> : k_tile<0>)<<<g, 1024>>>(d, per, pitch, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)g * 256 * 256 * 4 * reps;
            printf("%s G=%3d  %8.1f GB/s total  %7.2f GB/s per workgroup  (%.1f us per 256 KB tile)\n", mode ? "tile/rows " : "tile/accum", g,
                   bytes / ms * 1e-6, bytes / ms * 1e-6 / g, ms * 1e3 / reps);
        }
    }
    return 0;
}
