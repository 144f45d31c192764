// bf16 storage helpers shared by the kernels that can write (or read) GEMM-only activations as bf16: the "bf16" compute
// type of BASELINE configs 3 and 5 keeps every tensor whose only consumers are GEMMs in bf16 (raw uint16 bit patterns).
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

__device__ __forceinline__ uint32_t subgc_f2bf(float x) {      // round-to-nearest-even; NaN stays a quiet NaN
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float subgc_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint2 subgc_pack4(float a, float b, float c, float d) {
    uint2 o;
    o.x = subgc_f2bf(a) | (subgc_f2bf(b) << 16);
    o.y = subgc_f2bf(c) | (subgc_f2bf(d) << 16);
    return o;
}
// store VW (1 or 4) consecutive values to a destination that is fp32 (b16 == 0) or bf16; `idx` in elements
template <int VW>
__device__ __forceinline__ void subgc_store_act(void* base, int64_t idx, const float (&o)[VW], int b16) {
    if (b16) {
        uint16_t* p = static_cast<uint16_t*>(base) + idx;
        if (VW == 4) *reinterpret_cast<uint2*>(p) = subgc_pack4(o[0], o[1 % VW], o[2 % VW], o[3 % VW]);
        else *p = (uint16_t)subgc_f2bf(o[0]);
    } else {
        float* p = static_cast<float*>(base) + idx;
        if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % VW], o[2 % VW], o[3 % VW]);
        else *p = o[0];
    }
}
// load 4 consecutive bf16 as floats (8-byte aligned)
__device__ __forceinline__ float4 subgc_load4_bf(const uint16_t* p) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
}
