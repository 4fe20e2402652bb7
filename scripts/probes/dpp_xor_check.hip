// Developer probe: the DPP lane-exchange patterns used by the bound sort deliver lane ^ X (checked against ds_swizzle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int X>
__device__ __forceinline__ uint32_t swz_xor(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (X << 10) | 0x1f); }
template <int X>
__device__ __forceinline__ uint32_t dpp_xor(uint32_t v, int lane) {
    if constexpr (X == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    else if constexpr (X == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    else if constexpr (X == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xF, 0xF, true);
    else if constexpr (X == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);
    else if constexpr (X == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);
    else if constexpr (X == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);
    else if constexpr (X == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x12C, 0xF, 0xF, true);
        const uint32_t dn = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true);
        return (lane & 4) ? dn : up;
    } else return swz_xor<X>(v);
}
__global__ void k(uint32_t *out) {
    const int lane = threadIdx.x;
    const uint32_t v = 1000u + lane;
    uint32_t bad = 0;
    bad |= (dpp_xor<1>(v, lane) != swz_xor<1>(v)) << 0;
    bad |= (dpp_xor<2>(v, lane) != swz_xor<2>(v)) << 1;
    bad |= (dpp_xor<3>(v, lane) != swz_xor<3>(v)) << 2;
    bad |= (dpp_xor<4>(v, lane) != swz_xor<4>(v)) << 3;
    bad |= (dpp_xor<7>(v, lane) != swz_xor<7>(v)) << 4;
    bad |= (dpp_xor<8>(v, lane) != swz_xor<8>(v)) << 5;
    bad |= (dpp_xor<15>(v, lane) != swz_xor<15>(v)) << 6;
    bad |= (swz_xor<4>(v) != 1000u + (lane ^ 4)) << 7;
    out[lane] = bad;
}
int main() {
    uint32_t *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    uint32_t all = 0;
    for (int i = 0; i < 64; ++i) all |= h[i];
    printf("dpp_xor mismatch mask: 0x%x (%s)\n", all, all ? "BAD" : "ok");
    return all ? 1 : 0;
}
