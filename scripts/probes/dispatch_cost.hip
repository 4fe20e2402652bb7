// Developer probe: cost of dispatching many short workgroups that each hold a lot of LDS and VGPRs (the
// attention kernel's shape: 256 threads, ~72 KB LDS, ~240 VGPRs) vs a persistent grid looping over the same items.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PERSIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k(float *out, const float *in, int items) {
    __shared__ float lds[18000];  // 72 KB
    float acc[200];
    for (int it = blockIdx.x; it < items; it += PERSIST ? gridDim.x : items) {
        const float v = in[(it * 256 + threadIdx.x) & 0xffff];
#pragma unroll
        for (int i = 0; i < 200; ++i) acc[i] = v * (float)i;
        lds[threadIdx.x] = v;
        __syncthreads();
        float s = lds[(threadIdx.x + 1) & 255];
#pragma unroll
        for (int i = 0; i < 200; ++i) s += acc[i] * s;   // keeps the registers alive
        out[(size_t)it * 256 + threadIdx.x] = s;
        __syncthreads();
    }
}
int main() {
    const int items = 16384;
    float *out, *in;
    hipMalloc(&out, (size_t)items * 256 * 4);
    hipMalloc(&in, 65536 * 4);
    hipMemset(in, 0, 65536 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(items), dim3(256), 0, 0, out, in, items);
            else hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, out, in, items);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s: %d items in %.1f us (%.1f ns per item)\n", mode ? "persistent 512 WGs" : "one WG per item", items, ms * 1e3, ms * 1e6 / items);
        }
    }
    return 0;
}
