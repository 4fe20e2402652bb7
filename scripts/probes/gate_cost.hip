// Developer probe: what an early-exit ("gate") kernel costs in a stream of dependent kernels, by launch shape.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void gate_kernel(const unsigned *flag, float *out) {
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    if (*flag == 0u) return;
    lds[threadIdx.x] = 1;
    __syncthreads();
    out[blockIdx.x] = lds[(threadIdx.x + 1) % THREADS];
}
__global__ void work_kernel(float *out, int n) {  // a few microseconds of dependent work
    float v = out[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    out[threadIdx.x] = v;
}
template <int THREADS, int LDS_BYTES>
static void run(const char *label, int grid, unsigned *flag, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    float base, with;
    for (int pass = 0; pass < 2; ++pass) {
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(work_kernel, dim3(64), dim3(256), 0, 0, out, 2000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(work_kernel, dim3(64), dim3(256), 0, 0, out, 2000);
            if (pass == 1) hipLaunchKernelGGL((gate_kernel<THREADS, LDS_BYTES>), dim3(grid), dim3(THREADS), 0, 0, flag, out);
        }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        (pass ? with : base) = ms * 1e3f / reps;
    }
    printf("%-44s grid %4d: work alone %6.2f us, work + gate %6.2f us -> gate costs %5.2f us\n", label, grid, base, with, with - base);
}
int main() {
    unsigned *flag; float *out;
    hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
    run<512, 131072>("512 threads, 128 KiB LDS", 256, flag, out);
    run<512, 131072>("512 threads, 128 KiB LDS", 64, flag, out);
    run<512, 0>("512 threads, no LDS", 256, flag, out);
    run<256, 0>("256 threads, no LDS", 256, flag, out);
    run<256, 0>("256 threads, no LDS", 64, flag, out);
    run<64, 0>("64 threads, no LDS", 1, flag, out);
    return 0;
}
