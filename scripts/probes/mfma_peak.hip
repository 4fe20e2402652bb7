// Calibration probe (developer tool, not part of the library): sustained fp32 MFMA rate and shader clock of
// the whole chip under the scan kernel's issue pattern (8 waves per CU, runs of 8 v_mfma_f32_32x32x2_f32 per
// accumulator).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// `random` = 1: operands are 16 different pseudo-random values per lane (data toggling as in a real scan);
// 0: the same two values throughout (the low-power case)
__global__ __launch_bounds__(512) void probe(float *out, unsigned long long *clk, int iters, float seed, int random) {
    f32x16 acc0 = {0}, acc1 = {0};
    float av[8], bv[8];
    unsigned rs = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        rs = rs * 1664525u + 1013904223u;
        av[r] = random ? ((int)(rs >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.03f : seed + threadIdx.x;
        rs = rs * 1664525u + 1013904223u;
        bv[r] = random ? ((int)(rs >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.03f : seed * 0.5f;
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bv[r], acc0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[r], av[7 - r], acc1, 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

int main(int argc, char **argv) {
    const int G = 256;
    const int T = argc > 1 ? atoi(argv[1]) : 512;  // 512 = 2 waves per SIMD, 256 = 1 wave per SIMD
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, G * T * 4);
    hipMalloc(&clk, G * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int random : {0, 1})
    for (int iters : {2000, 20000, 200000}) {
        printf("%s operands: ", random ? "random" : "constant");
        hipLaunchKernelGGL(probe, dim3(G), dim3(T), 0, 0, out, clk, iters, 1.0f, random);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(G), dim3(T), 0, 0, out, clk, iters, 1.0f, random);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * G);
        hipMemcpy(h.data(), clk, G * 16, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int g = 0; g < G; ++g) { cyc += h[2 * g]; wall += h[2 * g + 1]; }
        cyc /= G; wall /= G;
        const double flop = (double)G * (T / 64) /*waves*/ * iters * 16.0 * (32.0 * 32 * 2 * 2);
        printf("iters=%d  kernel=%.3f ms  %.1f TFLOP/s  shader clock counter: %.0f ticks in %.0f x10ns -> %.3f ticks/ns;  "
               "MFMA cycles per instruction per SIMD (2 waves): %.1f ticks\n",
               iters, ms, flop / (ms * 1e-3) / 1e12, cyc, wall, cyc / (wall * 10.0), cyc / (iters * 16.0 * 2));
    }
    return 0;
}
