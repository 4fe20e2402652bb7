// Developer probe: what a kernel that ONLY reads the corpus reaches, at the corpus sizes of the bench (410 MB,
// 4.1 GB) and for different launch shapes.  The ceiling the prefilter scan is compared with in DESIGN.md: a
// short stream does not reach the steady HBM rate, whatever the arithmetic behind the loads.
//   hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

// Each workgroup reads `per_wg` bytes: contiguous (MODE 0: workgroup g owns [g*per_wg, (g+1)*per_wg)) or
// chunk-interleaved (MODE 1: 128 KB chunks g, g+G, g+2G ...).  DEPTH 16-byte loads per thread in flight.
template <int THREADS, int DEPTH, int MODE>
__global__ __launch_bounds__(THREADS) void read_kernel(const vf4 *__restrict__ src, size_t n_vec, float *sink) {
    const size_t chunk_vec = 128 * 1024 / 16;  // one 32-row tile of 1024 fp32 dims
    const size_t n_chunks = n_vec / chunk_vec;
    const size_t G = gridDim.x, g = blockIdx.x;
    size_t c0, c1, cstep;
    if (MODE == 0) {
        const size_t per = (n_chunks + G - 1) / G;
        c0 = g * per; c1 = c0 + per < n_chunks ? c0 + per : n_chunks; cstep = 1;
    } else {
        c0 = g; c1 = n_chunks; cstep = G;
    }
    vf4 acc = {0, 0, 0, 0};
    for (size_t c = c0; c < c1; c += cstep) {
        const vf4 *p = src + c * chunk_vec + threadIdx.x;
#pragma unroll 1
        for (size_t i = 0; i < chunk_vec; i += (size_t)THREADS * DEPTH) {
            vf4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(p + i + (size_t)d * THREADS);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[g] = acc.x;
}

__global__ void spacer_kernel(float *sink) { if (threadIdx.x == 1024) sink[0] = 1.f; }

template <int THREADS, int DEPTH, int MODE>
static void run(const char *label, const vf4 *src, size_t bytes, int grid, float *sink, bool spaced) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 40;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((read_kernel<THREADS, DEPTH, MODE>), dim3(grid), dim3(THREADS), 0, 0, src, bytes / 16, sink);
    hipDeviceSynchronize();
    double total = 0;
    if (!spaced) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<THREADS, DEPTH, MODE>), dim3(grid), dim3(THREADS), 0, 0, src, bytes / 16, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total = ms / reps;
    } else {  // each launch timed on its own, a small kernel between two reads (as the search has)
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(spacer_kernel, dim3(64), dim3(256), 0, 0, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL((read_kernel<THREADS, DEPTH, MODE>), dim3(grid), dim3(THREADS), 0, 0, src, bytes / 16, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            total += ms / reps;
        }
    }
    printf("%-34s %7.1f MB grid %5d x %3d %s: %8.1f us = %6.0f GB/s\n", label, bytes / 1e6, grid, THREADS,
           spaced ? "single " : "back2back", total * 1e3, bytes / (total * 1e-3) / 1e9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const size_t big = (size_t)1000000 * 4096;
    vf4 *src; float *sink;
    if (hipMalloc(&src, big) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, big);
    const size_t sizes[2] = {(size_t)100000 * 4096 / (128 * 1024) * (128 * 1024), big / (128 * 1024) * (128 * 1024)};
    for (int s = 0; s < 2; ++s) {
        const size_t b = sizes[s];
        for (int spaced = 0; spaced < 2; ++spaced) {
            run<512, 8, 0>("contiguous, 512 thr, depth 8", src, b, 256, sink, spaced);
            run<512, 8, 0>("contiguous, 512 thr, depth 8", src, b, 512, sink, spaced);
            run<512, 4, 0>("contiguous, 512 thr, depth 4", src, b, 1024, sink, spaced);
            run<256, 8, 0>("contiguous, 256 thr, depth 8", src, b, 1024, sink, spaced);
            run<256, 8, 0>("contiguous, 256 thr, depth 8", src, b, 2048, sink, spaced);
            run<256, 4, 0>("contiguous, 256 thr, depth 4", src, b, 3125, sink, spaced);
            run<512, 8, 1>("interleaved, 512 thr, depth 8", src, b, 256, sink, spaced);
            run<512, 8, 1>("interleaved, 512 thr, depth 8", src, b, 512, sink, spaced);
            run<256, 8, 1>("interleaved, 256 thr, depth 8", src, b, 1024, sink, spaced);
            run<256, 8, 1>("interleaved, 256 thr, depth 8", src, b, 2048, sink, spaced);
        }
    }
    return 0;
}
