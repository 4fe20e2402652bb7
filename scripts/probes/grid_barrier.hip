// Probe: what does a grid-wide barrier cost on this chip (256 workgroups of 512 threads, one per CU), against the
// ~4-5 us a kernel boundary costs the five-launch layer of crag_encoder_small.hip?
//   hipcc -O3 -Wno-unused-value --offload-arch=gfx950 scripts/probes/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

struct Bar {
    uint32_t count, gen, abort_, pad;
};

// sense-reversing barrier, bounded spin: returns false once the abort flag is up
__device__ __forceinline__ bool grid_barrier(Bar *b, uint32_t n_wg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t my = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == n_wg - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint32_t spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    __hip_atomic_store(&b->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if ((spins & 1023u) == 0 && __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            }
        }
        ok = __hip_atomic_load(&b->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    }
    __syncthreads();
    return ok;
}

// flag barrier: workgroup b stores the barrier's sequence number into flags[b]; wave 0 of every workgroup polls all
// 256 flags (one 16-byte load per lane).  No read-modify-write on a shared address.
__device__ __forceinline__ bool flag_barrier(uint32_t *flags, uint32_t *abort_, uint32_t n_wg, uint32_t seq) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t l = threadIdx.x;
        uint32_t spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = l * 4 + j;
                const uint32_t v = i < n_wg ? __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seq;
                all = all && (int32_t)(v - seq) >= 0;
            }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) {
                __hip_atomic_store(abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void flag_kernel(uint32_t *flags, uint32_t *abort_, uint32_t *data, int phases,
                                                   uint32_t *errors, uint32_t base) {
    const uint32_t n = gridDim.x;
    uint32_t seq = flags[blockIdx.x];  // every workgroup leaves the same value behind
    for (int ph = 0; ph < phases; ++ph) {
        if (threadIdx.x >= 448) data[blockIdx.x * 64 + (threadIdx.x - 448)] = base + ph * 1000003u + blockIdx.x * 64 + (threadIdx.x - 448);
        if (!flag_barrier(flags, abort_, n, ++seq)) return;
        const uint32_t other = (blockIdx.x + 1 + 37 * ph) % n;
        if (threadIdx.x >= 192 && threadIdx.x < 256) {
            const uint32_t l = threadIdx.x - 192;
            const uint32_t v = data[other * 64 + l];
            if (v != base + ph * 1000003u + other * 64 + l) atomicAdd(errors, 1u);
        }
        if (!flag_barrier(flags, abort_, n, ++seq)) return;
    }
}

// each phase: every workgroup writes a value the NEXT phase of another workgroup (another XCD) reads and checks
__global__ __launch_bounds__(512) void bar_kernel(Bar *b, uint32_t *data, int phases, uint32_t *errors, uint32_t base) {
    const uint32_t n = gridDim.x;
    for (int ph = 0; ph < phases; ++ph) {
        // written by the LAST wave, read by wave 3: the barrier's fences run on wave 0 only
        if (threadIdx.x >= 448) data[blockIdx.x * 64 + (threadIdx.x - 448)] = base + ph * 1000003u + blockIdx.x * 64 + (threadIdx.x - 448);
        if (!grid_barrier(b, n)) return;
        const uint32_t other = (blockIdx.x + 1 + 37 * ph) % n;  // neighbouring blockIdx = another XCD
        if (threadIdx.x >= 192 && threadIdx.x < 256) {
            const uint32_t l = threadIdx.x - 192;
            const uint32_t v = data[other * 64 + l];
            if (v != base + ph * 1000003u + other * 64 + l) atomicAdd(errors, 1u);
        }
        if (!grid_barrier(b, n)) return;  // nobody overwrites before everybody has read
    }
}

__global__ __launch_bounds__(512) void empty_kernel(uint32_t *data) {
    if (threadIdx.x == 0 && data == nullptr) __builtin_trap();
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    Bar *b;
    uint32_t *data, *err;
    hipMalloc(&b, sizeof(Bar));
    hipMemset(b, 0, sizeof(Bar));
    hipMalloc(&data, 1024 * 64 * 4);
    hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {cus, cus / 2, 32}) {
        for (int phases : {1, 50, 200}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(bar_kernel, dim3(grid), dim3(512), 0, st, b, data, phases, err, (uint32_t)(rep * 77u));
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            uint32_t h_err, h_bar[4];
            hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
            hipMemcpy(h_bar, b, 16, hipMemcpyDeviceToHost);
            printf("grid %3d phases %3d: %8.2f us per launch, %6.2f us per barrier (2 per phase)  errors %u abort %u\n", grid,
                   phases, best * 1e3f, best * 1e3f / (2 * phases), h_err, h_bar[2]);
        }
    }
    uint32_t *flags;
    hipMalloc(&flags, 1024 * 4);
    hipMemset(flags, 0, 1024 * 4);
    for (int grid : {cus, cus / 2, 32}) {
        for (int phases : {1, 50, 200}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(flags, 0, 1024 * 4);
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(flag_kernel, dim3(grid), dim3(512), 0, st, flags, &b->abort_, data, phases, err, (uint32_t)(rep * 77u));
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            uint32_t h_err, h_bar[4];
            hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
            hipMemcpy(h_bar, b, 16, hipMemcpyDeviceToHost);
            printf("FLAGS grid %3d phases %3d: %8.2f us per launch, %6.2f us per barrier (2 per phase)  errors %u abort %u\n", grid,
                   phases, best * 1e3f, best * 1e3f / (2 * phases), h_err, h_bar[2]);
        }
    }
    // kernel-boundary cost for comparison: 200 empty kernels back to back
    hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(cus), dim3(512), 0, st, data);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("200 empty kernels of %d workgroups back to back: %.2f us each\n", cus, ms * 1e3f / 200);
    return 0;
}
