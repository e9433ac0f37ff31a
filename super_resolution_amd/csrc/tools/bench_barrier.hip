// Measurement tool (not part of the library): what a dependency between two small convs costs as a dispatch
// boundary (a chain of dependent launches on one stream) and as a grid-wide barrier inside one launch
// (agent-scope release, counter, bounded spin, agent-scope acquire).  Every step a workgroup writes a 1 KB
// record and, after the dependency, checks the records of two other workgroups; stale reads are counted.
//   bench_barrier [workgroups=256] [steps=2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void write_record(unsigned *buf, int wg, unsigned value) {
    buf[wg * 256 + threadIdx.x] = value;
}
__device__ __forceinline__ unsigned check_records(const unsigned *buf, int wg, int nwg, unsigned value) {
    const unsigned a = buf[((wg + 1) % nwg) * 256 + threadIdx.x];
    const unsigned b = buf[((wg + 37) % nwg) * 256 + threadIdx.x];
    return (a != value) + (b != value);
}

__global__ __launch_bounds__(256) void step_kernel(unsigned *cur, const unsigned *prev, unsigned step, unsigned *stale) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    unsigned bad = step ? check_records(prev, wg, nwg, step - 1) : 0;
    write_record(cur, wg, step);
    if (bad) atomicAdd(stale, bad);
}

// counter[0] = arrivals, counter[1] = abort flag
__global__ __launch_bounds__(256) void barrier_kernel(unsigned *buf0, unsigned *buf1, unsigned steps, unsigned *counter,
                                                      unsigned *stale) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    unsigned bad = 0;
    for (unsigned s = 0; s < steps; ++s) {
        unsigned *cur = (s & 1) ? buf1 : buf0;
        write_record(cur, wg, s);
        __syncthreads();                                   // every wave's stores are issued
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // agent scope: stores drained, L2 written back
            __hip_atomic_fetch_add(&counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (s + 1) * nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(&counter[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 22) || __hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&counter[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // agent scope: this CU's L1 and the L2 invalidated
        }
        __syncthreads();
        if (__hip_atomic_load(&counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // every wave leaves
        bad += check_records(cur, wg, nwg, s);
    }
    if (bad) atomicAdd(stale, bad);
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const unsigned steps = argc > 2 ? (unsigned)atoi(argv[2]) : 2000;
    if (nwg < 1 || nwg > 512 || steps < 1 || steps > 100000) { printf("workgroups 1..512, steps 1..100000\n"); return 1; }
    unsigned *buf0, *buf1, *ctl;
    CHECK(hipMalloc(&buf0, (size_t)nwg * 1024));
    CHECK(hipMalloc(&buf1, (size_t)nwg * 1024));
    CHECK(hipMalloc(&ctl, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    unsigned host[4];
    float ms = 0.f;

    for (int rep = 0; rep < 2; ++rep) {                    // rep 0 warms up
        CHECK(hipMemsetAsync(ctl, 0, 64, st));
        CHECK(hipEventRecord(e0, st));
        for (unsigned s = 0; s < steps; ++s)
            hipLaunchKernelGGL(step_kernel, dim3(nwg), dim3(256), 0, st, (s & 1) ? buf1 : buf0, (s & 1) ? buf0 : buf1, s, ctl + 2);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipMemcpy(host, ctl, 16, hipMemcpyDeviceToHost));
    printf("dispatch boundary: %u dependent launches of %d workgroups: %.2f us per step, stale reads %u\n", steps, nwg,
           ms * 1e3 / steps, host[2]);

    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemsetAsync(ctl, 0, 64, st));
        CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(256), 0, st, buf0, buf1, steps, ctl, ctl + 2);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipMemcpy(host, ctl, 16, hipMemcpyDeviceToHost));
    printf("grid barrier:      %u steps inside one launch of %d workgroups: %.2f us per step, stale reads %u, aborted %u\n",
           steps, nwg, ms * 1e3 / steps, host[2], host[1]);
    return 0;
}
