// Memory-system rates of the GPU for the access mixes the conv kernels have (not part of the library): streaming reads,
// streaming writes, a copy, and reads : writes = 4 : 1, over buffers far larger than the 256 MB memory-side cache.
// Usage: membw [MiB per buffer]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup streams whole 4 KiB pieces: READS pieces read for every WRITES pieces written
template <int READS, int WRITES>
__global__ __launch_bounds__(256) void mix_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t pieces,
                                                  unsigned *sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
        const size_t i = p * 256 + threadIdx.x;
        if (READS > 0 && (WRITES == 0 || p % (READS + WRITES) < READS)) {
            const uint4 v = src[i];
            acc.x ^= v.x, acc.y ^= v.y, acc.z ^= v.z, acc.w ^= v.w;
        } else {
            dst[i] = make_uint4((unsigned)i, acc.x, 2u, 3u);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;   // keeps the loads alive
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t pieces) {
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) dst[p * 256 + threadIdx.x] = src[p * 256 + threadIdx.x];
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 2048;
    const size_t bytes = mib << 20, pieces = bytes / 4096;
    uint4 *a, *b;
    unsigned *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8, reps = 10;
    auto timed = [&](const char *name, auto launch, double moved) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %6.2f TB/s (%zu MiB per buffer, %.1f us per pass)\n", name, moved * reps / (ms * 1e-3) / 1e12, mib,
               ms * 1e3 / reps);
    };
    timed("read only", [&] { hipLaunchKernelGGL((mix_kernel<1, 0>), dim3(grid), dim3(256), 0, 0, a, b, pieces, sink); }, (double)bytes);
    timed("write only", [&] { hipLaunchKernelGGL((mix_kernel<0, 1>), dim3(grid), dim3(256), 0, 0, a, b, pieces, sink); }, (double)bytes);
    timed("reads : writes = 4 : 1", [&] { hipLaunchKernelGGL((mix_kernel<4, 1>), dim3(grid), dim3(256), 0, 0, a, b, pieces, sink); }, (double)bytes);
    timed("reads : writes = 1 : 1", [&] { hipLaunchKernelGGL((mix_kernel<1, 1>), dim3(grid), dim3(256), 0, 0, a, b, pieces, sink); }, (double)bytes);
    timed("copy (read + written bytes)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, pieces); }, 2.0 * bytes);
    return 0;
}
