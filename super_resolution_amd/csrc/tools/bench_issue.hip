// Issue cost of the two ways a workgroup can stage 1 KiB pieces from global memory into LDS (not part of the library):
// an LDS-DMA instruction (buffer_load ... lds) per piece, or a buffer load into registers plus a ds_write per piece.
// Four waves per workgroup, one workgroup per CU; every wave issues `n` pieces back to back between two cycle stamps,
// with sources in range (a 64 KiB window per workgroup: L2 hits) or out of range (no traffic).
// Usage: bench_issue [pieces per wave]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: LDS-DMA, 1: load to registers + ds_write, 2: load to registers only
__global__ __launch_bounds__(256, 1) void issue_kernel(const unsigned char *src, int bytes, int n, int oob, unsigned long long *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(src) + (size_t)blockIdx.x * 65536, 0, 65536, 0x00020000);
    const int vo = oob ? 0x7fffff00 : wave * 16384 + lane * 16;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    v4u acc = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int so = __builtin_amdgcn_readfirstlane(((i + k) & 15) * 1024);
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + wave * 16384 + ((i + k) & 15) * 1024), 16, vo, so, 0, 0);
            } else {
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
                if (MODE == 1)
                    *reinterpret_cast<v4u *>(smem + wave * 16384 + ((i + k) & 15) * 1024 + lane * 16) = v;
                else
                    acc ^= v;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) {
        out[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
        out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) out[0] = 0;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    const int grid = 256;
    unsigned char *src;
    unsigned long long *out;
    CK(hipMalloc(&src, (size_t)grid * 65536));
    CK(hipMemset(src, 1, (size_t)grid * 65536));
    CK(hipMalloc(&out, grid * 4 * 2 * 8));
    std::vector<unsigned long long> h(grid * 8);
    const char *names[3] = {"LDS-DMA", "load + ds_write", "load only"};
    for (int oob = 1; oob >= 0; --oob)
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(issue_kernel<0>, dim3(grid), dim3(256), 65536, 0, src, 65536, n, oob, out);
                if (mode == 1) hipLaunchKernelGGL(issue_kernel<1>, dim3(grid), dim3(256), 65536, 0, src, 65536, n, oob, out);
                if (mode == 2) hipLaunchKernelGGL(issue_kernel<2>, dim3(grid), dim3(256), 65536, 0, src, 65536, n, oob, out);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> a, b;
            for (int i = 0; i < grid * 4; ++i) a.push_back((double)h[2 * i] / n), b.push_back((double)h[2 * i + 1] / n);
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            printf("%-16s %-12s: %6.1f cycles of issue per piece (median over waves), %6.1f until all have landed\n", names[mode],
                   oob ? "out of range" : "L2 hits", a[a.size() / 2], b[b.size() / 2]);
        }
    return 0;
}
