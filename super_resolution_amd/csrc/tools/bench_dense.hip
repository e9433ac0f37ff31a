// Timing tool for the one-launch dense block (not part of the library): repeated launches on one buffer with
// hipEvent timing, and per-workgroup stamps of the phases.  Usage: bench_dense [debug_bits] [N] [H] [W] [mask]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../sr355_internal.h"

using namespace sr355;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int debug = argc > 1 ? atoi(argv[1]) : 0;
    const int N = argc > 2 ? atoi(argv[2]) : 16, H = argc > 3 ? atoi(argv[3]) : 32, W = argc > 4 ? atoi(argv[4]) : 32;
    const int with_mask = argc > 5 ? atoi(argv[5]) : 0;
    const size_t px = (size_t)N * H * W;
    const size_t buf_bytes = px * 24 * 16;
    std::vector<unsigned short> host(buf_bytes / 2);
    srand(1);
    for (auto &v : host) v = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));   // bf16 around +-0.01..0.03
    unsigned char *buf, *nxt, *mask;
    float *bias, *f32;
    CK(hipMalloc(&buf, buf_bytes));
    CK(hipMalloc(&nxt, buf_bytes));
    CK(hipMalloc(&mask, buf_bytes));
    CK(hipMalloc(&f32, px * 64 * 4));
    CK(hipMemcpy(buf, host.data(), buf_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(mask, host.data(), buf_bytes, hipMemcpyHostToDevice));
    CK(hipMemset(f32, 0, px * 64 * 4));
    CK(hipMalloc(&bias, 64 * 4));
    CK(hipMemset(bias, 0, 64 * 4));
    DenseArgs d;
    memset(&d, 0, sizeof(d));
    const int n_sets = 8;   // rotate weight sets as consecutive dense blocks do (cold in L2, warm in the memory-side cache)
    std::vector<unsigned char *> wsets;
    for (int s = 0; s < n_sets; ++s)
        for (int k = 0; k < 5; ++k) {
            const size_t bytes = (size_t)(8 + 4 * k) * 9 * (k < 4 ? 32 : 64) * 16;
            std::vector<unsigned short> w(bytes / 2);
            for (auto &v : w) v = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
            unsigned char *wp;
            CK(hipMalloc(&wp, bytes));
            CK(hipMemcpy(wp, w.data(), bytes, hipMemcpyHostToDevice));
            wsets.push_back(wp);
        }
    unsigned long long *trace;
    const int grid = N * ((H + 1) / 2);
    CK(hipMalloc(&trace, (size_t)grid * 32 * 8));
    CK(hipMemset(trace, 0, (size_t)grid * 32 * 8));
    d.buf = buf;
    d.growth_out = buf;
    d.slope = with_mask ? 1.f : 0.2f;
    d.mask = with_mask ? mask : nullptr;
    d.mask_units = 24;
    d.mask_off[0] = 20, d.mask_off[1] = 16, d.mask_off[2] = 12, d.mask_off[3] = 8;
    d.mask_slope = 0.2f;
    d.N = N, d.H = H, d.W = W, d.units = 24;
    d.debug = debug;
    d.trace = trace;
    d.tail.res1_w = d.tail.res2_w = 1.f;
    d.tail.slope = 1.f, d.tail.mask_slope = 1.f;
    d.tail.res1 = f32, d.tail.res1_scale = 0.2f;
    d.tail.out_f32 = f32;
    d.tail.out_act = nxt, d.tail.out_units = 24;
    d.tail.cout_real = d.tail.cout_pad = 64, d.tail.f32_units = 16;
    auto launch = [&](int it) {
        for (int k = 0; k < 5; ++k) d.w[k] = wsets[(it % n_sets) * 5 + k], d.bias[k] = bias;
        return launch_dense_block(d, 0);
    };
    for (int i = 0; i < 20; ++i)
        if (launch(i)) { printf("launch failed: %s\n", sr355_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 200;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("debug=%d N=%d H=%d W=%d mask=%d grid=%d: %.2f us per launch (back to back)\n", debug, N, H, W, with_mask, grid,
           ms * 1e3 / reps);
    std::vector<unsigned long long> t((size_t)grid * 32);
    CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
    const char *names[12] = {"start", "prologue", "step1 loop", "step1 epi", "step2 loop", "step2 epi", "step3 loop", "step3 epi", "step4 loop", "step4 epi", "conv5 loop", "end"};
    unsigned long long w0 = ~0ull;
    for (int b = 0; b < grid; ++b) w0 = std::min(w0, t[(size_t)b * 32 + 16]);
    for (int k = 1; k < 12; ++k) {
        std::vector<double> dc, dw, ew;
        for (int b = 0; b < grid; ++b) {
            dc.push_back((double)(t[(size_t)b * 32 + k] - t[(size_t)b * 32 + k - 1]));
            dw.push_back((double)(t[(size_t)b * 32 + 16 + k] - t[(size_t)b * 32 + 16 + k - 1]) / 100.0);
            ew.push_back((double)(t[(size_t)b * 32 + 16 + k] - w0) / 100.0);
        }
        std::sort(dc.begin(), dc.end());
        std::sort(dw.begin(), dw.end());
        std::sort(ew.begin(), ew.end());
        printf("  %-11s median %7.0f cycles %6.2f us (p90 %6.2f us); done at median %6.2f us, max %6.2f us\n", names[k],
               dc[grid / 2], dw[grid / 2], dw[grid * 9 / 10], ew[grid / 2], ew[grid - 1]);
    }
    return 0;
}
