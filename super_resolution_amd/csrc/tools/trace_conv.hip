// Timeline of one conv3x3 ring launch: every workgroup stamps the wall clock (100 MHz) at fixed
// points; this prints when workgroups start and end and how long each segment takes.
//   trace_conv <cin_units> <cout> [res: 0|1|2] [N] [H] [W]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../sr355_internal.h"

namespace sr355 {
extern unsigned long long *g_conv_trace;
}
using namespace sr355;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                          \
        }                                                                      \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 tbf16x8;
typedef __attribute__((ext_vector_type(16))) float tf32x16;

// Pure matrix-pipe loop: NACC independent accumulators per wave, no memory traffic.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, unsigned long long *clk, int random_data) {
    tf32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // NACC different operand pairs: constants, or pseudo-random values in [-1, 1) (data toggling costs power)
    tbf16x8 x[NACC], y[NACC];
    unsigned rs = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 8; ++e) {
            rs = rs * 1664525u + 1013904223u;
            const float a = random_data ? ((rs >> 8) & 0xffff) / 32768.f - 1.f : (float)(threadIdx.x & 7);
            rs = rs * 1664525u + 1013904223u;
            const float b = random_data ? ((rs >> 8) & 0xffff) / 32768.f - 1.f : 1.f;
            x[i][e] = (__bf16)a, y[i][e] = (__bf16)b;
        }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0, clk[1] = w1 - w0;
}

static int mfma_peak() {
    float *out;
    unsigned long long *clk, h[2];
    if (hipMalloc(&out, 4096) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int random_data = 0; random_data <= 1; ++random_data)
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
        for (int iters : {20000, 200000}) {
            const int grid = 256 * wg_per_cu;
            auto kern = mfma_peak_kernel<4>;
            kern<<<grid, 256>>>(out, iters, clk, random_data);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            kern<<<grid, 256>>>(out, iters, clk, random_data);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)grid * 4 * iters * 4 * 32768.0;
            printf("mfma peak (%s operands): %d WG/CU x 4 waves, %d iters x 4 acc: %.1f us, %.0f TFLOP/s; clock64 %llu ticks per %llu x10ns -> %.0f MHz; %.1f clock64 ticks per MFMA per wave\n",
                   random_data ? "random" : "constant", wg_per_cu, iters, ms * 1e3, flop / (ms * 1e-3) / 1e12, h[0], h[1], h[0] / (h[1] * 0.01),
                   (double)h[0] / (iters * 4.0));
        }
    return 0;
}

// TRACE_SMALL=1: through the shape router (small images take the 4-row, one-tile-per-workgroup instantiation)
static int launch_traced(const ConvArgs &a, int cin_units, hipStream_t st) {
    static const bool routed = getenv("TRACE_SMALL") != nullptr;
    return routed ? launch_conv3x3(a, SR355_DTYPE_BF16, cin_units, 0, st) : launch_conv3x3_ring(a, cin_units, 0, st);
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "mfma")) return mfma_peak();
    const int cin_units = argc > 1 ? atoi(argv[1]) : 24;
    const int cout = argc > 2 ? atoi(argv[2]) : 64;
    const int nres = argc > 3 ? atoi(argv[3]) : 1;
    const int N = argc > 4 ? atoi(argv[4]) : 16, H = argc > 5 ? atoi(argv[5]) : 128, W = argc > 6 ? atoi(argv[6]) : 128;
    const int units = 24;
    const size_t act_bytes = (size_t)N * H * units * W * 16;
    const size_t f32_bytes = (size_t)N * H * W * cout * 4;
    const int cout_pad = (cout + 31) / 32 * 32;
    const size_t w_bytes = (size_t)cin_units * 9 * cout_pad * 16;
    unsigned char *in, *out, *w;
    float *bias, *t1, *t2;
    unsigned long long *trace;
    const int max_wg = 4096;
    CK(hipMalloc(&in, act_bytes));
    CK(hipMalloc(&out, act_bytes));
    CK(hipMalloc(&w, w_bytes));
    CK(hipMalloc(&bias, cout_pad * 4));
    CK(hipMalloc(&t1, f32_bytes));
    CK(hipMalloc(&t2, f32_bytes));
    CK(hipMalloc(&trace, 2 * max_wg * 16 * 8));
    CK(hipMemset(in, 0x3c, act_bytes));
    CK(hipMemset(w, 0x3b, w_bytes));
    if (getenv("TRACE_RANDOM")) {   // random bf16 operands in (-2, 2): the data toggling of a real forward
        std::vector<unsigned short> hb(act_bytes / 2), hw(w_bytes / 2);
        unsigned long long sd = 88172645463325252ull;
        auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return sd; };
        for (auto &v : hb) { const unsigned long long r = rnd(); v = (unsigned short)(((r & 1) << 15) | (0x3f00 + ((r >> 1) & 0xff)) | (((r >> 9) & 1) << 7)); }
        for (auto &v : hw) { const unsigned long long r = rnd(); v = (unsigned short)(((r & 1) << 15) | (0x3c00 + ((r >> 1) & 0x1ff))); }
        CK(hipMemcpy(in, hb.data(), act_bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), w_bytes, hipMemcpyHostToDevice));
    }
    CK(hipMemset(bias, 0, cout_pad * 4));
    CK(hipMemset(t1, 0, f32_bytes));
    CK(hipMemset(t2, 0, f32_bytes));

    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in, a.w = w, a.bias = bias;
    a.out_act = out;
    a.res1_w = a.res2_w = 1.f, a.res1_scale = a.res2_scale = 0.2f;
    if (nres < 0) a.res_act = in, a.res_act_units = units;      // the skip read from the bf16 activations
    if (nres >= 1) a.res1 = t1, a.out_f32 = t1;
    if (nres >= 2) a.res2 = t2, a.out_f32 = t2;
    a.mask_slope = 1.f, a.slope = cout == 64 ? 1.f : 0.2f;
    a.N = N, a.H = H, a.W = W, a.in_units = units, a.ups = 0;
    a.out_units = units, a.out_unit_off = cout == 64 ? 0 : cin_units;
    a.cout_real = cout, a.cout_pad = cout_pad, a.f32_units = cout / 4;

    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < 20; ++i)
        if (launch_traced(a, cin_units, st)) return 2;
    CK(hipStreamSynchronize(st));
    CK(hipMemset(trace, 0, 2 * max_wg * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // a few untraced launches just before, so the traced one starts behind a busy queue as in a forward
    for (int i = 0; i < 3; ++i)
        if (launch_traced(a, cin_units, st)) return 2;
    g_conv_trace = trace;
    CK(hipEventRecord(e0, st));
    if (launch_traced(a, cin_units, st)) return 2;
    CK(hipEventRecord(e1, st));
    g_conv_trace = nullptr;
    if (launch_traced(a, cin_units, st)) return 2;
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(2 * max_wg * 16);
    CK(hipMemcpy(t.data(), trace, 2 * max_wg * 16 * 8, hipMemcpyDeviceToHost));
    int nwg = 0;
    unsigned long long t0 = ~0ull, tend = 0;
    for (int b = 0; b < max_wg; ++b)
        if (t[b * 16]) {
            nwg = b + 1;
            t0 = std::min(t0, t[b * 16]);
            tend = std::max(tend, t[b * 16 + 15]);
        }
    printf("cin_units %d cout %d res %d  N %d H %d W %d: event time %.1f us, %d workgroups, first start -> last end %.2f us\n",
           cin_units, cout, nres, N, H, W, ms * 1e3, nwg, (tend - t0) * 0.01);
    {   // shader clock under load: ticks (slot 14) over the wall time between the first and the last stamp
        std::vector<double> mhz;
        for (int b = 0; b < nwg; ++b)
            if (t[b * 16 + 14] && t[b * 16 + 15] > t[b * 16]) {
                mhz.push_back((double)t[b * 16 + 14] / ((t[b * 16 + 15] - t[b * 16]) * 0.01));
                t[b * 16 + 14] = 0;
            }
        if (!mhz.empty()) {
            std::sort(mhz.begin(), mhz.end());
            printf("  shader clock during the kernel: median %.0f MHz (p10 %.0f, p90 %.0f)\n", mhz[mhz.size() / 2],
                   mhz[mhz.size() / 10], mhz[mhz.size() * 9 / 10]);
        }
    }
    const char *names[16] = {"start", "prologue issued", "stage 0 landed", "tile0 mma done", "tile0 epilogue done",
                             "tile1 mma done", "tile1 epilogue done", "tile2 mma done", "tile2 epi done", "t3 mma",
                             "t3 epi", "t4 mma", "args read, before tile set-up", "tile set-up done", "(clock)", "end"};
    for (int k = 0; k < 16; ++k) {
        std::vector<double> v;
        for (int b = 0; b < nwg; ++b)
            if (t[b * 16 + k]) v.push_back((t[b * 16 + k] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-22s n=%4zu  min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", names[k], v.size(), v[0],
               v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
    // in-stage stamps (shader-clock ticks) of wave 0 in steady state: medians of the deltas over the workgroups
    const char *sn[8] = {"loop top", "vmcnt wait", "barrier", "col-0 frags landed", "col-0 mfma + col-1 reads + dma issue",
                         "col-1 mfma + col-2 reads", "col-2 mfma", "cursor advance"};
    for (int k = 1; k < 8; ++k) {
        std::vector<double> v;
        for (int b = 0; b < nwg; ++b) {
            const unsigned long long *s = &t[(size_t)(max_wg + b) * 16];
            if (s[0] && s[k] >= s[k - 1]) v.push_back((double)(s[k] - s[k - 1]));
        }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  stage: %-40s p10 %6.0f  median %6.0f  p90 %6.0f ticks\n", sn[k], v[v.size() / 10], v[v.size() / 2],
               v[v.size() * 9 / 10]);
    }
    return 0;
}
