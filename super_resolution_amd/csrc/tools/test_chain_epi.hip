// Stand-alone check of the chain kernel's epilogue variants against the shared tile epilogue (timing/diagnosis tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#define CHAIN_EPI_TEST 1
#include "../conv3x3_chain.hip"
using namespace sr355;

template <int MODE>
__global__ __launch_bounds__(256) void epi_kernel(ConvArgs a, const float *accs, const float *bias, int variant) {
    __shared__ __attribute__((aligned(16))) unsigned char bias_lds[256];
    if (threadIdx.x < 32) ((float *)bias_lds)[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    f32x16 acc[1][4];
    for (int nb = 0; nb < 4; ++nb)
        for (int e = 0; e < 16; ++e) acc[0][nb][e] = accs[((wave * 4 + nb) * 64 + lane) * 16 + e];
    if (variant == 0) conv_epilogue_tile<1, 4, 16>(a, acc, bias_lds, 0, wave * 4, r, 0, h, true);
    else if (MODE == 0) chain_epilogue_as<1, false, false, false>(a, acc, bias_lds, 0, wave * 4, r, 0, h);
    else if (MODE == 1) chain_epilogue_as<1, true, true, false>(a, acc, bias_lds, 0, wave * 4, r, 0, h);
    else chain_epilogue_as<2, true, true, false>(a, acc, bias_lds, 0, wave * 4, r, 0, h);
}

int main() {
    const int H = 16, W = 32, units = 24;
    const size_t act_bytes = (size_t)H * W * units * 16, f32_bytes = (size_t)H * W * 64 * 4;
    unsigned char *cur, *nxt[2];
    float *accs, *bias, *R[2], *F;
    hipMalloc(&cur, act_bytes); hipMalloc(&nxt[0], act_bytes); hipMalloc(&nxt[1], act_bytes);
    hipMalloc(&accs, 16 * 64 * 16 * 4); hipMalloc(&bias, 256); hipMalloc(&R[0], f32_bytes); hipMalloc(&R[1], f32_bytes); hipMalloc(&F, f32_bytes);
    std::vector<float> ha(16 * 64 * 16), hb(64), hf(f32_bytes / 4);
    std::vector<unsigned short> hc(act_bytes / 2);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto &v : ha) v = rnd(); for (auto &v : hb) v = rnd(); for (auto &v : hf) v = rnd();
    for (auto &v : hc) { float f = rnd(); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    hipMemcpy(accs, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(cur, hc.data(), act_bytes, hipMemcpyHostToDevice); hipMemcpy(F, hf.data(), f32_bytes, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {          // 0: bf16 skip only, 1: bf16 skip + fp32 skip + carrier, 2: fp32 skip + carrier
        for (int v = 0; v < 2; ++v) {
            hipMemset(nxt[v], 0, act_bytes);
            hipMemcpy(R[v], hf.data(), f32_bytes, hipMemcpyHostToDevice);
            ConvArgs a; memset(&a, 0, sizeof(a));
            a.res1_w = a.res2_w = 1.f; a.slope = 1.f; a.mask_slope = 1.f;
            a.out_act = nxt[v]; a.out_units = units; a.N = 1; a.H = H; a.W = W; a.cout_real = a.cout_pad = 32; a.f32_units = 16;
            a.res1_scale = 0.2f;
            if (mode < 2) a.res_act = cur, a.res_act_units = units; else a.res1 = F;
            if (mode >= 1) a.res2 = R[v], a.res2_scale = 0.2f, a.out_f32 = R[v];
            if (mode == 0) hipLaunchKernelGGL(epi_kernel<0>, dim3(1), dim3(256), 0, 0, a, accs, bias, v);
            else if (mode == 1) hipLaunchKernelGGL(epi_kernel<1>, dim3(1), dim3(256), 0, 0, a, accs, bias, v);
            else hipLaunchKernelGGL(epi_kernel<2>, dim3(1), dim3(256), 0, 0, a, accs, bias, v);
        }
        hipDeviceSynchronize();
        std::vector<unsigned short> o0(act_bytes / 2), o1(act_bytes / 2);
        std::vector<float> r0(f32_bytes / 4), r1(f32_bytes / 4);
        hipMemcpy(o0.data(), nxt[0], act_bytes, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), nxt[1], act_bytes, hipMemcpyDeviceToHost);
        hipMemcpy(r0.data(), R[0], f32_bytes, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), R[1], f32_bytes, hipMemcpyDeviceToHost);
        size_t bad = 0, badf = 0, first_bad = (size_t)-1;
        for (size_t i = 0; i < o0.size(); ++i) if (o0[i] != o1[i]) { if (!bad) first_bad = i; ++bad;
            if (bad <= 24) { size_t t = i / 8; printf("   bad c=%zu x=%zu u=%zu y=%zu: %04x vs %04x\n", i % 8, t % 32, (t / 32) % 24, t / 32 / 24, o0[i], o1[i]); } }
        for (size_t i = 0; i < r0.size(); ++i) if (memcmp(&r0[i], &r1[i], 4)) ++badf;
        printf("mode %d: bf16 mismatches %zu of %zu (first at %zu: %04x vs %04x), fp32 mismatches %zu\n", mode, bad, o0.size(), first_bad,
               first_bad != (size_t)-1 ? o0[first_bad] : 0, first_bad != (size_t)-1 ? o1[first_bad] : 0, badf);
    }
    return 0;
}
