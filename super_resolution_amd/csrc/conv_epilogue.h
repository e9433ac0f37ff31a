// Fused epilogue shared by the conv3x3 kernels: one call handles 4 consecutive output channels
// [c0, c0+4) of one pixel, held by one lane after the 32x32 MFMA accumulation.
//
// Forward use (esrgan_model.py:15-24, :39, :58): bias, LeakyReLU, scaled skips.
// Backward use (the autograd of the same lines): the "mask" multiplies the data gradient by the
// LeakyReLU derivative taken from the sign of the forward activation, and the weighted residuals
// add the gradient that arrives over the skip connections.
#pragma once
#include "sr355_internal.h"

namespace sr355 {

typedef __attribute__((ext_vector_type(4))) __bf16 epi_bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 epi_f16x4;

template <typename T>
__device__ __forceinline__ void conv_epilogue_quad(const ConvArgs &a, float (&v)[4], int n, int y, int x, int c0,
                                                   int h) {
    const int H = a.H, W = a.W;
    const size_t row = (size_t)n * H + y;
    const float slope = a.slope;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
    if (a.mask_act) {
        float m[4];
        if constexpr (sizeof(T) == 2) {
            const size_t o = ((row * a.mask_units + a.mask_unit_off + (c0 >> 3)) * W + x) * 16 + h * 8;
            const epi_bf16x4 mv = *reinterpret_cast<const epi_bf16x4 *>(a.mask_act + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = (float)mv[j];
        } else {
            const size_t o = ((row * a.mask_units + a.mask_unit_off + (c0 >> 2)) * W + x) * 16;
            const float4 mv = *reinterpret_cast<const float4 *>(a.mask_act + o);
            m[0] = mv.x, m[1] = mv.y, m[2] = mv.z, m[3] = mv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : v[j] * a.mask_slope;
    }
    const size_t fidx = (row * a.f32_units + (c0 >> 2)) * W + x;   // in float4 units
    if constexpr (sizeof(T) == 2) {
        if (a.res_act) {   // the skip read from the bf16 activations themselves (no fp32 carrier for this skip)
            const size_t o = ((row * a.res_act_units + (c0 >> 3)) * W + x) * 16 + h * 8;
            const epi_bf16x4 rv = *reinterpret_cast<const epi_bf16x4 *>(a.res_act + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a.res1_w * (float)rv[j] + a.res1_scale * v[j];
        }
    }
    if (a.res1) {
        const float4 rv = reinterpret_cast<const float4 *>(a.res1)[fidx];
        v[0] = a.res1_w * rv.x + a.res1_scale * v[0];
        v[1] = a.res1_w * rv.y + a.res1_scale * v[1];
        v[2] = a.res1_w * rv.z + a.res1_scale * v[2];
        v[3] = a.res1_w * rv.w + a.res1_scale * v[3];
    }
    if (a.res2) {
        const float4 rv = reinterpret_cast<const float4 *>(a.res2)[fidx];
        v[0] = a.res2_w * rv.x + a.res2_scale * v[0];
        v[1] = a.res2_w * rv.y + a.res2_scale * v[1];
        v[2] = a.res2_w * rv.z + a.res2_scale * v[2];
        v[3] = a.res2_w * rv.w + a.res2_scale * v[3];
    }
    if (a.out_f32) reinterpret_cast<float4 *>(a.out_f32)[fidx] = make_float4(v[0], v[1], v[2], v[3]);
    if (a.out_act) {
        if constexpr (sizeof(T) == 2) {
            const size_t o = ((row * a.out_units + a.out_unit_off + (c0 >> 3)) * W + x) * 16 + h * 8;
            if (a.f16_out) {
                // fp16 ends at 65504: saturate instead of storing an infinity the next conv would turn into NaN
                epi_f16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (_Float16)fminf(fmaxf(v[j], -65504.f), 65504.f);
                *reinterpret_cast<epi_f16x4 *>(a.out_act + o) = pk;
            } else {
                epi_bf16x4 pk;
                pk[0] = (__bf16)v[0];
                pk[1] = (__bf16)v[1];
                pk[2] = (__bf16)v[2];
                pk[3] = (__bf16)v[3];
                *reinterpret_cast<epi_bf16x4 *>(a.out_act + o) = pk;
            }
        } else {
            const size_t o = ((row * a.out_units + a.out_unit_off + (c0 >> 2)) * W + x) * 16;
            *reinterpret_cast<float4 *>(a.out_act + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (a.out_nchw) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < a.cout_real) a.out_nchw[(((size_t)n * a.cout_real + c0 + j) * H + y) * W + x] = v[j];
    }
    if (a.out_u8) {
        // what the reference's inference does to the output (/root/reference/esrgan/test.py:72-73): clamp(0, 1),
        // then ToPILImage's mul(255).byte() (truncation), stored as the HWC image
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < a.cout_real) {
                const float cl = v[j] < 0.f ? 0.f : (v[j] > 1.f ? 1.f : v[j]);
                a.out_u8[((row * W) + x) * a.cout_real + c0 + j] = (unsigned char)(int)(cl * 255.0f);
            }
    }
}

}  // namespace sr355
