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
typedef __attribute__((ext_vector_type(16))) float epi_f32x16;
typedef unsigned int epi_v2u __attribute__((ext_vector_type(2)));
typedef unsigned int epi_v4u __attribute__((ext_vector_type(4)));
typedef float epi_v4f __attribute__((ext_vector_type(4)));

// 16-byte buffer store at byte offset `voffset + uoffset` (per-lane + wave-uniform part).  The uniform part is added to
// the vector offset and the instruction's scalar offset stays the immediate 0 on purpose: a store of more than 64 bits
// keeps reading its data registers after it has issued, and the two wait states the ISA asks for before a VALU
// instruction may overwrite them are inserted by hipcc (ROCm 7.2) only for that form.  With the uniform part in an SGPR
// scalar offset the compiler's hazard recogniser exempts the store, the register allocator reuses the first data
// register in the very next instruction, and gfx950 then stores a few lanes of the NEW value: the discriminator's
// 64 -> 128 conv came out with wrong first channels of some quads about once in forty launches on a cold memory system
// (tools/diag_layers.py found the launch, tools/check_store_hazard.py finds the instruction pairs in the code object).
// An out-of-range `voffset` (0x7fffff00) stays out of range: the sum is below 2^32 and above any buffer's size.
template <int AUX = 0>
__device__ __forceinline__ void epi_store_b128(epi_v4u data, __amdgpu_buffer_rsrc_t rsrc, int voffset, int uoffset) {
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset + uoffset, 0, AUX);
}

// w * r + s * v, written once so that every epilogue path rounds it the same way
__device__ __forceinline__ float epi_axpby(float w, float r, float s, float v) { return __builtin_fmaf(w, r, s * v); }

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
            for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res1_w, (float)rv[j], a.res1_scale, v[j]);
        }
    }
    if (a.res1) {
        const float4 rv = reinterpret_cast<const float4 *>(a.res1)[fidx];
        v[0] = epi_axpby(a.res1_w, rv.x, a.res1_scale, v[0]);
        v[1] = epi_axpby(a.res1_w, rv.y, a.res1_scale, v[1]);
        v[2] = epi_axpby(a.res1_w, rv.z, a.res1_scale, v[2]);
        v[3] = epi_axpby(a.res1_w, rv.w, a.res1_scale, v[3]);
    }
    if (a.res2) {
        const float4 rv = reinterpret_cast<const float4 *>(a.res2)[fidx];
        v[0] = epi_axpby(a.res2_w, rv.x, a.res2_scale, v[0]);
        v[1] = epi_axpby(a.res2_w, rv.y, a.res2_scale, v[1]);
        v[2] = epi_axpby(a.res2_w, rv.z, a.res2_scale, v[2]);
        v[3] = epi_axpby(a.res2_w, rv.w, a.res2_scale, v[3]);
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

// ---- whole-tile epilogue of the ring kernels (bf16 path) ------------------------------------------------------
// The per-quad form above recomputes 64-bit addresses for every quad and keeps each residual load next to the store
// that depends on it, so a wave walks through its 16 quads one memory round trip at a time.  For the common shape
// (no NCHW / uint8 output, whole 32-channel blocks) this form addresses through buffer descriptors of the
// tile's image — one 32-bit lane offset per output row, the unit step in the scalar offset — issues the residual
// loads of four quads together, and drops lanes outside the image by an out-of-range offset instead of a branch.
// Same arithmetic, in the same order, as conv_epilogue_quad.
__device__ __forceinline__ bool conv_epilogue_tile_ok(const ConvArgs &a) {
    return !a.out_nchw && !a.out_u8 && (a.cout_real & 31) == 0;
}

// The commonest shape of all — bias, LeakyReLU, one blocked 16-bit output, nothing else (every growth conv, the
// up-sampling convs, most of the discriminator and the VGG stack) — without the per-quad tests of the general form
// below: a fifth of its instructions, which matters where a tile has few stages (64 input channels: the epilogue of
// the general form is as long as half the tile's MFMA stream).  Same arithmetic as the general form: for a slope in
// [0, 1], max(v, slope * v) is v > 0 ? v : slope * v.
__device__ __forceinline__ bool conv_epilogue_plain_ok(const ConvArgs &a) {
    return conv_epilogue_tile_ok(a) && a.out_act && !a.out_f32 && !a.mask_act && !a.res_act && !a.res1 && !a.res2 &&
           a.slope >= 0.f && a.slope <= 1.f;
}
// ... the same with the sign gate of a data gradient in front of the store (MASK), and with one fp32 output in place
// of the 16-bit one (F32: a conv in front of a BatchNorm)
__device__ __forceinline__ bool conv_epilogue_plain_mask_ok(const ConvArgs &a) {
    return conv_epilogue_tile_ok(a) && a.out_act && !a.out_f32 && a.mask_act && !a.res_act && !a.res1 && !a.res2 &&
           !a.f16_out && a.slope >= 0.f && a.slope <= 1.f;
}
__device__ __forceinline__ bool conv_epilogue_plain_f32_ok(const ConvArgs &a) {
    return conv_epilogue_tile_ok(a) && !a.out_act && a.out_f32 && !a.mask_act && !a.res_act && !a.res1 && !a.res2 &&
           a.slope >= 0.f && a.slope <= 1.f;
}

template <int MB, int NB, bool F16OUT, bool MASK = false, bool F32 = false>
__device__ __forceinline__ void conv_epilogue_tile_plain(const ConvArgs &a, epi_f32x16 (&acc)[MB][NB],
                                                         const unsigned char *bias_lds, int n, int y0, int x, int cb_co,
                                                         int h, bool live) {
    static_assert(!(F16OUT && (MASK || F32)) && !(MASK && F32), "one variant at a time");
    constexpr int OOB = 0x7fffff00;
    const int H = a.H, W = a.W;
    const int wstep = __builtin_amdgcn_readfirstlane(W * 16);          // bytes from one unit of a row to the next
    auto rsrc = [&](const void *base, size_t img_bytes) {
        const unsigned long long p = (unsigned long long)base + (size_t)n * img_bytes;
        return __builtin_amdgcn_make_buffer_rsrc(
            (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p)),
            0, __builtin_amdgcn_readfirstlane((int)img_bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_out = F32 ? rsrc(a.out_f32, (size_t)H * W * a.f32_units * 16)
                                              : rsrc(a.out_act, (size_t)H * W * a.out_units * 16);
    const __amdgpu_buffer_rsrc_t rs_mk = rsrc(MASK ? a.mask_act : nullptr, (size_t)H * W * a.mask_units * 16);
    const float slope = a.slope;
    [[maybe_unused]] const float mask_slope = a.mask_slope;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int y = y0 + nb;
            const bool ok = live && x < W && y < H;
            const int vo_act = ok ? ((y * a.out_units + a.out_unit_off + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
            const int vo_f = ok ? ((y * a.f32_units + (cb_co >> 2) + h) * W + x) * 16 : OOB;
            [[maybe_unused]] epi_v2u mk[4];
            if constexpr (MASK) {
                const int vo_mk = ok ? ((y * a.mask_units + a.mask_unit_off + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mk[q] = __builtin_amdgcn_raw_buffer_load_b64(rs_mk, vo_mk, (mb * 4 + q) * wstep, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + (mb * 32 + q * 8 + h * 4) * 4);
                float v[4] = {acc[mb][nb][4 * q + 0] + bv.x, acc[mb][nb][4 * q + 1] + bv.y,
                              acc[mb][nb][4 * q + 2] + bv.z, acc[mb][nb][4 * q + 3] + bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mb][nb][4 * q + e] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], v[j] * slope);
                if constexpr (MASK) {
                    const epi_bf16x4 mv = __builtin_bit_cast(epi_bf16x4, mk[q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)mv[j] > 0.f ? v[j] : v[j] * mask_slope;
                }
                if constexpr (F32) {
                    const epi_v4f of = {v[0], v[1], v[2], v[3]};
                    epi_store_b128(__builtin_bit_cast(epi_v4u, of), rs_out, vo_f, (mb * 8 + q * 2) * wstep);
                } else {
                    epi_v2u o;
                    if constexpr (F16OUT) {
                        epi_f16x4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (_Float16)fminf(fmaxf(v[j], -65504.f), 65504.f);
                        o = __builtin_bit_cast(epi_v2u, pk);
                    } else {
                        epi_bf16x4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (__bf16)v[j];
                        o = __builtin_bit_cast(epi_v2u, pk);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, vo_act, (mb * 4 + q) * wstep, 0);
                }
            }
        }
    }
}

// The last conv of a generator (at most 4 output channels padded to a block of 32, fp32 NCHW or uint8 HWC output,
// no residuals): only the first quad of the lanes h = 0 holds real channels.  Same arithmetic as conv_epilogue_quad.
__device__ __forceinline__ bool conv_epilogue_rgb_ok(const ConvArgs &a) {
    return a.cout_real <= 4 && (a.out_nchw || a.out_u8) && !a.out_act && !a.out_f32 && !a.mask_act && !a.res_act &&
           !a.res1 && !a.res2;
}

template <int NB>
__device__ __forceinline__ void conv_epilogue_rgb(const ConvArgs &a, epi_f32x16 (&acc)[1][NB], const unsigned char *bias_lds,
                                                  int n, int y0, int x, int h, bool live) {
    const int H = a.H, W = a.W, C = a.cout_real;
    const float slope = a.slope;
    const float4 bv = *reinterpret_cast<const float4 *>(bias_lds);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int y = y0 + nb;
        float v[4] = {acc[0][nb][0] + bv.x, acc[0][nb][1] + bv.y, acc[0][nb][2] + bv.z, acc[0][nb][3] + bv.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][nb][e] = 0.f;
        if (!(live && h == 0 && x < W && y < H)) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
        const size_t row = (size_t)n * H + y;
        if (a.out_nchw) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < C) a.out_nchw[(((size_t)n * C + j) * H + y) * W + x] = v[j];
        }
        if (a.out_u8) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < C) {
                    const float cl = v[j] < 0.f ? 0.f : (v[j] > 1.f ? 1.f : v[j]);
                    a.out_u8[((row * W) + x) * C + j] = (unsigned char)(int)(cl * 255.0f);
                }
        }
    }
}

// AUX is the cache-policy operand of the stores: 0 = plain, 16 = write-through (sc1), which is what a kernel whose
// workgroups read each other's output inside one launch needs.
template <int MB, int NB, int AUX = 0>
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs &a, epi_f32x16 (&acc)[MB][NB],
                                                   const unsigned char *bias_lds, int n, int y0, int x, int cb_co,
                                                   int h, bool live) {
    constexpr int OOB = 0x7fffff00;
    const int H = a.H, W = a.W;
    const int wstep = __builtin_amdgcn_readfirstlane(W * 16);          // bytes from one unit of a row to the next
    auto rsrc = [&](const void *base, size_t img_bytes) {
        const unsigned long long p = (unsigned long long)base + (size_t)n * img_bytes;
        return __builtin_amdgcn_make_buffer_rsrc(
            (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p)),
            0, __builtin_amdgcn_readfirstlane((int)img_bytes), 0x00020000);
    };
    const size_t f32_img = (size_t)H * W * a.f32_units * 16;
    const __amdgpu_buffer_rsrc_t rs_act = rsrc(a.out_act, (size_t)H * W * a.out_units * 16);
    const __amdgpu_buffer_rsrc_t rs_ra = rsrc(a.res_act, (size_t)H * W * a.res_act_units * 16);
    const __amdgpu_buffer_rsrc_t rs_mk = rsrc(a.mask_act, (size_t)H * W * a.mask_units * 16);
    const __amdgpu_buffer_rsrc_t rs_r1 = rsrc(a.res1, f32_img);
    const __amdgpu_buffer_rsrc_t rs_r2 = rsrc(a.res2, f32_img);
    const __amdgpu_buffer_rsrc_t rs_f32 = rsrc(a.out_f32, f32_img);
    const float slope = a.slope, mask_slope = a.mask_slope;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int y = y0 + nb;
        const bool ok = live && x < W && y < H;
        const int vo_act = ok ? ((y * a.out_units + a.out_unit_off + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
        const int vo_ra = ok ? ((y * a.res_act_units + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
        const int vo_mk = ok ? ((y * a.mask_units + a.mask_unit_off + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
        const int vo_f = ok ? ((y * a.f32_units + (cb_co >> 2) + h) * W + x) * 16 : OOB;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            epi_v2u ra[4], mk[4];
            epi_v4u r1[4], r2[4];
            if (a.mask_act) {   // sign gate of a data gradient: the saved activation at the output position
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    mk[q] = __builtin_amdgcn_raw_buffer_load_b64(rs_mk, vo_mk, (mb * 4 + q) * wstep, 0);
            }
            if (a.res_act) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ra[q] = __builtin_amdgcn_raw_buffer_load_b64(rs_ra, vo_ra, (mb * 4 + q) * wstep, 0);
            }
            if (a.res1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r1[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, vo_f, (mb * 8 + q * 2) * wstep, 0);
            }
            if (a.res2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r2[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, vo_f, (mb * 8 + q * 2) * wstep, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = mb * 32 + q * 8 + h * 4;
                const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + cl * 4);
                float v[4] = {acc[mb][nb][4 * q + 0] + bv.x, acc[mb][nb][4 * q + 1] + bv.y,
                              acc[mb][nb][4 * q + 2] + bv.z, acc[mb][nb][4 * q + 3] + bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mb][nb][4 * q + e] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
                if (a.mask_act) {
                    const epi_bf16x4 mv = __builtin_bit_cast(epi_bf16x4, mk[q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)mv[j] > 0.f ? v[j] : v[j] * mask_slope;
                }
                if (a.res_act) {
                    const epi_bf16x4 rv = __builtin_bit_cast(epi_bf16x4, ra[q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res1_w, (float)rv[j], a.res1_scale, v[j]);
                }
                if (a.res1) {   // (the whole vector is re-typed: a bit_cast of one indexed element reads element 0)
                    const epi_v4f rf = __builtin_bit_cast(epi_v4f, r1[q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res1_w, rf[j], a.res1_scale, v[j]);
                }
                if (a.res2) {
                    const epi_v4f rf = __builtin_bit_cast(epi_v4f, r2[q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res2_w, rf[j], a.res2_scale, v[j]);
                }
                if (a.out_f32) {
                    const epi_v4f of = {v[0], v[1], v[2], v[3]};
                    epi_store_b128<AUX>(__builtin_bit_cast(epi_v4u, of), rs_f32, vo_f, (mb * 8 + q * 2) * wstep);
                }
                if (a.out_act) {
                    epi_v2u o;
                    if (a.f16_out) {
                        epi_f16x4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (_Float16)fminf(fmaxf(v[j], -65504.f), 65504.f);
                        o = __builtin_bit_cast(epi_v2u, pk);
                    } else {
                        epi_bf16x4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (__bf16)v[j];
                        o = __builtin_bit_cast(epi_v2u, pk);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs_act, vo_act, (mb * 4 + q) * wstep, AUX);
                }
            }
        }
    }
}

}  // namespace sr355
