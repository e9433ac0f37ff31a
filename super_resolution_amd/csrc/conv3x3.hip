// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces nn.Conv2d(cin, cout, 3, 1, 1) of /root/reference/esrgan/models/esrgan_model.py:10-14,
// :44-52 together with what surrounds it in the forward pass: the torch.cat that feeds it
// (:20-23, input = units [0, cin_units) of one blocked buffer), the in-place LeakyReLU (:15),
// the scaled skips (:24, :39, :58) and the nearest x2 upsample in front of it (:59-60).
//
// GEMM view:  D[cout][pixel] = sum_{tap, cin} Wt[cout][tap, cin] * X[tap, cin][pixel]
//   A operand = weights (rows = output channels), B operand = pixels (columns), so that the
//   32x32 accumulator has the pixel on the lane and 4 consecutive output channels in each
//   register quad: stores into the blocked layout are 512 contiguous bytes per instruction.
//
// Workgroup = 256 threads = 4 waves.  Output tile = (4*NB) rows x 32 pixels x (32*MB) channels;
// wave w owns rows [w*NB, (w+1)*NB).  Input channels stream through LDS in stages of SU units
// (bf16: 8 channels per 16-byte unit, fp32: 4), double buffered: the global loads of stage
// s+1 are in flight while stage s is multiplied.  LDS images are [unit][pixel][16 B] for the
// haloed input tile and [unit][tap][cout][16 B] for the weights, so every fragment read is a
// conflict-free ds_read_b128 of 32 consecutive 16-byte slots per half-wave.
// For one kx the (NB+2) input rows are read once and reused by the three ky taps.
//
// Split weights (bf16 only): the packed weights may carry a second image holding the bf16
// rounding residue of every weight (w = hi + lo).  The kernel then runs n_stages = 2 * n_real
// stages; stage s >= n_real re-reads input stage s - n_real against the residue weights, which
// removes the weight-rounding error at twice the MFMA count.  Used for the four HR-side convs
// and conv_first, whose weight rounding dominates the bf16 output error.
#include <stdlib.h>

#include "conv_epilogue.h"
#include <atomic>

#include "sr355_internal.h"

namespace sr355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <typename T> struct Frag;
template <> struct Frag<__bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<_Float16> { typedef f16x8 type; };

__device__ __forceinline__ void mma(f32x16 &c, const bf16x8 &a, const bf16x8 &b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16 &c, const f16x8 &a, const f16x8 &b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// One 16-byte unit pair (lane half h holds unit 2s+h) = 8 fp32 channels = four K=2 steps; step t
// multiplies element t of both operands, i.e. channel 4*(2s+h)+t on lane half h.
__device__ __forceinline__ void mma(f32x16 &c, const f32x4 &a, const f32x4 &b) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
}

constexpr int TWP = 34;  // tile width 32 + 1 halo pixel each side

// ST = 2: stride-2 convolution (nn.Conv2d(c, c, 3, 2, 1) of the discriminator,
// /root/reference/esrgan/models/esrgan_model.py:67-84): the input tile of a TH x 32 output tile is (2 TH + 1) x 65
// pixels (pitch 66) and output pixel r of a row reads input pixels 2 r + kx (16-byte slots 32 bytes apart: two-way bank
// conflicts on the fragment reads, no wasted MFMA).
template <typename T, int MB, int NB, int SU, int ST = 1>
struct Geo {
    static constexpr int TH = 4 * NB;
    static constexpr int PITCH = ST == 1 ? TWP : 66;
    static constexpr int NPIX = (ST * TH + (ST == 1 ? 2 : 1)) * PITCH;
    static constexpr int CO = MB * 32;
    static constexpr int NIN = SU * NPIX;          // 16-byte items of the input image per stage
    static constexpr int KIN = (NIN + 255) / 256;
    static constexpr int NWT = SU * 9 * CO;        // 16-byte items of the weight image per stage
    static constexpr int KW = (NWT + 255) / 256;
    static constexpr int IN_BYTES = NIN * 16;
    static constexpr int BUF_BYTES = IN_BYTES + NWT * 16;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

template <typename T, int MB, int NB, int SU, int ST = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma(const ConvArgs a) {
    typedef Geo<T, MB, NB, SU, ST> G;
    typedef typename Frag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
    // tiles so neighbouring tiles' halo rows hit the same L2.  Bijective for any grid size.
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int tile_x = bid % a.tiles_x;
    const int trest = bid / a.tiles_x;
    const int tile_y = trest % a.tiles_y;
    const int n = trest / a.tiles_y;
    const int x0 = tile_x * 32, y0 = tile_y * G::TH;
    const int cb = blockIdx.y;  // block of CO output channels

    const int H = a.H, W = a.W;
    const int Hin = ST == 2 ? a.in_H : (H + a.zins) >> a.ups, Win = ST == 2 ? a.in_W : (W + a.zins) >> a.ups;
    const unsigned char *in_img = a.in + (size_t)n * Hin * a.in_units * Win * 16;
    const int in_stage_stride = SU * Win * 16;
    const unsigned char *w_base = a.w + (size_t)cb * G::CO * 16;
    const int w_stage_stride = SU * 9 * a.cout_pad * 16;

    // Per-thread staging descriptors (stage invariant).
    int goff[G::KIN];
#pragma unroll
    for (int k = 0; k < G::KIN; ++k) {
        const int i = tid + k * 256;
        const int u = i / G::NPIX;
        const int p = i - u * G::NPIX;
        const int ty = p / G::PITCH;
        const int tx = p - ty * G::PITCH;
        const int y = ST * y0 - 1 + ty, x = ST * x0 - 1 + tx;
        bool valid;
        if constexpr (ST == 2)
            valid = (i < G::NIN) && ((unsigned)y < (unsigned)Hin) && ((unsigned)x < (unsigned)Win);
        else
            valid = (i < G::NIN) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W) &&
                    !(a.zins && ((y | x) & 1));
        goff[k] = (valid && !SR355_DBG(a, 1)) ? (((y >> a.ups) * a.in_units + u) * Win + (x >> a.ups)) * 16 : -1;
    }
    int woff[G::KW];
#pragma unroll
    for (int k = 0; k < G::KW; ++k) {
        const int j = tid + k * 256;
        const int u = j / (9 * G::CO);
        const int rem = j - u * 9 * G::CO;
        const int tap = rem / G::CO;
        const int co = rem - tap * G::CO;
        woff[k] = (j < G::NWT && !SR355_DBG(a, 1)) ? ((u * 9 + tap) * a.cout_pad + co) * 16 : 0;
    }

    u32x4 sin[G::KIN], sw[G::KW];
#define SR355_STAGE_LOAD(S)                                                                      \
    {                                                                                            \
        const int si_ = (S) >= n_real ? (S) - n_real : (S); /* split weights: input stages wrap */ \
        const unsigned char *ip_ = in_img + (size_t)si_ * in_stage_stride;                       \
        const unsigned char *wp_ = w_base + (size_t)(S) * w_stage_stride;                        \
        _Pragma("unroll") for (int k = 0; k < G::KIN; ++k) {                                     \
            const u32x4 *src_ = reinterpret_cast<const u32x4 *>(ip_ + (goff[k] >= 0 ? goff[k] : 0)); \
            u32x4 v_ = *src_;                                                                    \
            if (goff[k] < 0) v_ = u32x4{0u, 0u, 0u, 0u};                                        \
            sin[k] = v_;                                                                         \
        }                                                                                        \
        _Pragma("unroll") for (int k = 0; k < G::KW; ++k)                                        \
            sw[k] = *reinterpret_cast<const u32x4 *>(wp_ + woff[k]);                             \
    }
#define SR355_STAGE_STORE(BUF)                                                                   \
    {                                                                                            \
        unsigned char *b_ = smem + (BUF) * G::BUF_BYTES;                                         \
        _Pragma("unroll") for (int k = 0; k < G::KIN; ++k) {                                     \
            if (tid + k * 256 < G::NIN) *reinterpret_cast<u32x4 *>(b_ + (tid + k * 256) * 16) = sin[k]; \
        }                                                                                        \
        _Pragma("unroll") for (int k = 0; k < G::KW; ++k) {                                      \
            if (tid + k * 256 < G::NWT)                                                          \
                *reinterpret_cast<u32x4 *>(b_ + G::IN_BYTES + (tid + k * 256) * 16) = sw[k];     \
        }                                                                                        \
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;

    const int n_real = a.n_real_stages;
    const int lane_in = (h * G::NPIX + ST * wave * NB * G::PITCH + ST * r) * 16;
    const int lane_w = G::IN_BYTES + (h * 9 * G::CO + r) * 16;

    SR355_STAGE_LOAD(0);
    SR355_STAGE_STORE(0);
    __syncthreads();

    const int n_stages = a.n_stages;
    for (int s = 0; s < n_stages; ++s) {
        const bool more = (s + 1 < n_stages);
        if (more) SR355_STAGE_LOAD(s + 1);

        const unsigned char *buf = smem + (s & 1) * G::BUF_BYTES;
        const unsigned char *pin = buf + lane_in;
        const unsigned char *pw = buf + lane_w;
#pragma unroll
        for (int s2 = 0; s2 < SU / 2; ++s2) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                frag_t wf[3][MB];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        wf[ky][mb] = *reinterpret_cast<const frag_t *>(
                            pw + ((2 * s2 * 9 + ky * 3 + kx) * G::CO + mb * 32) * 16);
                if constexpr (ST == 1) {
#pragma unroll
                    for (int rho = 0; rho < NB + 2; ++rho) {
                        const frag_t xf = *reinterpret_cast<const frag_t *>(
                            pin + (2 * s2 * G::NPIX + rho * TWP + kx) * 16);
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int nb = rho - ky;
                            if (nb >= 0 && nb < NB) {
#pragma unroll
                                for (int mb = 0; mb < MB; ++mb) mma(acc[mb][nb], wf[ky][mb], xf);
                            }
                        }
                    }
                } else {   // stride 2: output row nb reads input rows 2 nb + ky, nothing shared between taps
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const frag_t xf = *reinterpret_cast<const frag_t *>(
                                pin + (2 * s2 * G::NPIX + (2 * nb + ky) * G::PITCH + kx) * 16);
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) mma(acc[mb][nb], wf[ky][mb], xf);
                        }
                }
            }
        }

        if (more) SR355_STAGE_STORE((s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: bias, LeakyReLU, mask, scaled skips, stores ----
    const int x = x0 + r;
    if (x >= W || SR355_DBG(a, 2)) return;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int y = y0 + wave * NB + nb;
        if (y >= H) continue;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = cb * G::CO + mb * 32 + q * 8 + h * 4;
                if (c0 >= a.cout_real) continue;
                const float4 bv = *reinterpret_cast<const float4 *>(a.bias + c0);
                float v[4] = {acc[mb][nb][4 * q + 0] + bv.x, acc[mb][nb][4 * q + 1] + bv.y,
                              acc[mb][nb][4 * q + 2] + bv.z, acc[mb][nb][4 * q + 3] + bv.w};
                conv_epilogue_quad<T>(a, v, n, y, x, c0, h);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
template <typename T, int MB, int NB, int SU, int ST = 1>
static int launch_inst(const ConvArgs &a0, hipStream_t stream) {
    typedef Geo<T, MB, NB, SU, ST> G;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    auto kern = conv3x3_mfma<T, MB, NB, SU, ST>;
    static std::atomic<bool> attr_done[64];         // (zero-initialised; acquire / release: threads may race here)
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES),
                      "hipFuncSetAttribute"))
            return SR355_ERR_HIP;
        attr_done[dev].store(true, std::memory_order_release);
    }
    const long long nblk = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        set_error("conv3x3: grid of %lld tiles is out of range", nblk);
        return SR355_ERR_INVALID;
    }
    dim3 grid((unsigned)nblk, a.cout_pad / G::CO, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, stream, a);
    return check_hip(hipGetLastError(), "conv3x3 launch");
}

template <typename T, int MB, int SU>
static int launch_nb(const ConvArgs &a, hipStream_t stream) {
    // Tallest tile that still gives every CU two workgroups; short images get short tiles.
    const long long tx = (a.W + 31) / 32;
    const long long cblk = a.cout_pad / (32 * MB);
    auto blocks = [&](int th) { return (long long)a.N * tx * ((a.H + th - 1) / th) * cblk; };
    if (blocks(16) >= 512) return launch_inst<T, MB, 4, SU>(a, stream);
    if (blocks(8) >= 512) return launch_inst<T, MB, 2, SU>(a, stream);
    return launch_inst<T, MB, 1, SU>(a, stream);
}

// bf16 at small tile counts: with 4-row tiles a stage is 9*MB MFMAs per wave against one global-load latency, so the
// K loop is latency-bound; stages of four units (32 channels) halve the number of those latencies.
template <int MB, typename T16 = __bf16>
static int launch_bf16(const ConvArgs &a, int cin_units, int weight_split, hipStream_t stream) {
    const long long tx = (a.W + 31) / 32;
    const long long cblk = a.cout_pad / (32 * MB);
    static const int su4 = [] { const char *e = getenv("SR355_SMALL_SU4"); return e ? atoi(e) : 1; }();
    if (su4 && (cin_units % 4) == 0 && (long long)a.N * tx * ((a.H + 7) / 8) * cblk < 512) {
        ConvArgs b = a;
        b.n_real_stages = cin_units / 4;
        b.n_stages = weight_split ? 2 * b.n_real_stages : b.n_real_stages;
        return launch_inst<T16, MB, 1, 4>(b, stream);
    }
    return launch_nb<T16, MB, 2>(a, stream);
}

int launch_conv3x3(const ConvArgs &a, int dtype, int cin_units, int weight_split, hipStream_t stream) {
    if (cin_units <= 0 || (cin_units & 1)) {
        set_error("conv3x3: cin_units=%d must be a positive even number", cin_units);
        return SR355_ERR_INVALID;
    }
    ConvArgs b = a;
    if (weight_split && dtype != SR355_DTYPE_BF16) {
        set_error("conv3x3: split weights exist only for the bf16 path");
        return SR355_ERR_INVALID;
    }
    b.n_real_stages = cin_units / 2;
    b.n_stages = weight_split ? 2 * b.n_real_stages : b.n_real_stages;
    b.debug = 0;
    if (a.stride == 2) {
        // stride 2 (bf16, unsplit): the register-staged kernel on 4-row output tiles, 64 or 32 outputs per workgroup
        if (dtype != SR355_DTYPE_BF16 || weight_split || a.ups || a.f16_in) {
            set_error("conv3x3: stride 2 runs on plain bf16 inputs only");
            return SR355_ERR_UNSUPPORTED;
        }
        return (a.cout_pad % 64) == 0 ? launch_inst<__bf16, 2, 1, 2, 2>(b, stream) : launch_inst<__bf16, 1, 1, 2, 2>(b, stream);
    }
    if (dtype == SR355_DTYPE_BF16 && (a.cout_pad == 32 || a.cout_pad == 64)) {
        // Large images go to the persistent LDS-DMA ring kernel (one workgroup per CU needs at least
        // a CU's worth of 16x32 tiles); SR355_RING=0/1 forces the choice for A/B measurements.
        static const int force = [] { const char *e = getenv("SR355_RING"); return e ? atoi(e) : -1; }();
        const long long tiles = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 15) / 16);
        if (force == 1 || (force != 0 && tiles >= 192)) return launch_conv3x3_ring(a, cin_units, weight_split, stream);
    }
    if (dtype == SR355_DTYPE_BF16 && (a.cout_pad == 32 || a.cout_pad == 64)) {
        // Launches that cannot fill the chip (training patches): the LDS-DMA kernel on 4-row tiles, one per
        // workgroup, instead of the register-staged kernel below (pixel iteration 11.4 -> 10.4 ms);
        // SR355_SMALL_FINE=0 keeps the register-staged kernel for A/B measurements.
        static const int small_fine = [] { const char *e = getenv("SR355_SMALL_FINE"); return e ? atoi(e) : 1; }();
        const long long blocks8 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 7) / 8) * (a.cout_pad / 32);
        if (small_fine && blocks8 < 512) return launch_conv3x3_fine_small(b, stream);
    }
    if (dtype == SR355_DTYPE_BF16 && a.cout_pad > 64 && (a.cout_pad % 64) == 0 && !a.f16_in) {
        // More than 64 output channels (discriminator, VGG19, sub-pixel convs): the LDS-DMA kernel with one grid row
        // per 64-channel block, at any image size (adversarial iteration 23.5 -> 22.5 ms); SR355_WIDE_FINE=0 keeps
        // the register-staged kernel for A/B measurements.
        static const int wide_fine = [] { const char *e = getenv("SR355_WIDE_FINE"); return e ? atoi(e) : 1; }();
        if (wide_fine) return launch_conv3x3_ring(a, cin_units, weight_split, stream);
    }
    bool wide = (a.cout_pad % 64) == 0;
    if (wide && dtype == SR355_DTYPE_BF16) {
        // too few tiles to give every CU a workgroup: one workgroup per 32 output channels instead of 64
        static const int split_knob = [] { const char *e = getenv("SR355_SMALL_SPLIT"); return e ? atoi(e) : 1; }();
        const long long blocks4 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 3) / 4) * (a.cout_pad / 64);
        if (split_knob && blocks4 < 256) wide = false;
    }
    if (dtype == SR355_DTYPE_BF16 && a.f16_in)      // the same 16-byte units holding IEEE fp16
        return wide ? launch_bf16<2, _Float16>(b, cin_units, weight_split, stream)
                    : launch_bf16<1, _Float16>(b, cin_units, weight_split, stream);
    if (dtype == SR355_DTYPE_BF16)
        return wide ? launch_bf16<2>(b, cin_units, weight_split, stream) : launch_bf16<1>(b, cin_units, weight_split, stream);
    if (dtype == SR355_DTYPE_F32)
        return wide ? launch_nb<float, 2, 2>(b, stream) : launch_nb<float, 1, 2>(b, stream);
    set_error("conv3x3: unknown dtype %d", dtype);
    return SR355_ERR_INVALID;
}

}  // namespace sr355
