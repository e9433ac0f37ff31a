// conv3x3 forward, persistent LDS-DMA ring variant (bf16 path, large images).
//
// Same arithmetic, operand maps and epilogue as conv3x3_mfma (conv3x3.hip); what changes is
// how bytes reach the LDS and how the matrix pipe is kept fed:
//   * persistent workgroups walk a list of (WAVES*NB)x32 tiles, XCD-contiguous, and the loads of the
//     next tile's first stages are already in flight while a tile is being finished;
//   * stages (16 input channels: haloed input tile + all taps of the weights) are written straight
//     into LDS by `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write; padding pixels read
//     beyond the buffer descriptor's range and arrive as zeros);
//   * a ring of R stage slots keeps R-1 stages in flight behind a counted `s_waitcnt vmcnt(N)` and
//     one raw `s_barrier` per stage;
//   * the MFMA operands of tap column kx+1 are read from LDS while the MFMAs of column kx issue
//     (explicit register double buffering of the fragments).
// Replaces nn.Conv2d + cat + LeakyReLU + skips of /root/reference/esrgan/models/esrgan_model.py:18-24.
#include <stdlib.h>

#include "conv_epilogue.h"
#include <atomic>

#include "sr355_internal.h"

namespace sr355 {

// Timeline instrumentation (tools/trace_conv): with -DSR355_TRACE every workgroup stamps the 100 MHz wall
// clock at fixed points of its life into 16 slots of the buffer this pointer names.
unsigned long long *g_conv_trace = nullptr;
#ifdef SR355_TRACE
#define SR355_STAMP(K)                                                                                \
    if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (K)] = wall_clock64();
#else
#define SR355_STAMP(K)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

namespace {
constexpr int TWP = 34;          // tile width 32 + halo
constexpr int RSU = 2;           // units per stage
constexpr int OOB = 0x7fffff00;  // byte offset beyond any descriptor range -> DMA writes zeros

template <int MB, int NB, int WAVES, int R>
struct RingGeo {
    static constexpr int CO = MB * 32;
    static constexpr int TH = WAVES * NB;                 // tile height
    static constexpr int NPIX = (TH + 2) * TWP;           // haloed pixels
    static constexpr int NIN = RSU * NPIX;                // input items (16 B each) per stage
    static constexpr int NIN_I = (NIN + 63) / 64;         // wave-instructions for the input image
    static constexpr int NWT = RSU * 9 * CO;              // weight items per stage (multiple of 64)
    static constexpr int NWT_I = NWT / 64;
    static constexpr int JTOT = NIN_I + NWT_I;            // wave-instructions per stage
    static constexpr int KMAX = (JTOT + WAVES - 1) / WAVES;
    static constexpr int KMIN = JTOT / WAVES;
    static constexpr int W_OFF = NIN_I * 1024;            // weights follow the padded input image
    static constexpr int STAGE_BYTES = JTOT * 1024;
    static constexpr int BIAS_OFF = R * STAGE_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + 256;
    static constexpr int THREADS = WAVES * 64;
    static_assert(NWT % 64 == 0, "weight image must be whole wave-instructions");
    static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");
    static_assert(CO <= 64, "bias image is one 256-byte DMA");
};
}  // namespace

template <int MB, int NB, int WAVES, int R, int WPS>
__global__ __launch_bounds__(WAVES * 64, WPS) void conv3x3_ring(const ConvArgs a) {
    typedef RingGeo<MB, NB, WAVES, R> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SR355_STAMP(0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int H = a.H, W = a.W;
    const int Win = (W + a.zins) >> a.ups;   // (zins: the zero-inserted x2 image of an (H+1)/2 x (W+1)/2 input)
    const int img_bytes = ((H + a.zins) >> a.ups) * a.in_units * Win * 16;
    const int in_stage_stride = RSU * Win * 16;
    const int w_stage_stride = RSU * 9 * a.cout_pad * 16;
    const int S = a.n_stages, n_real = a.n_real_stages;
    const int cb = blockIdx.y;

    // Tile list of this workgroup.  Blocks b and b+8 share an XCD; remap so that the blocks of one
    // XCD take consecutive tiles of every round (neighbouring tiles share halo rows in that L2).
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;
    const int nwg = gridDim.x;
    int first;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        first = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int n_my = first < n_tiles ? (n_tiles - first + nwg - 1) / nwg : 0;
    const int total_g = n_my * S;

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(a.w) + (size_t)cb * G::CO * 16, 0, w_stage_stride * S, 0x00020000);
    // bias -> LDS by one 4-byte-per-lane DMA (oldest operation of wave 0, so every later wait covers it)
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.bias) + cb * G::CO, 0, G::CO * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void *)(smem + G::BIAS_OFF), 4, lane * 4, 0, 0, 0);
    }

    // ---- issue side: per-lane source offsets of this wave's DMA instructions (j = k*WAVES + wave) ----
    int voff[G::KMAX];
#pragma unroll
    for (int k = 0; k < G::KMAX; ++k) {
        const int j = k * WAVES + wave;
        voff[k] = OOB;
        if (j >= G::NIN_I && j < G::JTOT) {   // weight items are tile invariant
            const int jw = (j - G::NIN_I) * 64 + lane;
            const int u = jw / (9 * G::CO);
            const int rem = jw - u * 9 * G::CO;
            const int tap = rem / G::CO;
            const int co = rem - tap * G::CO;
            voff[k] = ((u * 9 + tap) * a.cout_pad + co) * 16;
        }
    }
    __amdgpu_buffer_rsrc_t in_rsrc = w_rsrc;
    int it_tile = 0, it_stage = 0;

#define SR355_TILE_SETUP(ORD)                                                                         \
    {                                                                                                 \
        const int t_ = first + (ORD) * nwg;                                                           \
        const int txi_ = t_ % a.tiles_x;                                                              \
        const int tr_ = t_ / a.tiles_x;                                                               \
        const int tyi_ = tr_ % a.tiles_y;                                                             \
        const int n_ = tr_ / a.tiles_y;                                                               \
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(a.in) + (size_t)n_ * img_bytes, \
                                                    0, img_bytes, 0x00020000);                        \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) {                                         \
            const int j = k * WAVES + wave;                                                           \
            if (j < G::NIN_I) {                                                                       \
                const int i = j * 64 + lane;                                                          \
                const int u = i / G::NPIX;                                                            \
                const int p = i - u * G::NPIX;                                                        \
                const int ty = p / TWP;                                                               \
                const int tx = p - ty * TWP;                                                          \
                const int y = tyi_ * G::TH - 1 + ty, x = txi_ * 32 - 1 + tx;                          \
                const bool valid = (i < G::NIN) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W) && \
                                   !SR355_DBG(a, 1) && !(a.zins && ((y | x) & 1));                   \
                voff[k] = valid ? (((y >> a.ups) * a.in_units + u) * Win + (x >> a.ups)) * 16 : OOB;  \
            }                                                                                         \
        }                                                                                             \
    }

    // One stage of DMA into ring slot SLOT.  Past the end of the tile list the same instructions are
    // issued with out-of-range sources (zeros into a free slot) so the vmcnt arithmetic stays uniform.
#define SR355_ISSUE_STAGE(SLOT)                                                                       \
    {                                                                                                 \
        const bool live_ = it_tile < n_my;                                                            \
        const int si_ = it_stage >= n_real ? it_stage - n_real : it_stage;                            \
        const int soff_in_ = si_ * in_stage_stride;                                                   \
        const int soff_w_ = it_stage * w_stage_stride;                                                \
        unsigned char *dst_ = smem + (SLOT) * G::STAGE_BYTES + wave * 1024;                           \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) {                                         \
            const int j = k * WAVES + wave;                                                           \
            lds_void *l_ = (lds_void *)(dst_ + k * WAVES * 1024);                                     \
            const int vo_ = live_ ? voff[k] : OOB;                                                    \
            if (k * WAVES + WAVES - 1 < G::NIN_I) {                                                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, l_, 16, vo_, soff_in_, 0, 0);       \
            } else if (k * WAVES >= G::NIN_I && k * WAVES + WAVES - 1 < G::JTOT) {                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, l_, 16, vo_, soff_w_, 0, 0);         \
            } else if (j < G::JTOT) {                                                                 \
                if (j < G::NIN_I)                                                                     \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, l_, 16, vo_, soff_in_, 0, 0);   \
                else                                                                                  \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, l_, 16, vo_, soff_w_, 0, 0);     \
            }                                                                                         \
        }                                                                                             \
        if (live_ && ++it_stage == S) {                                                               \
            it_stage = 0;                                                                             \
            if (++it_tile < n_my) SR355_TILE_SETUP(it_tile);                                          \
        }                                                                                             \
    }

    if (n_my > 0) SR355_TILE_SETUP(0);
#pragma unroll
    for (int g = 0; g < R - 1; ++g) SR355_ISSUE_STAGE(g);
    SR355_STAMP(1);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;

    const int lane_in = (h * G::NPIX + wave * NB * TWP + r) * 16;
    const int lane_w = G::W_OFF + (h * 9 * G::CO + r) * 16;

#define SR355_LOAD_FRAGS(WF, XF, KX)                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                              \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                         \
                WF[ky][mb] = *reinterpret_cast<const bf16x8 *>(pw + ((ky * 3 + (KX)) * G::CO + mb * 32) * 16); \
        _Pragma("unroll") for (int rho = 0; rho < NB + 2; ++rho)                                      \
            XF[rho] = *reinterpret_cast<const bf16x8 *>(pin + (rho * TWP + (KX)) * 16);               \
    }
#define SR355_MMA_FRAGS(WF, XF)                                                                       \
    {                                                                                                 \
        _Pragma("unroll") for (int rho = 0; rho < NB + 2; ++rho)                                      \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                        \
                const int nb = rho - ky;                                                              \
                if (nb >= 0 && nb < NB) {                                                             \
                    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                 \
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ky][mb], XF[rho], acc[mb][nb], 0, 0, 0); \
                }                                                                                     \
            }                                                                                         \
    }

    int slot = 0, issue_slot = R - 1, ct_stage = 0, ct_tile = 0;
    for (int g = 0; g < total_g; ++g) {
        // stage g has landed for this wave when at most (R-2) younger stages are outstanding;
        // after the barrier it has landed for every wave, and slot (g-1)%R is free to refill.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * G::KMIN) : "memory");
        __builtin_amdgcn_s_barrier();
        if (g == 0) SR355_STAMP(2);
        const unsigned char *buf = smem + slot * G::STAGE_BYTES;
        const unsigned char *pin = buf + lane_in;
        const unsigned char *pw = buf + lane_w;
        bf16x8 wfA[3][MB], xfA[NB + 2], wfB[3][MB], xfB[NB + 2];
        SR355_LOAD_FRAGS(wfA, xfA, 0);
        SR355_ISSUE_STAGE(issue_slot);
        issue_slot = issue_slot + 1 == R ? 0 : issue_slot + 1;
        __builtin_amdgcn_sched_barrier(0);
        SR355_LOAD_FRAGS(wfB, xfB, 1);
        __builtin_amdgcn_sched_barrier(0);
        SR355_MMA_FRAGS(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        SR355_LOAD_FRAGS(wfA, xfA, 2);
        __builtin_amdgcn_sched_barrier(0);
        SR355_MMA_FRAGS(wfB, xfB);
        __builtin_amdgcn_sched_barrier(0);
        SR355_MMA_FRAGS(wfA, xfA);
        slot = slot + 1 == R ? 0 : slot + 1;

        if (++ct_stage == S) {
            // ---- epilogue of tile ct_tile ----
            if (ct_tile < 6) SR355_STAMP(3 + 2 * ct_tile);
            const int t = first + ct_tile * nwg;
            const int tx_i = t % a.tiles_x;
            const int tr = t / a.tiles_x;
            const int ty_i = tr % a.tiles_y;
            const int n = tr / a.tiles_y;
            const int x = tx_i * 32 + r;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int y = ty_i * G::TH + wave * NB + nb;
                const bool ok = (x < W) && (y < H) && !SR355_DBG(a, 2);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = mb * 32 + q * 8 + h * 4;
                        const int c0 = cb * G::CO + cl;
                        const float4 bv = *reinterpret_cast<const float4 *>(smem + G::BIAS_OFF + cl * 4);
                        float v[4] = {acc[mb][nb][4 * q + 0] + bv.x, acc[mb][nb][4 * q + 1] + bv.y,
                                      acc[mb][nb][4 * q + 2] + bv.z, acc[mb][nb][4 * q + 3] + bv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mb][nb][4 * q + e] = 0.f;
                        if (!ok || c0 >= a.cout_real) continue;
                        conv_epilogue_quad<__bf16>(a, v, n, y, x, c0, h);
                    }
                }
            }
            ct_stage = 0;
            if (ct_tile < 6) SR355_STAMP(4 + 2 * ct_tile);
            ++ct_tile;
        }
    }
    // drain the DMA still in flight (dummy stages) before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SR355_STAMP(15);
#undef SR355_TILE_SETUP
#undef SR355_ISSUE_STAGE
#undef SR355_LOAD_FRAGS
#undef SR355_MMA_FRAGS
}

// WPS = waves per SIMD the register budget is sized for (workgroups per CU = WPS * 4 / WAVES).
template <int MB, int NB, int WAVES, int R, int WPS = 2>
static int launch_ring_inst(const ConvArgs &a0, hipStream_t stream) {
    typedef RingGeo<MB, NB, WAVES, R> G;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    auto kern = conv3x3_ring<MB, NB, WAVES, R, WPS>;
    static std::atomic<bool> attr_done[64];         // (zero-initialised; acquire / release: threads may race here)
    static std::atomic<int> n_cu[64];
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES),
                      "hipFuncSetAttribute"))
            return SR355_ERR_HIP;
        int cu = 0;
        if (check_hip(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev),
                      "hipDeviceGetAttribute"))
            return SR355_ERR_HIP;
        n_cu[dev].store(cu > 0 ? cu : 256, std::memory_order_relaxed);
        attr_done[dev].store(true, std::memory_order_release);
    }
    const long long n_tiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (n_tiles <= 0 || n_tiles > 0x7fffffffLL) {
        set_error("conv3x3 ring: %lld tiles is out of range", n_tiles);
        return SR355_ERR_INVALID;
    }
    int wg_per_cu = (160 * 1024) / G::LDS_BYTES;
    const int by_waves = WPS * 4 / WAVES;   // register budget
    if (wg_per_cu > by_waves) wg_per_cu = by_waves;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const long long slots = (long long)n_cu[dev].load(std::memory_order_relaxed) * wg_per_cu;
    dim3 grid((unsigned)(n_tiles < slots ? n_tiles : slots), a.cout_pad / G::CO, 1);
    hipLaunchKernelGGL(kern, grid, dim3(G::THREADS), G::LDS_BYTES, stream, a);
    return check_hip(hipGetLastError(), "conv3x3 ring launch");
}

// Ring kernel entry: bf16 only, cin_units even, cout_pad 32 or 64.
int launch_conv3x3_ring(const ConvArgs &a, int cin_units, int weight_split, hipStream_t stream) {
    ConvArgs b = a;
    b.n_real_stages = cin_units / 2;
    b.n_stages = weight_split ? 2 * b.n_real_stages : b.n_real_stages;
#ifdef SR355_TRACE
    static const int dbg = [] { const char *e = getenv("SR355_DEBUG"); return e ? atoi(e) : 0; }();
#else
    const int dbg = 0;
#endif
    static const int var = [] { const char *e = getenv("SR355_VARIANT"); return e ? atoi(e) : 9; }();
    b.debug = dbg;
    b.trace = g_conv_trace;
    const bool wide = (a.cout_pad % 64) == 0;
    // 9 (default): the branch-free, MFMA-interleaved stage body of conv3x3_fine.hip; 5: this file's phase-ordered
    // body (two workgroups per CU; one stage in flight for the 64-channel convs, two for the 32-channel ones)
    if (var != 5 || a.f16_in) return launch_conv3x3_fine(b, stream);
    return wide ? launch_ring_inst<2, 2, 4, 2>(b, stream) : launch_ring_inst<1, 2, 4, 3>(b, stream);
}

}  // namespace sr355
