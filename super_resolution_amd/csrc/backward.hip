// Backward-pass kernels of the 3x3 convolution (autograd of nn.Conv2d in
// /root/reference/esrgan/models/esrgan_model.py:10-14 as driven by loss.backward() in
// /root/reference/esrgan/train.py:161, :220).
//
// * data gradient: the forward MFMA kernels run it unchanged on TRANSPOSED packed weights
//   (pack_weight_t below: input/output channels swapped, taps mirrored); a dense block's five data
//   gradients read a growing stack of gradient channels exactly as its forward convs read a growing
//   stack of activations, so no read-modify-write accumulation is needed;
// * weight gradient: dW[co][ci][tap] = sum_pixels dY[co][p] * X[ci][p + tap] contracts over pixels,
//   so both MFMA operands must hold 8 consecutive pixels of one channel per lane.  The LDS images
//   stay pixel-major ([unit][pixel][8 channels], filled by LDS-DMA like the forward) and the
//   fragments are fetched with the transposing read ds_read_b64_tr_b16;
// * partial weight gradients of different pixel ranges go to per-workgroup slabs (plain stores) and
//   a second kernel sums them in a fixed order: deterministic, no float atomics;
// * 2x2 sum pooling: the adjoint of the nearest x2 upsample fused in front of upconv1/upconv2.
#include <string.h>

#include <mutex>

#include <atomic>

#include "sr355_internal.h"

namespace sr355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 cvt<__bf16>(float v) { return (__bf16)v; }

// ------------------------------------------------------------------------------------------
// Transposed packing.  Forward weight w[Co][Ci][3][3]; the data-gradient conv takes the Co forward
// outputs as its input channels and produces the forward input channels [ci_lo, ci_lo + n_out):
//   wt[c'][o'][tap'] = scale * w[c'][ci_lo + o'][8 - tap'].
// The destination is a packed tensor [units][9][cout_pad][unit]; this call fills units
// [unit_off, unit_off + ceil(Co / unit)).
template <typename T>
__device__ __forceinline__ void pack_weight_t_body(const float *__restrict__ w, T *__restrict__ wp, int Co, int Ci,
                                                   int ci_lo, int n_out, int unit_off, int cout_pad, float scale) {
    constexpr int CPU = 16 / sizeof(T);
    const int units = (Co + CPU - 1) / CPU;
    const int total = units * 9 * cout_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        // consecutive threads take consecutive (output, tap) pairs of one unit: their reads of w are consecutive
        // floats (the 16-byte stores are scattered instead; there are eight reads to one store)
        const int tap = 8 - i % 9;
        const int t = i / 9;
        const int o = t % cout_pad;
        const int u = t / cout_pad;
        T v[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) {
            const int c = u * CPU + e;
            float f = 0.f;
            if (c < Co && o < n_out) f = scale * w[((size_t)c * Ci + ci_lo + o) * 9 + (8 - tap)];
            v[e] = cvt<T>(f);
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(wp)[((size_t)(unit_off + u) * 9 + tap) * cout_pad + o] = pk;
    }
}

template <typename T>
__global__ void pack_weight_t_kernel(const float *__restrict__ w, T *__restrict__ wp, int Co, int Ci, int ci_lo,
                                     int n_out, int unit_off, int cout_pad, float scale) {
    pack_weight_t_body<T>(w, wp, Co, Ci, ci_lo, n_out, unit_off, cout_pad, scale);
}

struct PackTJobTable {
    PackTJob j[PACK_BATCH];
};

// blockIdx.y = job; the blocks of a row stride over that job's items
template <typename T>
__global__ void pack_weight_t_batch_kernel(const PackTJobTable t) {
    const PackTJob &j = t.j[blockIdx.y];
    pack_weight_t_body<T>(j.w, (T *)j.wp, j.Co, j.Ci, j.ci_lo, j.n_out, j.unit_off, j.cout_pad, j.scale);
}

// the jobs in device memory: any number of them in one launch (blockIdx.y = job).  Eight blocks a job: a dense block's
// transposed packs are 2 k - 14 k items each, and with 32 blocks a job most of the grid's 11 k blocks had nothing to do
// (the whole generator: 107 -> 58 us).
constexpr int PACKT_BLOCKS = 8;
template <typename T>
__global__ void pack_weight_t_table_kernel(const PackTJob *__restrict__ jobs) {
    const PackTJob j = jobs[blockIdx.y];
    pack_weight_t_body<T>(j.w, (T *)j.wp, j.Co, j.Ci, j.ci_lo, j.n_out, j.unit_off, j.cout_pad, j.scale);
}

// ------------------------------------------------------------------------------------------
// out[n][y][c][x] = sum of the 2x2 block of in at (2y.., 2x..), optionally gated by the sign of a
// forward activation; written as fp32 trunk and/or as blocked activations at a unit offset.
template <typename T>
__global__ void pool2x2_kernel(const T *__restrict__ in, int N, int H, int W, int U, const T *__restrict__ mask,
                               int mask_units, int mask_unit_off, float mask_slope, float *__restrict__ out_f32,
                               T *__restrict__ out_act, int out_units, int out_unit_off) {
    constexpr int CPU = 16 / sizeof(T);
    const size_t total = (size_t)N * H * U * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int u = (int)(t % U);
        t /= U;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float acc[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const size_t src = ((((size_t)n * 2 * H + 2 * y + dy) * U + u) * 2 * W + 2 * x + dx);
                uint4 pk = reinterpret_cast<const uint4 *>(in)[src];
                T v[CPU];
                __builtin_memcpy(v, &pk, 16);
#pragma unroll
                for (int e = 0; e < CPU; ++e) acc[e] += (float)v[e];
            }
        if (mask) {
            const size_t m = ((((size_t)n * H + y) * mask_units + mask_unit_off + u) * W + x);
            uint4 pk = reinterpret_cast<const uint4 *>(mask)[m];
            T v[CPU];
            __builtin_memcpy(v, &pk, 16);
#pragma unroll
            for (int e = 0; e < CPU; ++e) acc[e] = (float)v[e] > 0.f ? acc[e] : acc[e] * mask_slope;
        }
        if (out_f32) {
            constexpr int Q = CPU / 4;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const size_t o = ((((size_t)n * H + y) * (U * Q) + u * Q + q) * W + x);
                reinterpret_cast<float4 *>(out_f32)[o] =
                    make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        if (out_act) {
            T v[CPU];
#pragma unroll
            for (int e = 0; e < CPU; ++e) v[e] = cvt<T>(acc[e]);
            uint4 pk;
            __builtin_memcpy(&pk, v, 16);
            const size_t o = ((((size_t)n * H + y) * out_units + out_unit_off + u) * W + x);
            reinterpret_cast<uint4 *>(out_act)[o] = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient.  Workgroup = 3 waves; wave ky owns the three taps of kernel row ky.  Work item =
// (4 x 32 pixel tile, chunk of 32 input channels); a workgroup sums its items in registers and
// writes one slab [9][CO][32] at the end.
constexpr int WGRAD_MAX_JOBS = 54;   // convs (or 32-output slices of one conv) a batched launch takes: nine dense blocks of six
struct WgradArgs {
    const unsigned char *in;   // blocked bf16 activations (conv input)
    const unsigned char *dy;   // blocked bf16 gradient of the conv output
    float *slabs;              // [chunks][G][9][CO][32]
    float *bias_slabs;         // [G][CO] per-workgroup sums of dy over its pixels, or null
    int N, H, W;               // output-resolution dims
    int in_units, ups;         // input geometry (ups: input at half resolution, nearest x2)
    int dy_units, dy_unit_off;
    int dy_s2;                 // dy is stored at (H+1)/2 x (W+1)/2 and read as its zero-inserted x2 image: the weight
                               // gradient of a stride-2 conv whose input has H x W pixels
    int tiles_x, tiles_y;
    // batched launch (blockIdx.z = conv of a dense block sharing `in` and `dy`): per-conv channel chunks,
    // workgroups per chunk, position of its gradient in dy and of its slabs; n_batch = 0: single conv
    int n_batch;
    int z_chunks[WGRAD_MAX_JOBS], z_groups[WGRAD_MAX_JOBS], z_dy_unit_off[WGRAD_MAX_JOBS];
    int z_co_off[WGRAD_MAX_JOBS], z_co_stride[WGRAD_MAX_JOBS];   // a job's 32 outputs sit at rows [co_off, co_off + 32) of slabs co_stride wide
    long long z_slab_off[WGRAD_MAX_JOBS], z_bias_off[WGRAD_MAX_JOBS];   // in floats from `slabs`; bias < 0: none
    long long z_in_off[WGRAD_MAX_JOBS], z_dy_off[WGRAD_MAX_JOBS];       // in bytes from `in` / `dy`: jobs of several dense blocks
};

#ifdef SR355_EXP_WG_NOLOAD   /* timing experiment (wrong results): every tile load out of range, no memory traffic */
#define SR355_EXP_WG_NOLOAD_V true
#else
#define SR355_EXP_WG_NOLOAD_V false
#endif
namespace {
constexpr int WTH = 4;                     // tile rows
constexpr int WPX = WTH * 32;              // 128 output pixels
constexpr int WIPX = (WTH + 2) * 34;       // 204 haloed input pixels
constexpr int W_OOB = 0x7fffff00;
// In the LDS a unit (8 channels) of the output-gradient tile is DY_PITCH pixels apart, not WPX: 128 pixels x 16 bytes
// is a multiple of the 256-byte bank row, and the four units a transposing read's half-wave touches would then sit on
// the same banks (4-way conflicts: 43 % of this kernel's LDS cycles by SQ_LDS_BANK_CONFLICT); 132 pixels shift each
// unit by 16 banks.  (The input tile's unit pitch, 204 pixels, already does.)
constexpr int DY_PITCH = WPX + 4;
template <int MB>
struct WGeo {
    static constexpr int CO = MB * 32;
    static constexpr int DY_ITEMS = MB * 4 * WPX;          // 512 * MB
    static constexpr int DY_I = DY_ITEMS / 64;             // two DMA instructions per unit
    static constexpr int IN_ITEMS = 4 * WIPX;              // 816
    static constexpr int IN_I = (IN_ITEMS + 63) / 64;      // 13
    static constexpr int JTOT = DY_I + IN_I;
    static constexpr int KMAX = (JTOT + 2) / 3;
    static constexpr int IN_OFF = MB * 4 * DY_PITCH * 16;
    static constexpr int STAGE_BYTES = IN_OFF + IN_I * 1024;
    static constexpr int N_STAGES = MB == 1 ? 3 : 2;       // tiles in LDS: one being multiplied, the others in flight
    static constexpr int LDS_BYTES = N_STAGES * STAGE_BYTES;
    static_assert(WPX == 128, "a unit of the gradient tile is two 64-item DMA instructions");
};
}  // namespace

// Two transposing reads = one MFMA operand: 8 consecutive pixels (rows of the LDS image) of the channel this lane
// owns.  `addr` is this lane's piece address (pixel q, channels 4p..4p+3) for the first 4 pixels.
// Issued through inline asm, and waited for by wg_frags_wait below: a transposing read the compiler can see is ordered
// behind ALL LDS-DMA still in flight (it puts `s_waitcnt vmcnt(0)` in front of the first one), i.e. behind the tiles
// this workgroup has just requested for its NEXT iterations, which turns the ring of tiles into one memory latency per
// tile.  The kernel's own counted wait and barrier at the top of an iteration are what guarantee that the tile being
// read has landed.
struct WgFrag {
    v2u_t lo, hi;
};
__device__ __forceinline__ void wg_frag_issue(WgFrag &f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:64"
                 : "=&v"(f.lo), "=&v"(f.hi)
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ bf16x8 wg_frag_value(const WgFrag &f) {
    const v4u_t raw = {f.lo.x, f.lo.y, f.hi.x, f.hi.y};
    return __builtin_bit_cast(bf16x8, raw);
}
// Waits until at most `PENDING` LDS reads issued after these N fragments are outstanding (LDS reads return in order)
// and ties the fragments' registers to the wait, so that nothing reads them earlier.
template <int PENDING, int N>
__device__ __forceinline__ void wg_frags_wait(WgFrag (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].lo), "+v"(f[i].hi));
}

// WAVE = ky, a template argument: the three waves of a workgroup run three copies of the loop in which every "does this
// wave's k-th DMA instruction belong to the gradient tile, the input tile or neither" is decided at compile time (as a
// run-time value of the wave each was a scalar branch, some twenty per tile).
template <int MB, int WAVE>
__device__ __forceinline__ void wgrad3x3_body(const WgradArgs &a0) {
    typedef WGeo<MB> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int wave = WAVE;   // = ky
    const int chunk = blockIdx.y;
    // view of the arguments for this conv of a batched launch
    struct {
        const unsigned char *in, *dy;
        float *slabs, *bias_slabs;
        int N, H, W, in_units, ups, dy_units, dy_unit_off, tiles_x, tiles_y, dy_s2;
    } a = {a0.in, a0.dy, a0.slabs, a0.bias_slabs, a0.N, a0.H, a0.W, a0.in_units, a0.ups, a0.dy_units, a0.dy_unit_off,
           a0.tiles_x, a0.tiles_y, a0.dy_s2};
    int nwg = gridDim.x;
    int co_off = 0, co_stride = G::CO;
    if (a0.n_batch > 0) {
        const int z = blockIdx.z;
        nwg = a0.z_groups[z];
        if (chunk >= a0.z_chunks[z] || (int)blockIdx.x >= nwg) return;   // whole workgroup, before any barrier
        a.dy_unit_off = a0.z_dy_unit_off[z];
        a.in += a0.z_in_off[z];
        a.dy += a0.z_dy_off[z];
        a.bias_slabs = a0.z_bias_off[z] >= 0 ? a0.slabs + a0.z_bias_off[z] : nullptr;
        a.slabs = a0.slabs + a0.z_slab_off[z];
        co_off = a0.z_co_off[z];
        co_stride = a0.z_co_stride[z];
    }
    const int H = a.H, W = a.W;
    const int Win = W >> a.ups;
    const int in_img_bytes = (H >> a.ups) * a.in_units * Win * 16;
    const int Wd = (W + a.dy_s2) >> a.dy_s2;
    const int dy_img_bytes = ((H + a.dy_s2) >> a.dy_s2) * a.dy_units * Wd * 16;
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;

    // Source offsets of this wave's DMA instructions (item j = k * 3 + wave of a tile: a 16-byte piece of the output
    // gradient [u][py][px] or of the haloed input [u][ty][tx]).  What does not depend on the tile is computed once:
    // the item's position relative to the tile origin (ry, rx; an item that never exists gets a row no image has)
    // and the part of its offset that position contributes (cst).  Per tile there remain one add and two unsigned
    // compares per item, and the tile coordinates advance by the launch's stride without a division (a tile is
    // 24 MFMAs a wave: a set-up of a few hundred instructions per tile would be longer than the tile itself).
    int ry[G::KMAX], rx[G::KMAX], cst[G::KMAX];
#pragma unroll
    for (int k = 0; k < G::KMAX; ++k) {
        const int j = k * 3 + wave;
        ry[k] = 0x40000000, rx[k] = 0, cst[k] = 0;
        if (j < G::DY_I) {
            const int i = j * 64 + lane; /* [u][py][px] */
            const int u = i / WPX;
            const int pp = i - u * WPX;
            const int py = pp >> 5, px = pp & 31;
            if (!(a.dy_s2 && ((py | px) & 1))) ry[k] = py;
            rx[k] = px;
            cst[k] = (((py >> a.dy_s2) * a.dy_units + a.dy_unit_off + u) * Wd + (px >> a.dy_s2)) * 16;
        } else if (j < G::JTOT) {
            const int i = (j - G::DY_I) * 64 + lane; /* [u][ty][tx] haloed */
            const int u = i / WIPX;
            const int pp = i - u * WIPX;
            const int ty = pp / 34;
            const int tx = pp - ty * 34;
            const int uu = chunk * 4 + u;
            if (i < G::IN_ITEMS && uu < a.in_units) ry[k] = ty - 1;
            rx[k] = tx - 1;
            cst[k] = ((((ty - 1) >> a.ups) * a.in_units + uu) * Win + ((tx - 1) >> a.ups)) * 16;
        }
        asm volatile("" : "+v"(ry[k]), "+v"(rx[k]), "+v"(cst[k]));   // (kept in registers, not recomputed per tile)
    }
    // issue cursor: coordinates of the tile set up last, and what one step of nwg tiles adds to them
    int c_tx, c_ty, c_n;
    {
        const int t0 = (int)blockIdx.x;
        c_tx = t0 % a.tiles_x;
        const int tr = t0 / a.tiles_x;
        c_ty = tr % a.tiles_y;
        c_n = tr / a.tiles_y;
    }
    const int d_tx = nwg % a.tiles_x, d_q = nwg / a.tiles_x;
    const int d_ty = d_q % a.tiles_y, d_n = d_q / a.tiles_y;
    __amdgpu_buffer_rsrc_t in_rsrc, dy_rsrc;
    in_rsrc = dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(a.in), 0, 0, 0x00020000);
#define SR355_WG_ADVANCE()                                                                              \
    {                                                                                                   \
        c_tx += d_tx;                                                                                   \
        const int cx_ = c_tx >= a.tiles_x;                                                              \
        c_tx -= cx_ ? a.tiles_x : 0;                                                                    \
        c_ty += d_ty + cx_;                                                                             \
        const int cy_ = c_ty >= a.tiles_y;                                                              \
        c_ty -= cy_ ? a.tiles_y : 0;                                                                    \
        c_n += d_n + cy_;                                                                               \
    }
// Tile-level part of the addresses of the tile at the cursor (scalar values) ...
    int y0_t = 0, x0_t = 0, base_dy_t = 0, base_in_t = 0;
#define SR355_WG_BASES()                                                                                \
    {                                                                                                   \
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(a.in) + (size_t)c_n * in_img_bytes, 0, \
                                                    in_img_bytes, 0x00020000);                          \
        dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(a.dy) + (size_t)c_n * dy_img_bytes, 0, \
                                                    dy_img_bytes, 0x00020000);                          \
        y0_t = c_ty * WTH, x0_t = c_tx * 32;   /* (multiples of 4 and 32: the shifts below split exactly) */ \
        base_dy_t = (((y0_t >> a.dy_s2) * a.dy_units) * Wd + (x0_t >> a.dy_s2)) * 16;                   \
        base_in_t = (((y0_t >> a.ups) * a.in_units) * Win + (x0_t >> a.ups)) * 16;                      \
    }
// ... and this wave's K-th DMA instruction of that tile into slot SLOT (LIVE false: the tile does not exist, the
// instruction goes out with an out-of-range source and fills its piece of a free slot with zeros, so that every tile
// costs every wave the same number of DMA instructions, which is what the counted waits rely on)
#define SR355_WG_PIECE(K, SLOT, LIVE)                                                                   \
    {                                                                                                   \
        const int j_ = (K) * 3 + wave;   /* (a constant once the loops around this are unrolled) */      \
        if (j_ < G::JTOT) {                                                                             \
            const int base_ = j_ < G::DY_I ? base_dy_t : base_in_t;                                     \
            const bool ok_ = (LIVE) && (unsigned)(y0_t + ry[K]) < (unsigned)H &&                        \
                             (unsigned)(x0_t + rx[K]) < (unsigned)W && !SR355_EXP_WG_NOLOAD_V;          \
            const int vo_ = ok_ ? base_ + cst[K] : W_OOB;                                               \
            unsigned char *dst_ = smem + (SLOT) * G::STAGE_BYTES;                                       \
            if (j_ < G::DY_I) {                                                                         \
                lds_void *l = (lds_void *)(dst_ + ((j_ >> 1) * DY_PITCH + (j_ & 1) * 64) * 16);         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rsrc, l, 16, vo_, 0, 0, 0);                 \
            } else {                                                                                    \
                lds_void *l = (lds_void *)(dst_ + G::IN_OFF + (j_ - G::DY_I) * 1024);                   \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, l, 16, vo_, 0, 0, 0);                 \
            }                                                                                           \
        }                                                                                               \
    }
#define SR355_WG_TILE(SLOT, LIVE)                                                                       \
    {                                                                                                   \
        SR355_WG_BASES();                                                                               \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) SR355_WG_PIECE(k, SLOT, LIVE);              \
    }

    f32x16 acc[3][MB];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[kx][mb][e] = 0.f;
    // Bias gradient = dy summed over pixels = dy x ones: wave 0 of the first channel chunk spends one more
    // MFMA per dy fragment on it (every column of the product holds the sum).
    const bool do_bias = a.bias_slabs && chunk == 0 && wave == 0;
    f32x16 accb[MB];
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[mb][e] = 0.f;

    // transposing-read lane geometry: 16-lane group g reads a 4-pixel x 16-channel block; lane 4q+p of
    // the group supplies the address of pixel q, channels 4p..4p+3, and receives channel (lane & 15).
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int ch_half = g & 1, k_half = g >> 1;
    const int unit_in_block = 2 * ch_half + (p >> 1);          // of the 4 units of a 32-channel block
    const int byte_in_unit = (p & 1) * 8;
    const int dy_lane = (unit_in_block * DY_PITCH + 8 * k_half + q) * 16 + byte_in_unit;
    const int in_lane = G::IN_OFF + (unit_in_block * WIPX + wave * 34 + 8 * k_half + q) * 16 + byte_in_unit;

    // Tiles blockIdx.x, + nwg, ...: a ring of N_STAGES slots, AHEAD tiles in flight behind the one being multiplied
    // (a tile is 24 MFMAs per wave against one memory latency, so the depth of the ring is what sets the pace).
    // With 32 outputs every wave issues the same number of DMA instructions per tile (JTOT is a multiple of the 3
    // waves), which is what the counted wait of the three-slot ring relies on.
    static_assert(MB != 1 || G::JTOT % 3 == 0, "counted waits need equal DMA counts per wave");
    const int n_my = (int)blockIdx.x < n_tiles ? (n_tiles - (int)blockIdx.x + nwg - 1) / nwg : 0;
    constexpr int AHEAD = G::N_STAGES - 1;
    // the first AHEAD tiles (requested with out-of-range sources where this workgroup has fewer)
    SR355_WG_TILE(0, n_my > 0);
    if (AHEAD > 1) {
        SR355_WG_ADVANCE();
        SR355_WG_TILE(1, n_my > 1);
    }
    int slot = 0;
    for (int it = 0; it < n_my; ++it) {
        // tile `it` has landed once at most the AHEAD - 1 younger tiles' instructions of this wave are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * G::KMAX) : "memory");
        __builtin_amdgcn_s_barrier();
        // tile it + AHEAD goes into the slot tile it - 1 has just left; its DMA instructions are issued one or two a group
        // below, in the shadow of that group's MFMAs
        const bool live = it + AHEAD < n_my;
        const int slot2 = slot + AHEAD >= G::N_STAGES ? slot + AHEAD - G::N_STAGES : slot + AHEAD;
        SR355_WG_ADVANCE();
        SR355_WG_BASES();
        // eight groups (tile row y, 16-pixel half s) of MB + 3 fragments and 3 MB MFMAs; the fragments of group g + 1 are
        // requested before the MFMAs of group g
        const unsigned buf = (unsigned)(size_t)(smem + slot * G::STAGE_BYTES);
        constexpr int NF = MB + 3, GROUPS = WTH * 2;
        WgFrag fr[2][NF];
#define SR355_WG_FRAGS(SET, GI)                                                                         \
    {                                                                                                   \
        const int y_ = (GI) >> 1, s_ = (GI) & 1;                                                        \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                               \
            wg_frag_issue(fr[SET][mb], buf + dy_lane + (mb * 4 * DY_PITCH + y_ * 32 + 16 * s_) * 16);  \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                \
            wg_frag_issue(fr[SET][MB + kx], buf + in_lane + (y_ * 34 + 16 * s_ + kx) * 16);            \
    }
        SR355_WG_FRAGS(0, 0);
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            const int cur = gi & 1;
            if (gi + 1 < GROUPS) {
                SR355_WG_FRAGS(cur ^ 1, gi + 1);
                wg_frags_wait<2 * NF>(fr[cur]);      // the next group's reads (two instructions a fragment) may be pending
            } else {
                wg_frags_wait<0>(fr[cur]);
            }
#pragma unroll
            for (int k = gi; k < G::KMAX; k += GROUPS) SR355_WG_PIECE(k, slot2, live);
            bf16x8 af[MB], bfr[3];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) af[mb] = wg_frag_value(fr[cur][mb]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) bfr[kx] = wg_frag_value(fr[cur][MB + kx]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[kx][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[kx], acc[kx][mb], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    accb[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], ones, accb[mb], 0, 0, 0);
            }
        }
#undef SR355_WG_FRAGS
        slot = slot + 1 == G::N_STAGES ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // slab [tap][co][ci32]; accumulator element: row (co) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (ci) = lane&31
    float *slab = a.slabs + ((size_t)chunk * nwg + blockIdx.x) * 9 * co_stride * 32 + (size_t)co_off * 32;
    const int col = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rowc = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                slab[((wave * 3 + kx) * co_stride + rowc) * 32 + col] = acc[kx][mb][e];
            }
    if (do_bias && col == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                a.bias_slabs[(size_t)blockIdx.x * co_stride + co_off + mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = accb[mb][e];
    }
}

#undef SR355_WG_BASES
#undef SR355_WG_PIECE
#undef SR355_WG_TILE
#undef SR355_WG_ADVANCE

// Non-template entry points: hipcc (ROCm 7.2) emits no host stub for a kernel TEMPLATE whose body
// issues LDS-DMA loads through two run-time selected descriptors; plain kernels are unaffected.
template <int MB>
__device__ __forceinline__ void wgrad3x3_waves(const WgradArgs &a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0)
        wgrad3x3_body<MB, 0>(a);
    else if (wave == 1)
        wgrad3x3_body<MB, 1>(a);
    else
        wgrad3x3_body<MB, 2>(a);
}
__global__ __launch_bounds__(192) void wgrad3x3_kernel_mb1(const WgradArgs a) { wgrad3x3_waves<1>(a); }
__global__ __launch_bounds__(192) void wgrad3x3_kernel_mb2(const WgradArgs a) { wgrad3x3_waves<2>(a); }

// dW[co][ci][tap] (=|+=) scale * sum_g slab[ci/32][g][tap][co][ci%32], and, when db is given,
// db[co] (=|+=) scale * sum_g bias_slab[g][co]; a fixed summation order, so results are reproducible.
// One workgroup of 256 threads = 64 items x 4 waves; an item is four consecutive input channels of one (tap, output)
// pair (one float4 of a slab row).  Wave k sums the slabs g = k, k + 4, ...; the four partial sums are added in wave
// order through the LDS.  Bias items (one output channel each) follow the weight items.
__device__ __forceinline__ void wgrad_reduce_body(const WgradReduceJob &j, int accumulate) {
    __shared__ float4 part[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cin4 = (j.cin + 3) >> 2;
    const int n_w = 9 * j.cout * cin4;
    const int total = n_w + (j.db ? j.cout : 0);
    for (int base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {   // (uniform per workgroup: barriers inside)
        const int i = base + lane;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        int ci = 0, co = 0, tap = 0;
        const bool is_w = i < n_w, is_b = !is_w && i < total;
        if (is_w) {
            ci = (i % cin4) * 4;
            const int tt = i / cin4;
            co = tt % j.cout;
            tap = tt / j.cout;
            const float *s = j.slabs + ((size_t)(ci >> 5) * j.G * 9 + tap) * j.CO * 32 + co * 32 + (ci & 31);
            const size_t stride = (size_t)9 * j.CO * 32;
            for (int g = wave; g < j.G; g += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(s + g * stride);
                sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
            }
        } else if (is_b) {
            co = i - n_w;
            for (int g = wave; g < j.G; g += 4) sum.x += j.bias_slabs[(size_t)g * j.CO + co];
        }
        if (wave) part[wave - 1][lane] = sum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 v = part[k][lane];
                sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
            }
            if (is_w) {
                const float r[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ci + e >= j.cin) break;
                    float *d = j.dw + ((size_t)co * j.cin + ci + e) * 9 + tap;
                    *d = accumulate ? *d + j.scale * r[e] : j.scale * r[e];
                }
            } else if (is_b) {
                j.db[co] = accumulate ? j.db[co] + j.scale * sum.x : j.scale * sum.x;
            }
        }
        __syncthreads();
    }
}

// dW[co][ci][tap] (=|+=) scale * sum_g slab[ci/32][g][tap][co][ci%32], and, when db is given,
// db[co] (=|+=) scale * sum_g bias_slab[g][co]; a fixed summation order, so results are reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dw,
                                                           int cout, int cin, int CO, int G, float scale,
                                                           int accumulate, const float *__restrict__ bias_slabs,
                                                           float *__restrict__ db) {
    const WgradReduceJob j = {slabs, bias_slabs, dw, db, cout, cin, CO, G, scale};
    wgrad_reduce_body(j, accumulate);
}

// The same reduction for up to WGRAD_MAX_JOBS convs or slices in one launch (blockIdx.y = job).
struct WgradReduceTable {
    WgradReduceJob j[WGRAD_MAX_JOBS];
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradReduceTable t, int accumulate) {
    wgrad_reduce_body(t.j[blockIdx.y], accumulate);
}

// ------------------------------------------------------------------------------------------
static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
static unsigned reduce_grid_for(size_t items) {   // 64 items per workgroup of the weight-gradient reductions
    size_t b = (items + 63) / 64;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int launch_pack_weight_t(const float *w, int Co, int Ci, int ci_lo, int n_out, int dtype, float scale, void *wp,
                         int unit_off, int cout_pad, hipStream_t stream) {
    const int cpu = unit_channels(dtype);
    const size_t total = (size_t)((Co + cpu - 1) / cpu) * 9 * cout_pad;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_t_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, stream, w, (__bf16 *)wp,
                           Co, Ci, ci_lo, n_out, unit_off, cout_pad, scale);
    else
        hipLaunchKernelGGL(pack_weight_t_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, w, (float *)wp, Co,
                           Ci, ci_lo, n_out, unit_off, cout_pad, scale);
    return check_hip(hipGetLastError(), "pack_weight_t launch");
}

int launch_pack_weight_t_batch(const PackTJob *jobs, int n, int dtype, hipStream_t stream) {
    for (int at = 0; at < n; at += PACK_BATCH) {
        PackTJobTable t;
        const int m = n - at < PACK_BATCH ? n - at : PACK_BATCH;
        size_t biggest = 0;      // items of the largest job: the blocks of a row stride over their job's items
        for (int i = 0; i < m; ++i) {
            t.j[i] = jobs[at + i];
            const size_t items = (size_t)((jobs[at + i].Co + 7) / 8) * 9 * jobs[at + i].cout_pad;
            if (items > biggest) biggest = items;
        }
        unsigned gx = (unsigned)((biggest + 1023) / 1024);
        gx = gx < 32 ? 32 : gx > 512 ? 512 : gx;
        if (dtype == SR355_DTYPE_BF16)
            hipLaunchKernelGGL(pack_weight_t_batch_kernel<__bf16>, dim3(gx, m), dim3(256), 0, stream, t);
        else
            hipLaunchKernelGGL(pack_weight_t_batch_kernel<float>, dim3(gx, m), dim3(256), 0, stream, t);
        if (check_hip(hipGetLastError(), "pack_weight_t batch launch")) return SR355_ERR_HIP;
    }
    return SR355_OK;
}

int launch_pack_weight_t_table(const PackTJob *jobs_host, int n, int dtype, void *table_dev, hipStream_t stream) {
    if (n < 1) return SR355_OK;
    if (n > 65535) {
        set_error("pack_weight_t table: %d jobs exceed one launch", n);
        return SR355_ERR_UNSUPPORTED;
    }
    if (check_hip(hipMemcpyAsync(table_dev, jobs_host, (size_t)n * sizeof(PackTJob), hipMemcpyHostToDevice, stream),
                  "hipMemcpyAsync"))
        return SR355_ERR_HIP;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_t_table_kernel<__bf16>, dim3(PACKT_BLOCKS, n), dim3(256), 0, stream, (const PackTJob *)table_dev);
    else
        hipLaunchKernelGGL(pack_weight_t_table_kernel<float>, dim3(PACKT_BLOCKS, n), dim3(256), 0, stream, (const PackTJob *)table_dev);
    return check_hip(hipGetLastError(), "pack_weight_t table launch");
}

int launch_pool2x2(const void *in, int N, int H, int W, int U, int dtype, const void *mask, int mask_units,
                   int mask_unit_off, float mask_slope, float *out_f32, void *out_act, int out_units, int out_unit_off,
                   hipStream_t stream) {
    const size_t total = (size_t)N * H * U * W;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(pool2x2_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, stream, (const __bf16 *)in, N, H,
                           W, U, (const __bf16 *)mask, mask_units, mask_unit_off, mask_slope, out_f32,
                           (__bf16 *)out_act, out_units, out_unit_off);
    else
        hipLaunchKernelGGL(pool2x2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float *)in, N, H, W,
                           U, (const float *)mask, mask_units, mask_unit_off, mask_slope, out_f32, (float *)out_act,
                           out_units, out_unit_off);
    return check_hip(hipGetLastError(), "pool2x2 launch");
}

// Slab geometry of a weight-gradient launch: chunks of 32 input channels x G pixel-range workgroups.
void wgrad_geometry(int cin, int N, int H, int W, int *chunks, int *G) {
    const int cin_units = (cin + 7) / 8;
    *chunks = (cin_units + 3) / 4;
    const long long tiles = (long long)N * ((H + WTH - 1) / WTH) * ((W + 31) / 32);
    // sixteen tiles per workgroup (SR355_WGRAD_TILES overrides): every workgroup ends by writing a 36.9 KB slab that the
    // reduction reads again, and at eight tiles that traffic was half of a wide layer's launch; measured per
    // discriminator / VGG layer at 32 images and on the training iterations: 8 -> 16 tiles 1,834 -> 1,554 us summed over the
    // layers, pixel iteration -1.8 %, adversarial -2 %; 32 is level with 16 overall and worse at the training shape (128
    // tiles), 64 is 35 % slower.
    static const int per_wg = [] { const char *e = getenv("SR355_WGRAD_TILES"); const int v = e ? atoi(e) : 16; return v < 1 ? 16 : v; }();
    long long g = tiles / per_wg;
    const long long cap = 768 / *chunks;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *G = (int)g;
}

static size_t wgrad_slice_floats(int cin, int N, int H, int W) {   // one 32-output slice of a sliced launch, 256-byte units
    int chunks, G;
    wgrad_geometry(cin, N, H, W, &chunks, &G);
    return ((((size_t)chunks * G * 9 * 32 + G) * 32 * sizeof(float) + 255) & ~(size_t)255) / sizeof(float);
}

size_t wgrad_workspace_bytes(int cout, int cin, int N, int H, int W) {
    if (cout > 64) return (size_t)((cout + 31) / 32) * wgrad_slice_floats(cin, N, H, W) * sizeof(float);
    int chunks, G;
    wgrad_geometry(cin, N, H, W, &chunks, &G);
    // weight slabs, then one row of per-workgroup bias sums
    return ((size_t)chunks * G * 9 * 32 + G) * cout_pad_for(cout) * sizeof(float);
}

static int wgrad_set_attr(int mbi) {
    static std::atomic<bool> attr_done[2][64];      // (zero-initialised; acquire / release: threads may race here)
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[mbi][dev].load(std::memory_order_acquire)) {
        hipError_t e = mbi ? hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad3x3_kernel_mb2),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WGeo<2>::LDS_BYTES)
                           : hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad3x3_kernel_mb1),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WGeo<1>::LDS_BYTES);
        if (check_hip(e, "hipFuncSetAttribute")) return SR355_ERR_HIP;
        attr_done[mbi][dev].store(true, std::memory_order_release);
    }
    return SR355_OK;
}

size_t wgrad_block_workspace_bytes(int nf, int gc, int N, int H, int W) {
    size_t total = 0;
    for (int k = 0; k < 5; ++k) total += (wgrad_workspace_bytes(k < 4 ? gc : nf, nf + k * gc, N, H, W) + 255) & ~(size_t)255;
    return total;
}

// The five weight / bias gradients of each of n_blocks (1..6) dense blocks (esrgan_model.py:18-24) in ONE launch (blockIdx.z
// = (block, conv): the growth convs, gc <= 32 outputs each, and conv5 as two jobs of 32 outputs) plus one reduction.
// `act[b]` = block b's forward buffer, `stack[b]` = its gradient stack [dOut(nf) | dY4 | dY3 | dY2 | dY1];
// dw[b][k] / db[b][k] receive conv(k+1)'s gradients; conv5's is scaled by s5[b], the growth convs' by s_growth.
// Block b's slabs start at slabs + b * wgrad_block_workspace_bytes.  With `defer` (room for 5 * n_blocks jobs) the
// reduction is left to the caller: launch_wgrad_reduce_jobs on the collected jobs.
// At training-patch sizes (16 patches of 32 x 32: 128 tiles) a block's launch is a few hundred workgroups on 256 CUs,
// bound by the latency of each tile's fetch: several blocks in one launch keep every CU supplied with resident
// workgroups (measured with eight tiles a workgroup: 19 us per block alone, 14.7 with three blocks a launch, 12.9 with six;
// at sixteen tiles a workgroup nine blocks are 1,872 workgroups, 3.7 rounds of the 512 the chip holds).
int launch_wgrad3x3_blocks(int n_blocks, const void *const *act, const void *const *stack, int units, int nf, int gc,
                           int N, int H, int W, const float *s5, float *const (*dw)[5], float *const (*db)[5],
                           float *slabs, hipStream_t stream, WgradReduceJob *defer, float s_growth) {
    if (cout_pad_for(gc) != 32 || cout_pad_for(nf) > 64 || ((nf | gc) & 7)) {
        set_error("wgrad3x3 block: nf=%d gc=%d not supported", nf, gc);
        return SR355_ERR_UNSUPPORTED;
    }
    const int n5 = cout_pad_for(nf) / 32, per_block = 4 + n5;
    if (n_blocks < 1 || n_blocks * per_block > WGRAD_MAX_JOBS || n_blocks * 5 > WGRAD_MAX_JOBS) {
        set_error("wgrad3x3 blocks: %d blocks exceed one launch", n_blocks);
        return SR355_ERR_INVALID;
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const unsigned char *)act[0];
    a.dy = (const unsigned char *)stack[0];
    a.slabs = slabs;
    a.N = N, a.H = H, a.W = W;
    a.in_units = units, a.ups = 0, a.dy_units = units;
    a.tiles_x = (W + 31) / 32;
    a.tiles_y = (H + WTH - 1) / WTH;
    WgradReduceTable rt;
    size_t off = 0;   // floats
    int max_chunks = 0, max_g = 0;
    for (int b = 0; b < n_blocks; ++b) {
        const long long in_off = (const unsigned char *)act[b] - a.in, dy_off = (const unsigned char *)stack[b] - a.dy;
        for (int k = 0; k < 5; ++k) {
            const int cin = nf + k * gc, cout = k < 4 ? gc : nf, co_pad = cout_pad_for(cout);
            int chunks, G;
            wgrad_geometry(cin, N, H, W, &chunks, &G);
            const int dy_unit = k < 4 ? (nf + (3 - k) * gc) / 8 : 0;   // stack order: dOut, dY4, dY3, dY2, dY1
            const size_t bias_off = off + (size_t)chunks * G * 9 * co_pad * 32;
            rt.j[b * 5 + k] = WgradReduceJob{slabs + off, slabs + bias_off, dw[b][k], db[b][k], cout, cin, co_pad, G,
                                             k < 4 ? s_growth : s5[b]};
            // conv5 takes part as n5 jobs of 32 outputs each, written side by side into its slabs
            for (int half = 0; half < (k < 4 ? 1 : n5); ++half) {
                const int z = b * per_block + k + half;
                a.z_chunks[z] = chunks, a.z_groups[z] = G, a.z_dy_unit_off[z] = dy_unit + half * 4;
                a.z_co_off[z] = half * 32, a.z_co_stride[z] = co_pad;
                a.z_slab_off[z] = (long long)off, a.z_bias_off[z] = (long long)bias_off;
                a.z_in_off[z] = in_off, a.z_dy_off[z] = dy_off;
            }
            if (chunks > max_chunks) max_chunks = chunks;
            if (G > max_g) max_g = G;
            off += ((wgrad_workspace_bytes(cout, cin, N, H, W) + 255) & ~(size_t)255) / sizeof(float);
        }
    }
    int rc;
    if ((rc = wgrad_set_attr(0))) return rc;
    a.n_batch = n_blocks * per_block;
    hipLaunchKernelGGL(wgrad3x3_kernel_mb1, dim3(max_g, max_chunks, a.n_batch), dim3(192), WGeo<1>::LDS_BYTES, stream, a);
    if (check_hip(hipGetLastError(), "wgrad3x3 block launch")) return SR355_ERR_HIP;
    if (defer) {
        for (int k = 0; k < 5 * n_blocks; ++k) defer[k] = rt.j[k];
        return SR355_OK;
    }
    return launch_wgrad_reduce_jobs(rt.j, 5 * n_blocks, nf, gc, stream);
}

int launch_wgrad3x3_block(const void *act, const void *stack, int units, int nf, int gc, int N, int H, int W, float s5,
                          float *const dw[5], float *const db[5], float *slabs, hipStream_t stream,
                          WgradReduceJob *defer, float s_growth) {
    float *dw1[1][5], *db1[1][5];
    for (int k = 0; k < 5; ++k) dw1[0][k] = dw[k], db1[0][k] = db[k];
    return launch_wgrad3x3_blocks(1, &act, &stack, units, nf, gc, N, H, W, &s5, dw1, db1, slabs, stream, defer, s_growth);
}

// One launch for the reductions of up to three dense blocks (blockIdx.y = job).
int launch_wgrad_reduce_jobs(const WgradReduceJob *jobs, int n, int nf, int gc, hipStream_t stream) {
    if (n < 1) return SR355_OK;
    if (n > WGRAD_MAX_JOBS) {
        set_error("wgrad reduce: %d jobs exceed one launch", n);
        return SR355_ERR_INVALID;
    }
    WgradReduceTable rt;
    for (int k = 0; k < n; ++k) rt.j[k] = jobs[k];
    const size_t biggest = (size_t)9 * nf * ((nf + 4 * gc + 3) / 4) + nf;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(reduce_grid_for(biggest), n), dim3(256), 0, stream, rt, 0);
    return check_hip(hipGetLastError(), "wgrad reduce batch launch");
}

// The fixed-order reduction of per-workgroup slabs [chunk][G][9][CO][32] (+ bias rows [G][CO]) on its own: the
// exact-fp32 weight gradient (wgrad_f32.hip) writes the same slab layout.
int launch_wgrad_reduce(const float *slabs, float *dw, int cout, int cin, int CO, int G, float scale, int accumulate,
                        const float *bias_slabs, float *db, hipStream_t stream) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid_for((size_t)9 * cout * ((cin + 3) / 4) + cout)), dim3(256), 0,
                       stream, slabs, dw, cout, cin, CO, G, scale, accumulate, bias_slabs, db);
    return check_hip(hipGetLastError(), "wgrad reduce launch");
}

// A conv with more than 64 output channels: its 32-output slices are the jobs of batched launches
// (blockIdx.z = slice, all reading the same input), WGRAD_MAX_JOBS slices per launch, each followed by one batched
// reduction.  One launch per 64 outputs would leave most of the chip idle at the small images such layers have.
static int launch_wgrad3x3_sliced(const void *in, int in_units, int ups, const void *dy, int dy_units, int dy_unit_off,
                                  int N, int H, int W, int cout, int cin, float scale, int accumulate, float *dw,
                                  float *db, float *slabs, hipStream_t stream, int dy_stride2) {
    int rc;
    if ((rc = wgrad_set_attr(0))) return rc;
    int chunks, G;
    wgrad_geometry(cin, N, H, W, &chunks, &G);
    const size_t slice_floats = wgrad_slice_floats(cin, N, H, W);
    // (a last slice of fewer than 32 outputs: its workgroups still multiply four units of dy -- whatever lies behind
    // the conv's own units in the image, zeros past its end -- into slab rows the reduction never reads)
    const int n_slices = (cout + 31) / 32;
    for (int s0 = 0; s0 < n_slices; s0 += WGRAD_MAX_JOBS) {
        const int m = n_slices - s0 < WGRAD_MAX_JOBS ? n_slices - s0 : WGRAD_MAX_JOBS;
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.in = (const unsigned char *)in;
        a.dy = (const unsigned char *)dy;
        a.slabs = slabs;
        a.N = N, a.H = H, a.W = W;
        a.in_units = in_units, a.ups = ups;
        a.dy_units = dy_units;
        a.dy_s2 = dy_stride2 ? 1 : 0;
        a.tiles_x = (W + 31) / 32;
        a.tiles_y = (H + WTH - 1) / WTH;
        a.n_batch = m;
        WgradReduceTable rt;
        memset(&rt, 0, sizeof(rt));
        for (int k = 0; k < m; ++k) {
            const int sl = s0 + k;
            const size_t off = (size_t)sl * slice_floats;
            const size_t bias_off = off + (size_t)chunks * G * 9 * 32 * 32;
            a.z_chunks[k] = chunks, a.z_groups[k] = G, a.z_dy_unit_off[k] = dy_unit_off + sl * 4;
            a.z_co_off[k] = 0, a.z_co_stride[k] = 32;
            a.z_slab_off[k] = (long long)off;
            a.z_bias_off[k] = db ? (long long)bias_off : -1;
            rt.j[k] = WgradReduceJob{slabs + off, slabs + bias_off, dw + (size_t)sl * 32 * cin * 9, db ? db + sl * 32 : nullptr,
                                     cout - sl * 32 < 32 ? cout - sl * 32 : 32, cin, 32, G, scale};
        }
        hipLaunchKernelGGL(wgrad3x3_kernel_mb1, dim3(G, chunks, m), dim3(192), WGeo<1>::LDS_BYTES, stream, a);
        if (check_hip(hipGetLastError(), "wgrad3x3 sliced launch")) return SR355_ERR_HIP;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(reduce_grid_for((size_t)9 * 32 * ((cin + 3) / 4) + 32), m),
                           dim3(256), 0, stream, rt, accumulate);
        if (check_hip(hipGetLastError(), "wgrad sliced reduce launch")) return SR355_ERR_HIP;
    }
    return SR355_OK;
}

int launch_wgrad3x3(const void *in, int in_units, int ups, const void *dy, int dy_units, int dy_unit_off, int N, int H,
                    int W, int cout, int cin, float scale, int accumulate, float *dw, float *db, float *slabs,
                    hipStream_t stream, int dy_stride2) {
    if (cout > 64)
        return launch_wgrad3x3_sliced(in, in_units, ups, dy, dy_units, dy_unit_off, N, H, W, cout, cin, scale,
                                      accumulate, dw, db, slabs, stream, dy_stride2);
    const int co_pad = cout_pad_for(cout);
    if (co_pad != 32 && co_pad != 64) {
        set_error("wgrad3x3: cout=%d not supported", cout);
        return SR355_ERR_UNSUPPORTED;
    }
    int chunks, G;
    wgrad_geometry(cin, N, H, W, &chunks, &G);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const unsigned char *)in;
    a.dy = (const unsigned char *)dy;
    a.slabs = slabs;
    a.bias_slabs = db ? slabs + (size_t)chunks * G * 9 * co_pad * 32 : nullptr;
    a.N = N, a.H = H, a.W = W;
    a.in_units = in_units, a.ups = ups;
    a.dy_units = dy_units, a.dy_unit_off = dy_unit_off;
    a.dy_s2 = dy_stride2 ? 1 : 0;
    a.tiles_x = (W + 31) / 32;
    a.tiles_y = (H + WTH - 1) / WTH;
    static std::atomic<bool> attr_done[2][64];      // (zero-initialised; acquire / release: threads may race here)
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) dev = 0;
    const int mbi = co_pad == 64 ? 1 : 0;
    if (!attr_done[mbi][dev].load(std::memory_order_acquire)) {
        hipError_t e = mbi ? hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad3x3_kernel_mb2),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WGeo<2>::LDS_BYTES)
                           : hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad3x3_kernel_mb1),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WGeo<1>::LDS_BYTES);
        if (check_hip(e, "hipFuncSetAttribute")) return SR355_ERR_HIP;
        attr_done[mbi][dev].store(true, std::memory_order_release);
    }
    dim3 grid(G, chunks, 1);
    if (mbi)
        hipLaunchKernelGGL(wgrad3x3_kernel_mb2, grid, dim3(192), WGeo<2>::LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL(wgrad3x3_kernel_mb1, grid, dim3(192), WGeo<1>::LDS_BYTES, stream, a);
    if (check_hip(hipGetLastError(), "wgrad3x3 launch")) return SR355_ERR_HIP;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_grid_for((size_t)9 * cout * ((cin + 3) / 4) + cout)), dim3(256), 0,
                       stream, slabs, dw, cout, cin, co_pad, G, scale, accumulate, a.bias_slabs, db);
    if (check_hip(hipGetLastError(), "wgrad reduce launch")) return SR355_ERR_HIP;
    return SR355_OK;
}

// ------------------------------------------------------------------------------------------
// Scratch of the allocation-free weight gradient (sr355_conv3x3_wgrad_static): the per-workgroup partial slabs go
// to this array of the code object instead of a caller workspace.  Every GPU the library runs on holds its own
// copy until the process ends.  In host code the array's name is only a handle for the symbol lookup; the device
// address of the current device's copy comes from hipGetSymbolAddress, once per device, and is what the kernels get.
static __device__ __align__(256) unsigned char scratch[SR355_STATIC_SCRATCH_BYTES];

int wgrad_static_scratch(float **out) {
    static std::mutex mu;
    static void *ptr_of[64] = {};
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) {
        set_error("conv3x3_wgrad_static: device ordinal %d outside [0, 64)", dev);
        return SR355_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of[dev]) {
        void *p = nullptr;
        if (check_hip(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)")) return SR355_ERR_HIP;
        if (!p) {
            set_error("conv3x3_wgrad_static: the scratch symbol has no address on device %d", dev);
            return SR355_ERR_HIP;
        }
        ptr_of[dev] = p;
    }
    *out = (float *)ptr_of[dev];
    return SR355_OK;
}

}  // namespace sr355
