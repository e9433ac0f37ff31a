// conv3x3 forward, ring kernel with a branch-free stage body (bf16 path, large images).
//
// Same tiles, ring, operand maps and epilogue as conv3x3_ring (conv3x3_ring.hip); replaces nn.Conv2d + cat +
// LeakyReLU + skips of /root/reference/esrgan/models/esrgan_model.py:18-24.  What changes is how one wave
// keeps the matrix pipe busy on its own.  In conv3x3_ring the LDS fragment reads, the DMA issue and the MFMAs
// of a stage run as separate phases and a wave alone on its SIMD keeps the pipe about half busy: fine while
// two workgroups share the SIMD in the same phase, wasteful whenever the partner is in its prologue, at a
// barrier or in its epilogue (the partner's gap could be filled, but is not).  Here
//   * the DMA issue of a stage has no branches: every wave issues the same number of instructions per stage
//     (a wave whose last slot does not exist sends an out-of-range load to a 1 KiB dummy area), the buffer
//     descriptors are rebuilt from scalarised words so that no readfirstlane loop wraps the loads, and the
//     issue cursor / next-tile set-up (which do branch) run after the MFMAs of the stage;
//   * so the MFMAs of tap column kx, the LDS reads of column kx+1 and the DMA issue form one basic block, and
//     `sched_group_barrier` interleaves them one MFMA at a time (an MFMA occupies the pipe for 32 cycles; an
//     LDS read or a DMA issues in its shadow).
#include <stdlib.h>

#include "conv_epilogue.h"
#include <atomic>

#include "sr355_internal.h"

namespace sr355 {

extern unsigned long long *g_conv_trace;
#ifdef SR355_TRACE
#define SR355_FSTAMP(K)                                                                               \
    if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (K)] = wall_clock64();
// in-stage stamps of wave 0 of every workgroup, taken in steady state (global stage 3): shader-clock ticks
#define SR355_SSTAMP(K)                                                                               \
    if (a.trace && g == 3) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        st_[K] = __builtin_readcyclecounter();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#else
#define SR355_FSTAMP(K)
#define SR355_SSTAMP(K)
#endif

#ifdef SR355_EXP_NOWEIGHTS   /* timing experiment (wrong results): no weight traffic, as if the weights were resident */
#define SR355_EXP_NOWEIGHTS_V true
#else
#define SR355_EXP_NOWEIGHTS_V false
#endif

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

// one MFMA on two 16-byte fragments: bf16, or the same bits read as IEEE fp16
template <bool F16>
__device__ __forceinline__ f32x16 mma16(const bf16x8 &w, const bf16x8 &x, const f32x16 &c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
}

constexpr int TWP = 34;          // tile width 32 + halo
constexpr int RSU = 2;           // units per stage
constexpr int OOB = 0x7fffff00;  // byte offset beyond any descriptor range -> DMA writes zeros

template <int MB, int NB, int R>
struct FineGeo {
    static constexpr int WAVES = 4;
    static constexpr int CO = MB * 32;
    static constexpr int TH = WAVES * NB;
    static constexpr int NPIX = (TH + 2) * TWP;
    static constexpr int NIN = RSU * NPIX;
    static constexpr int NIN_I = (NIN + 63) / 64;
    static constexpr int NWT = RSU * 9 * CO;
    static constexpr int NWT_I = NWT / 64;
    static constexpr int JTOT = NIN_I + NWT_I;
    static constexpr int KMAX = (JTOT + WAVES - 1) / WAVES;     // DMA instructions per wave and stage, all waves
    static constexpr int W_OFF = NIN_I * 1024;
    static constexpr int STAGE_BYTES = JTOT * 1024;
    static constexpr int BIAS_OFF = R * STAGE_BYTES;
    static constexpr int DUMMY_OFF = BIAS_OFF + 256;            // landing area of the padding loads
    static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
    static constexpr int N_READS = 3 * MB + NB + 2;             // LDS reads of one tap column
    static constexpr int N_MFMA = 3 * MB * NB;                  // MFMAs of one tap column
    static_assert(NWT % 64 == 0, "weight image must be whole wave-instructions");
    static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");
    static_assert(CO <= 64, "bias image is one 256-byte DMA");
    static_assert((R - 2) * KMAX <= 63, "vmcnt field is 6 bits");
    // (4-row tiles, NB = 1, are for launches that cannot fill the chip: their stage is short on purpose)
    static_assert(2 * N_MFMA >= N_READS && (N_MFMA >= KMAX || NB == 1), "not enough MFMAs to hide the reads / the DMA issue");
};
}  // namespace

// EPI: 0 = every epilogue form, chosen per tile from the arguments; 1 / 2 = the launcher has checked that the lean
// bias + LeakyReLU + bf16 (1) / fp16 (2) output form applies and the kernel holds nothing else (143 registers instead
// of 241, no scalar spills in the stage loop, a third less code: the 64 -> 64 convs of the image side run 4 % faster).
// -DSR355_NO_LEAN_INST builds the library without these instantiations.
template <int MB, int NB, int R, int WPS, bool F16, int EPI = 0>
__global__ __launch_bounds__(256, WPS) void conv3x3_fine(const ConvArgs a) {
    typedef FineGeo<MB, NB, R> G;
    constexpr int WAVES = G::WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SR355_FSTAMP(0);
    {
        // The argument block spans four 64-byte lines and the set-up code below reads it piecemeal, one scalar-cache
        // miss after the other; touching every line first puts the misses in flight together.
        typedef __attribute__((address_space(4))) const unsigned kernarg_word;
        kernarg_word *ka = (kernarg_word *)__builtin_amdgcn_kernarg_segment_ptr();
        static_assert(sizeof(ConvArgs) > 192 && sizeof(ConvArgs) <= 256, "argument block: four lines");
        const unsigned k0 = ka[0], k1 = ka[16], k2 = ka[32], k3 = ka[48];
        asm volatile("" ::"s"(k0), "s"(k1), "s"(k2), "s"(k3));
    }
#ifdef SR355_TRACE
    const unsigned long long sclk0_ = __builtin_readcyclecounter();
#endif

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

    // Blocks b and b+8 share an XCD: give the blocks of one XCD consecutive tiles of every round.
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;
    const int nwg = gridDim.x;
    int first;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        first = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int n_my = first < n_tiles ? (n_tiles - first + nwg - 1) / nwg : 0;
    const int total_g = n_my * S;

    // descriptors are kept as scalar words and rebuilt where they are used
    const unsigned long long w_base = (unsigned long long)(a.w) + (size_t)cb * G::CO * 16;
    const int w_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)w_base);
    const int w_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(w_base >> 32));
    const int w_len = __builtin_amdgcn_readfirstlane(w_stage_stride * S);
    const int in_len = __builtin_amdgcn_readfirstlane(img_bytes);
#define SR355_RSRC(LO, HI, LEN)                                                                       \
    __builtin_amdgcn_make_buffer_rsrc(                                                                \
        (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(HI) << 32) |           \
                 (unsigned)__builtin_amdgcn_readfirstlane(LO)),                                       \
        0, __builtin_amdgcn_readfirstlane(LEN), 0x00020000)
    if (wave == 0) {   // bias: oldest operation of wave 0, retired by its first wait
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.bias) + cb * G::CO, 0, G::CO * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void *)(smem + G::BIAS_OFF), 4, lane * 4, 0, 0, 0);
    }

    // per-lane source offsets of this wave's DMA instructions (j = k*WAVES + wave).  What does not depend on the tile
    // is computed once: for an activation item its position relative to the tile origin (row + 1 in the low half
    // of `rel`, column + 1 in the high half; an item that never exists gets 0x7fff in both: an image's bytes fit 31
    // bits, so no image has that many rows AND columns) and the part of its
    // offset that position contributes (`cst`); weight items do not depend on the tile at all.  Per tile there remain
    // two adds, two unsigned compares and a select per activation item, and the tile coordinates advance by the
    // launch's stride without a division.
    constexpr int KIN = (G::NIN_I + WAVES - 1) / WAVES;   // instructions of a wave that can be activation items
    int voff[G::KMAX], rel[KIN], cst[KIN];
#pragma unroll
    for (int k = 0; k < G::KMAX; ++k) {
        const int j = k * WAVES + wave;
        voff[k] = OOB;
        if (j >= G::NIN_I && j < G::JTOT) {   // weight items do not depend on the tile
            const int jw = (j - G::NIN_I) * 64 + lane;
            const int u = jw / (9 * G::CO);
            const int rem = jw - u * 9 * G::CO;
            const int tap = rem / G::CO;
            const int co = rem - tap * G::CO;
            voff[k] = ((u * 9 + tap) * a.cout_pad + co) * 16;
        }
        if (k < KIN) {
            rel[k < KIN ? k : 0] = 0x7fff7fff, cst[k < KIN ? k : 0] = 0;
            if (j < G::NIN_I) {
                const int i = j * 64 + lane;
                const int u = i / G::NPIX;
                const int p = i - u * G::NPIX;
                const int ty = p / TWP;
                const int tx = p - ty * TWP;
                const bool exists = i < G::NIN && !SR355_DBG(a, 1) && !(a.zins && (((ty - 1) | (tx - 1)) & 1));
                rel[k < KIN ? k : 0] = exists ? (ty | (tx << 16)) : 0x7fff7fff;
                cst[k < KIN ? k : 0] = ((((ty - 1) >> a.ups) * a.in_units + u) * Win + ((tx - 1) >> a.ups)) * 16;
            }
            asm volatile("" : "+v"(rel[k < KIN ? k : 0]), "+v"(cst[k < KIN ? k : 0]));   // (kept, not recomputed per tile)
        }
    }
    int in_lo = w_lo, in_hi = w_hi;   // base of the image the issue cursor is in
    int it_tile = 0, it_stage = 0;
    // tile coordinates of the issue cursor and of the tile whose epilogue comes next; one step of nwg tiles
    int c_tx, c_ty, c_n;
    {
        c_tx = first % a.tiles_x;
        const int tr_ = first / a.tiles_x;
        c_ty = tr_ % a.tiles_y;
        c_n = tr_ / a.tiles_y;
    }
    int e_tx = c_tx, e_ty = c_ty, e_n = c_n;
    const int d_tx = nwg % a.tiles_x, d_q = nwg / a.tiles_x;
    const int d_ty = d_q % a.tiles_y, d_n = d_q / a.tiles_y;
#define SR355_TILE_STEP(TX, TY, NN)                                                                   \
    {                                                                                                 \
        TX += d_tx;                                                                                   \
        const int cx_ = TX >= a.tiles_x;                                                              \
        TX -= cx_ ? a.tiles_x : 0;                                                                    \
        TY += d_ty + cx_;                                                                             \
        const int cy_ = TY >= a.tiles_y;                                                              \
        TY -= cy_ ? a.tiles_y : 0;                                                                    \
        NN += d_n + cy_;                                                                              \
    }

    // offsets of the tile at (c_tx, c_ty, c_n)
#define SR355_TILE_SETUP()                                                                            \
    {                                                                                                 \
        const unsigned long long ib_ = (unsigned long long)(a.in) + (size_t)c_n * img_bytes;          \
        in_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)ib_);                                   \
        in_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib_ >> 32));                           \
        const int y0_ = c_ty * G::TH, x0_ = c_tx * 32;   /* (even: the shifts below split exactly) */ \
        const int base_ = (((y0_ >> a.ups) * a.in_units) * Win + (x0_ >> a.ups)) * 16;                \
        _Pragma("unroll") for (int k = 0; k < KIN; ++k) {                                             \
            const int j = k * WAVES + wave;                                                           \
            if (j < G::NIN_I) {                                                                       \
                const int y = y0_ - 1 + (rel[k] & 0xffff), x = x0_ - 1 + (rel[k] >> 16);              \
                const bool valid = ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W);        \
                voff[k] = valid ? base_ + cst[k] : OOB;                                               \
            }                                                                                         \
        }                                                                                             \
    }

    // One stage of DMA into ring slot SLOT, without a branch: instruction k of wave w is item j = k*WAVES + w;
    // whether it is an activation item, a weight item or padding is decided by scalar selects.  Past the end of
    // the tile list the same instructions go out with out-of-range sources (zeros into a free slot).
#define SR355_ISSUE_DMA(SLOT)                                                                         \
    {                                                                                                 \
        const bool live_ = it_tile < n_my;                                                            \
        const int si_ = it_stage >= n_real ? it_stage - n_real : it_stage;                            \
        const int soff_in_ = __builtin_amdgcn_readfirstlane(si_ * in_stage_stride);                   \
        const int soff_w_ = __builtin_amdgcn_readfirstlane(it_stage * w_stage_stride);                \
        const int dst0_ = (SLOT) * G::STAGE_BYTES + wave * 1024;                                      \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) {                                         \
            const int j = k * WAVES + wave;                                                           \
            const bool all_in_ = k * WAVES + WAVES - 1 < G::NIN_I;                                    \
            const bool all_w_ = k * WAVES >= G::NIN_I && k * WAVES + WAVES - 1 < G::JTOT;             \
            const bool is_in_ = all_in_ || (!all_w_ && j < G::NIN_I);                                 \
            const bool exists_ = all_in_ || all_w_ || j < G::JTOT;                                    \
            const int lo_ = is_in_ ? in_lo : w_lo, hi_ = is_in_ ? in_hi : w_hi;                       \
            const int len_ = is_in_ ? in_len : w_len, so_ = is_in_ ? soff_in_ : soff_w_;              \
            const int dst_ = exists_ ? dst0_ + k * WAVES * 1024 : G::DUMMY_OFF;                       \
            const int vo_ = (live_ && exists_ && !(SR355_EXP_NOWEIGHTS_V && !is_in_)) ? voff[k] : OOB; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(SR355_RSRC(lo_, hi_, len_), (lds_void *)(smem + dst_), 16, vo_, \
                                                     so_, 0, 0);                                      \
        }                                                                                             \
    }
    // advance the issue cursor (and set the next tile up): branches, kept out of the MFMA blocks
#define SR355_ISSUE_ADVANCE()                                                                         \
    if (it_tile < n_my && ++it_stage == S) {                                                          \
        it_stage = 0;                                                                                 \
        if (++it_tile < n_my) {                                                                       \
            SR355_TILE_STEP(c_tx, c_ty, c_n);                                                         \
            SR355_TILE_SETUP();                                                                       \
        }                                                                                             \
    }

    SR355_FSTAMP(12);
    if (n_my > 0) SR355_TILE_SETUP();
    SR355_FSTAMP(13);
#pragma unroll
    for (int g = 0; g < R - 1; ++g) {
        SR355_ISSUE_DMA(g);
        SR355_ISSUE_ADVANCE();
    }
    SR355_FSTAMP(1);

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
                        acc[mb][nb] = mma16<F16>(WF[ky][mb], XF[rho], acc[mb][nb]);                   \
                }                                                                                     \
            }                                                                                         \
    }
    // MFMA = 0x008, VMEM read = 0x020, DS read = 0x100
#define SR355_INTERLEAVE(N_DS, N_VMEM, SYNC)                                                          \
    {                                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < G::N_MFMA; ++i_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SYNC);                                     \
            if (i_ < (N_DS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, SYNC);                    \
            if (i_ + G::N_MFMA < (N_DS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, SYNC);        \
            if (i_ < (N_VMEM)) __builtin_amdgcn_sched_group_barrier(0x020, 1, SYNC);                  \
        }                                                                                             \
    }

    int slot = 0, issue_slot = R - 1, ct_stage = 0, ct_tile = 0;
#ifdef SR355_TRACE
    unsigned long long st_[8] = {};
#endif
    for (int g = 0; g < total_g; ++g) {
        SR355_SSTAMP(0);
        // stage g has landed for this wave when at most (R-2) younger stages are outstanding; after the
        // barrier it has landed for every wave, and slot (g-1)%R is free to refill
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * G::KMAX) : "memory");
        SR355_SSTAMP(1);
        __builtin_amdgcn_s_barrier();
        SR355_SSTAMP(2);
        if (g == 0) SR355_FSTAMP(2);
        const unsigned char *buf = smem + slot * G::STAGE_BYTES;
        const unsigned char *pin = buf + lane_in;
        const unsigned char *pw = buf + lane_w;
        bf16x8 wfA[3][MB], xfA[NB + 2], wfB[3][MB], xfB[NB + 2];
        SR355_LOAD_FRAGS(wfA, xfA, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef SR355_TRACE
        if (a.trace && g == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        SR355_SSTAMP(3);
        // column 0: its MFMAs, the reads of column 1 and this stage's DMA issue, one of each at a time
        SR355_ISSUE_DMA(issue_slot);
        SR355_LOAD_FRAGS(wfB, xfB, 1);
        SR355_MMA_FRAGS(wfA, xfA);
        SR355_INTERLEAVE(G::N_READS, G::KMAX, 0);
        __builtin_amdgcn_sched_barrier(0);
        SR355_SSTAMP(4);
        // column 1 under the reads of column 2
        SR355_LOAD_FRAGS(wfA, xfA, 2);
        SR355_MMA_FRAGS(wfB, xfB);
        SR355_INTERLEAVE(G::N_READS, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        SR355_SSTAMP(5);
        SR355_MMA_FRAGS(wfA, xfA);
        __builtin_amdgcn_sched_barrier(0);
        SR355_SSTAMP(6);
        issue_slot = issue_slot + 1 == R ? 0 : issue_slot + 1;
        SR355_ISSUE_ADVANCE();
        slot = slot + 1 == R ? 0 : slot + 1;
        SR355_SSTAMP(7);

        if (++ct_stage == S) {
            // ---- epilogue of tile ct_tile ----
            if (ct_tile < 4) SR355_FSTAMP(3 + 2 * ct_tile);
            const int ty_i = e_ty, n = e_n;
            const int x = e_tx * 32 + r;
            SR355_TILE_STEP(e_tx, e_ty, e_n);
            if constexpr (EPI == 1) {
                conv_epilogue_tile_plain<MB, NB, false>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x,
                                                        cb * G::CO, h, !SR355_DBG(a, 2));
            } else if constexpr (EPI == 2) {
                conv_epilogue_tile_plain<MB, NB, true>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x,
                                                       cb * G::CO, h, !SR355_DBG(a, 2));
            } else if (conv_epilogue_plain_ok(a)) {
                if (a.f16_out)
                    conv_epilogue_tile_plain<MB, NB, true>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x,
                                                           cb * G::CO, h, !SR355_DBG(a, 2));
                else
                    conv_epilogue_tile_plain<MB, NB, false>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x,
                                                            cb * G::CO, h, !SR355_DBG(a, 2));
            } else if (conv_epilogue_plain_mask_ok(a)) {
                conv_epilogue_tile_plain<MB, NB, false, true, false>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB,
                                                                     x, cb * G::CO, h, !SR355_DBG(a, 2));
            } else if (conv_epilogue_plain_f32_ok(a)) {
                conv_epilogue_tile_plain<MB, NB, false, false, true>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB,
                                                                     x, cb * G::CO, h, !SR355_DBG(a, 2));
            } else if (MB == 1 && cb == 0 && conv_epilogue_rgb_ok(a)) {
                if constexpr (MB == 1)
                    conv_epilogue_rgb<NB>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x, h, !SR355_DBG(a, 2));
            } else if (conv_epilogue_tile_ok(a)) {
                conv_epilogue_tile<MB, NB>(a, acc, smem + G::BIAS_OFF, n, ty_i * G::TH + wave * NB, x, cb * G::CO, h,
                                           !SR355_DBG(a, 2));
            } else {
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
            }
            ct_stage = 0;
            if (ct_tile < 4) SR355_FSTAMP(4 + 2 * ct_tile);
            ++ct_tile;
        }
    }
    // drain the DMA still in flight (dummy stages) before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SR355_FSTAMP(15);
#ifdef SR355_TRACE
    if (a.trace && threadIdx.x == 0)   // shader-clock ticks from the first to the last stamp (clock under load)
        a.trace[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 14] = __builtin_readcyclecounter() - sclk0_;
#endif
#ifdef SR355_TRACE
    if (a.trace && threadIdx.x == 0)
        for (int k = 0; k < 8; ++k) a.trace[(4096 + blockIdx.y * gridDim.x + blockIdx.x) * 16 + k] = st_[k];
#endif
#undef SR355_RSRC
#undef SR355_TILE_SETUP
#undef SR355_TILE_STEP
#undef SR355_ISSUE_DMA
#undef SR355_ISSUE_ADVANCE
#undef SR355_LOAD_FRAGS
#undef SR355_MMA_FRAGS
#undef SR355_INTERLEAVE
}

template <int MB, int NB, int R, int WPS, bool F16, int EPI = 0>
static int launch_fine_typed(const ConvArgs &a0, hipStream_t stream) {
    typedef FineGeo<MB, NB, R> G;
    ConvArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    auto kern = conv3x3_fine<MB, NB, R, WPS, F16, EPI>;
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
        set_error("conv3x3 fine: %lld tiles is out of range", n_tiles);
        return SR355_ERR_INVALID;
    }
    int wg_per_cu = (160 * 1024) / G::LDS_BYTES;
    if (wg_per_cu > WPS) wg_per_cu = WPS;
    if (wg_per_cu < 1) wg_per_cu = 1;
    const long long slots = (long long)n_cu[dev].load(std::memory_order_relaxed) * wg_per_cu;
    dim3 grid((unsigned)(n_tiles < slots ? n_tiles : slots), a.cout_pad / G::CO, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, stream, a);
    return check_hip(hipGetLastError(), "conv3x3 fine launch");
}

#ifndef SR355_NO_LEAN_INST
// (host-side twin of conv_epilogue_plain_ok)
static bool plain_epilogue(const ConvArgs &a) {
    return !a.out_nchw && !a.out_u8 && (a.cout_real & 31) == 0 && a.out_act && !a.out_f32 && !a.mask_act && !a.res_act &&
           !a.res1 && !a.res2 && a.slope >= 0.f && a.slope <= 1.f;
}
#endif

template <int MB, int NB, int R, int WPS = 2>
static int launch_fine_inst(const ConvArgs &a, hipStream_t stream) {
#ifndef SR355_NO_LEAN_INST
    if constexpr (NB >= 2 && R == 2) {
        if (plain_epilogue(a)) {
            if (a.f16_in)
                return a.f16_out ? launch_fine_typed<MB, NB, R, WPS, true, 2>(a, stream)
                                 : launch_fine_typed<MB, NB, R, WPS, true, 1>(a, stream);
            return a.f16_out ? launch_fine_typed<MB, NB, R, WPS, false, 2>(a, stream)
                             : launch_fine_typed<MB, NB, R, WPS, false, 1>(a, stream);
        }
    }
#endif
    return a.f16_in ? launch_fine_typed<MB, NB, R, WPS, true>(a, stream)
                    : launch_fine_typed<MB, NB, R, WPS, false>(a, stream);
}

// Entry: bf16 only, cin_units even, cout_pad 32 or 64; `a` carries n_stages / n_real_stages / debug.
int launch_conv3x3_fine(const ConvArgs &a, hipStream_t stream) {
    ConvArgs b = a;
    b.trace = g_conv_trace;
    if ((a.cout_pad % 64) == 0) return launch_fine_inst<2, 2, 2>(b, stream);
    // 32 output channels: a wave's MFMA run per stage is short, and the barrier plus the first fragment reads of
    // every stage are exposed; 16-row tiles double the run (and halve the weight traffic per pixel) wherever they
    // still give every CU a tile
    static std::atomic<int> cu_of[64];
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) dev = 0;
    if (!cu_of[dev].load(std::memory_order_acquire)) {
        int n = 0;
        if (check_hip(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute"))
            return SR355_ERR_HIP;
        cu_of[dev].store(n > 0 ? n : 256, std::memory_order_release);
    }
    const int cu = cu_of[dev].load(std::memory_order_acquire);
    const long long tiles16 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 15) / 16);
    static const int force_nb = [] { const char *e = getenv("SR355_NARROW_NB"); return e ? atoi(e) : 0; }();
    if (force_nb == 4 || (force_nb == 0 && tiles16 >= cu)) return launch_fine_inst<1, 4, 2>(b, stream);
    return launch_fine_inst<1, 2, 3>(b, stream);
}


// Small images (a launch of 4-row tiles does not fill the chip): one tile per workgroup, a three-deep ring, 32
// output channels per workgroup so that a 64-channel conv spreads over twice the workgroups.
int launch_conv3x3_fine_small(const ConvArgs &a, hipStream_t stream) {
    ConvArgs b = a;
    b.trace = g_conv_trace;
    return launch_fine_inst<1, 1, 3>(b, stream);   // ring depth measured: 2 -> 11.3 ms, 3 -> 10.3, 4 -> 10.4, 8 -> 11.0 per pixel iteration
}

}  // namespace sr355
