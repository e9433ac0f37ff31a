// A chain of 3x3 convs over one image geometry as ONE persistent launch: the body of the generator
// (/root/reference/esrgan/models/esrgan_model.py:18-24, :35-39, :57: 69 dense blocks of five convs and trunk_conv)
// without a kernel boundary between its 346 convs.
//
// Launched one after the other, every conv pays a dependent-kernel boundary (with the write-back of the 17 MB it left
// dirty), a prologue in which every workgroup waits for its first tile to arrive, and an epilogue in which all 512
// workgroups store at once while the matrix pipe idles: 15-18 us around an MFMA loop of 6-16 us.  Here a workgroup
// keeps its 16 x 32 pixel tile through the whole chain.  A "pass" is one conv restricted to 32 output channels (conv5
// and trunk_conv are two passes).  Passes run in chain order; the stage ring (conv3x3_fine's: LDS-DMA, two slots,
// MFMAs interleaved with the fragment reads and the DMA issue) runs across pass boundaries, so the stores of pass p
// drain under the first stages of pass p + 1, which read older channels only.
//
// What replaces the kernel boundary is a per-tile counter of completed passes:
//   * producer: the epilogue stores are write-through (sc1); at the next stage boundary every wave has drained them
//     (s_waitcnt vmcnt(0), then the workgroup barrier) and one lane raises the tile's counter with an agent-scope
//     store;
//   * consumer: before the first stage that reads the previous pass's 32 channels is requested (the last two stages
//     of a growth conv; every stage of a block's first conv), wave 0 polls the counters of the eight neighbouring
//     tiles (relaxed, agent scope), then runs ONE agent-scope acquire, which drops this CU's L1 lines; the stage
//     boundary's wait + barrier hold the other waves until it has completed.  All later loads are LDS-DMA or plain
//     loads behind that acquire.
// Older channels need no wait: pass p - 1 on a neighbouring tile finished only after pass p - 2 had finished on all of
// ITS neighbours, this tile's neighbourhood included.  The same argument orders every write after the reads of the
// data it replaces.  Workgroups never wait for a later pass, so the waits cannot form a cycle; every workgroup of the
// grid must be resident (the launcher sizes the grid to the occupancy it queries), and a poll gives up after two
// seconds and raises an error word instead of hanging.
#include <stdlib.h>

#include <mutex>

#include "conv_epilogue.h"
#include "sr355_internal.h"

namespace sr355 {

// Timing builds (SR355_TRACE): wave 0 adds up where its wall-clock time goes (100 MHz ticks), eight words per workgroup.
#ifdef SR355_TRACE
#define CHAIN_T0(V) const unsigned long long V = c.trace ? wall_clock64() : 0ull
#define CHAIN_ADD(K, V) if (c.trace) tr_[K] += wall_clock64() - (V)
#else
#define CHAIN_T0(V)
#define CHAIN_ADD(K, V)
#endif

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int TWP = 34;          // tile width 32 + halo
constexpr int RSU = 2;           // units per stage
constexpr int OOB = 0x7fffff00;  // byte offset beyond any descriptor range -> DMA writes zeros

struct ChainGeo {
    static constexpr int WAVES = 4, NB = CHAIN_NB, CO = 32;
    static constexpr int TH = WAVES * NB;                      // 16 rows
    static constexpr int NPIX = (TH + 2) * TWP;
    static constexpr int NIN = RSU * NPIX;
    static constexpr int NIN_I = (NIN + 63) / 64;
    static constexpr int NWT = RSU * 9 * CO;
    static constexpr int NWT_I = NWT / 64;
    static constexpr int JTOT = NIN_I + NWT_I;
    static constexpr int KMAX = (JTOT + WAVES - 1) / WAVES;    // DMA instructions per wave and stage
    static constexpr int W_OFF = NIN_I * 1024;
    static constexpr int STAGE_BYTES = JTOT * 1024;
    static constexpr int BIAS_OFF = 2 * STAGE_BYTES;           // two bias images: the pass in its epilogue, the next one
    static constexpr int CUR_OFF = BIAS_OFF + 512;             // two cursor records (64 bytes used of 256 each)
    static constexpr int DUMMY_OFF = CUR_OFF + 512;            // landing area of the padding loads
    static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
    static constexpr int N_READS = 3 + NB + 2;                 // LDS reads of one tap column
    static constexpr int N_MFMA = 3 * NB;                      // MFMAs of one tap column
    static_assert(NWT % 64 == 0, "weight image must be whole wave-instructions");
    static_assert(CHAIN_WG_PER_CU * LDS_BYTES <= 160 * 1024, "workgroups per CU");
    static_assert(2 * N_MFMA >= N_READS && N_MFMA >= KMAX, "not enough MFMAs to hide the reads / the DMA issue");
};
typedef ChainGeo G;

// The tile epilogue of conv_epilogue.h (same arithmetic in the same order: bias, LeakyReLU, the skip read from the bf16
// activations, the two fp32 skips) for a launch whose workgroups read each other's output:
//   * stores are write-through (sc1) and 16 bytes per lane: the two half-waves hold the two halves of every 8-channel
//     unit and swap them, so that a wave instruction writes whole 128-byte lines (an 8-byte write-through store is one
//     fabric write per lane);
//   * every skip load of the tile is issued before the first store.  Loads and stores retire through one in-order
//     counter: a load issued behind a store cannot be waited for without waiting for that store, and a write-through
//     store takes microseconds.
// A chain pass has no sign mask and no NCHW / uint8 output (chain_supported).
// The 16-byte write-through store goes out as inline asm, and the compiler's hazard recogniser does not look inside an
// asm statement.  Two of the architecture's manual wait-state rules apply to it and are part of the statement:
//   * a scalar register written by a vector instruction (v_readlane reloading a spilled SGPR, v_readfirstlane) must not
//     be read by a memory instruction within five wait states, and the store's descriptor and scalar offset are such
//     registers (a build whose spill reload landed directly in front of the store wrote to stale addresses whenever
//     the wave issued back to back);
//   * the data registers of a store of more than 64 bits must not be overwritten by a vector instruction within two
//     wait states of its issue: the next quad's arithmetic reuses them, and without wait states here the kernel stores
//     wrong values (tests/test_gpu_chain.py fails).  Four are inserted.
typedef int chain_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void chain_store16(const epi_v4u &data, chain_v4i rsrc, int voffset, int soffset) {
#if defined(SR355_EXP_NOSTORE)      /* timing experiments only: wrong results */
    asm volatile("s_nop 4\n\ts_nop 3" : : "v"(data), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
#elif defined(SR355_EXP_NTSTORE)    /* timing experiment: write-through and non-temporal */
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen sc1 nt\n\ts_nop 3"
                 :
                 : "v"(data), "v"(voffset), "s"(rsrc), "s"(soffset)
                 : "memory");
#elif defined(SR355_EXP_PLAINSTORE)
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3"
                 :
                 : "v"(data), "v"(voffset), "s"(rsrc), "s"(soffset)
                 : "memory");
#else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 3"
                 :
                 : "v"(data), "v"(voffset), "s"(rsrc), "s"(soffset)
                 : "memory");
#endif
}

// FIRST: the first skip is absent (0), the bf16 activations (1) or an fp32 carrier (2); SECOND: a.res2 is present;
// OUTF32: the fp32 carrier a.out_f32 is written; F16: a.out_act is stored as fp16.  a.out_act always exists.
// Compile-time, so that only the loads a pass has occupy registers and no quad branches on what the pass writes.
template <int FIRST, bool SECOND, bool OUTF32, bool F16>
__device__ __forceinline__ void chain_epilogue_as(const ConvArgs &a, f32x16 (&acc)[1][G::NB], const unsigned char *bias_lds,
                                                  int n, int y0, int x, int cb_co, int h) {
    constexpr int NB = G::NB;
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
    auto rsrc_words = [&](const void *base, size_t img_bytes) {     // the same descriptor as four scalar words
        const unsigned long long p = (unsigned long long)base + (size_t)n * img_bytes;
        chain_v4i w;
        w[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        w[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) & 0xffff;
        w[2] = __builtin_amdgcn_readfirstlane((int)img_bytes);
        w[3] = 0x00020000;
        return w;
    };
    const chain_v4i rw_act = rsrc_words(a.out_act, (size_t)H * W * a.out_units * 16);
    const __amdgpu_buffer_rsrc_t rs_ra = rsrc(a.res_act, (size_t)H * W * a.res_act_units * 16);
    const __amdgpu_buffer_rsrc_t rs_r1 = rsrc(a.res1, f32_img);
    const __amdgpu_buffer_rsrc_t rs_r2 = rsrc(a.res2, f32_img);
    const chain_v4i rw_f32 = rsrc_words(a.out_f32, f32_img);
    const float slope = a.slope;
    constexpr bool has_ra = FIRST == 1, has_r1 = FIRST == 2, has_r2 = SECOND;
    int vo_act16[NB], vo_ra[NB], vo_f[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int y = y0 + nb;
        const bool ok = x < W && y < H;
        vo_act16[nb] = ok ? ((y * a.out_units + a.out_unit_off + (cb_co >> 3) + h) * W + x) * 16 : OOB;
        vo_ra[nb] = ok ? ((y * a.res_act_units + (cb_co >> 3)) * W + x) * 16 + h * 8 : OOB;
        vo_f[nb] = ok ? ((y * a.f32_units + (cb_co >> 2) + h) * W + x) * 16 : OOB;
    }
    // The bias image landed stages ago (its DMA was issued when the cursor entered this pass).  Read through inline asm:
    // the compiler puts a wait for ALL outstanding DMA (s_waitcnt vmcnt(0)) in front of any LDS read it can see, which
    // here would stall the epilogue on the next pass's first stage, requested a moment ago.
    float4 bv[4];
    {
        const unsigned b = (unsigned)(size_t)(bias_lds + h * 16);
        epi_v4f t0, t1, t2, t3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                     "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(b)
                     : "memory");
        bv[0] = make_float4(t0[0], t0[1], t0[2], t0[3]), bv[1] = make_float4(t1[0], t1[1], t1[2], t1[3]);
        bv[2] = make_float4(t2[0], t2[1], t2[2], t2[3]), bv[3] = make_float4(t3[0], t3[1], t3[2], t3[3]);
    }
    // Rows are handled in groups whose skip loads fit the register file beside the accumulators: all four rows at
    // once, or two and two when both skips are present (the second group's loads then wait for the first group's
    // stores: one write-through round trip per pass instead of one per row).
    constexpr int NBG = (has_r2 && FIRST != 0) ? (NB % 2 == 0 ? 2 : 1) : NB;
#pragma unroll
    for (int g0 = 0; g0 < NB; g0 += NBG) {
        epi_v2u ra[has_ra ? NBG : 1][4];
        epi_v4u r1[has_r1 ? NBG : 1][4], r2[has_r2 ? NBG : 1][4];
        if constexpr (has_ra) {
#pragma unroll
            for (int i = 0; i < NBG; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) ra[i][q] = __builtin_amdgcn_raw_buffer_load_b64(rs_ra, vo_ra[g0 + i], q * wstep, 0);
        } else if constexpr (has_r1) {
#pragma unroll
            for (int i = 0; i < NBG; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r1[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, vo_f[g0 + i], q * 2 * wstep, 0);
        }
        if constexpr (has_r2) {
#pragma unroll
            for (int i = 0; i < NBG; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r2[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, vo_f[g0 + i], q * 2 * wstep, 0);
        }
#pragma unroll
        for (int i = 0; i < NBG; ++i) {
            const int nb = g0 + i;
            epi_v2u o_even = {0u, 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[0][nb][4 * q + 0] + bv[q].x, acc[0][nb][4 * q + 1] + bv[q].y,
                              acc[0][nb][4 * q + 2] + bv[q].z, acc[0][nb][4 * q + 3] + bv[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][nb][4 * q + e] = 0.f;
                // LeakyReLU as max(v, slope * v): the same value as the compare-and-select of the other epilogues for
                // a slope in [0, 1] (the launcher checks it), one instruction per element and no condition register
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaxf(v[j], v[j] * slope);
                if constexpr (has_ra) {
                    const epi_bf16x4 rv = __builtin_bit_cast(epi_bf16x4, ra[i][q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res1_w, (float)rv[j], a.res1_scale, v[j]);
                } else if constexpr (has_r1) {
                    const epi_v4f rf = __builtin_bit_cast(epi_v4f, r1[i][q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res1_w, rf[j], a.res1_scale, v[j]);
                }
                if constexpr (has_r2) {
                    const epi_v4f rf = __builtin_bit_cast(epi_v4f, r2[i][q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = epi_axpby(a.res2_w, rf[j], a.res2_scale, v[j]);
                }
                if constexpr (OUTF32) {
                    const epi_v4f of = {v[0], v[1], v[2], v[3]};
                    chain_store16(__builtin_bit_cast(epi_v4u, of), rw_f32, vo_f[nb], q * 2 * wstep);
                }
                {
                    epi_v2u o;
                    if constexpr (F16) {
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
                    if ((q & 1) == 0) {
                        o_even = o;
                    } else {   // lanes 0-31 end up with unit q - 1, lanes 32-63 with unit q
                        const auto s0 = __builtin_amdgcn_permlane32_swap(o_even[0], o[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(o_even[1], o[1], false, false);
                        const epi_v4u unit = {s0[0], s1[0], s0[1], s1[1]};
                        chain_store16(unit, rw_act, vo_act16[nb], (q - 1) * wstep);
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void chain_epilogue(const ConvArgs &a, f32x16 (&acc)[1][G::NB], const unsigned char *bias_lds,
                                               int n, int y0, int x, int cb_co, int h) {
    // the combinations the generator's body has; anything else takes the shared tile epilogue
    const int first = a.res_act ? 1 : (a.res1 ? 2 : 0);
    const bool second = a.res2 != nullptr, outf32 = a.out_f32 != nullptr, f16 = a.f16_out != 0;
#define CHAIN_EPI(FI, SE, OF, FH)                                                                     \
    if (first == FI && second == SE && outf32 == OF && f16 == FH)                                     \
        return chain_epilogue_as<FI, SE, OF, FH>(a, acc, bias_lds, n, y0, x, cb_co, h)
    CHAIN_EPI(0, false, false, false);   // growth convs
    CHAIN_EPI(1, false, false, false);   // conv5, dense skip from the bf16 activations
    CHAIN_EPI(1, true, true, false);     // conv5 of an RRDB's last block: + RRDB skip, fp32 carrier out
    CHAIN_EPI(2, false, true, false);    // conv5, every skip on an fp32 carrier
    CHAIN_EPI(2, true, true, false);
    CHAIN_EPI(2, false, false, false);   // trunk_conv + global skip
    CHAIN_EPI(2, false, false, true);    // the same, feeding an fp16 conv
#undef CHAIN_EPI
    conv_epilogue_tile<1, G::NB, 16>(a, acc, bias_lds, n, y0, x, cb_co, h, true);
}
}  // namespace

__global__ __launch_bounds__(256, CHAIN_WG_PER_CU) void conv3x3_chain(const ChainArgs c) {
    constexpr int WAVES = G::WAVES, NB = G::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(4))) const ChainPass cpass_t;   // read through the scalar cache: the table is
    cpass_t *passes = (cpass_t *)c.passes;                               // written by the host before the launch only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int H = c.H, W = c.W;
    const int img_bytes = H * c.in_units * W * 16;
    const int in_stage_stride = RSU * W * 16;

    // Blocks b and b+8 share an XCD: give the blocks of one XCD consecutive tiles of every round.
    const int n_tiles = c.N * c.tiles_y * c.tiles_x;
    const int nwg = gridDim.x;
    int first;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        first = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    // The dispatcher fills every CU once and then a second time: blocks b and b + nwg/2 share a CU, and its
    // instruction arbiter serves the older wave first, so the first half of the grid would run ahead and leave the
    // second half to finish alone on half-empty CUs.  The blocks of the second half therefore run at wave priority 1
    // and those of the first half alternate between 2 and 0, pass by pass: one of the two blocks of a CU always runs
    // at full speed, the other fills its gaps, and both advance at about the same rate.  (Where the pairing does not
    // hold the priorities are merely arbitrary.)
    const bool prio_alternates = (int)blockIdx.x < (nwg >> 1);
    const int prio_class = CHAIN_WG_PER_CU == 3 ? (int)blockIdx.x / ((nwg + 2) / 3) : 0;   // (three per CU: rotation)
    if (CHAIN_WG_PER_CU == 2 && !prio_alternates) __builtin_amdgcn_s_setprio(1);
    const int n_my = first < n_tiles ? (n_tiles - first + nwg - 1) / nwg : 0;
    const int n_ord = c.n_passes * n_my;          // this workgroup's tasks: pass-major, its tiles within a pass
    if (n_ord == 0) return;

#define SR355_RSRC(LO, HI, LEN)                                                                       \
    __builtin_amdgcn_make_buffer_rsrc(                                                                \
        (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(HI) << 32) |           \
                 (unsigned)__builtin_amdgcn_readfirstlane(LO)),                                       \
        0, __builtin_amdgcn_readfirstlane(LEN), 0x00020000)

    // ---- issue cursor: the stage the DMA requests next ----
    int voff[G::KMAX];
    int in_lo = 0, in_hi = 0, w_lo = 0, w_hi = 0, w_len = 0, w_stage_stride = 0;
    const int in_len = __builtin_amdgcn_readfirstlane(img_bytes);
    int it_ord = 0, it_stage = 0, it_S = 0, it_dep = 0, it_need = 0, it_dep2 = 0, it_need2 = 0, it_p = -1;
    int it_tx = 0, it_ty = 0, it_n = 0, it_i = -1;
    unsigned it_perm_lo = 0x76543210u, it_perm_hi = 0xfedcba98u;   // stage order of the cursor's pass (ChainPass::perm)

    int cur_in_lo = 0, cur_in_hi = 0, cur_b_lo = 0, cur_b_hi = 0, cur_cout_pad = 32;
    // the 64-byte cursor record of pass P into LDS slot P & 1 (wave 0; lanes 16.. read beyond the range: zeros)
#define SR355_CURSOR_DMA(P)                                                                           \
    {                                                                                                 \
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(                      \
            const_cast<ChainPass *>(c.passes) + (P), 0, (int)sizeof(ChainCursor), 0x00020000);        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c_rsrc, (lds_void *)(smem + G::CUR_OFF + ((P)&1) * 256), 4, lane * 4, \
                                                 0, 0, 0);                                            \
    }
    // Set the cursor up for task ORD: descriptor of its pass, tile coordinates, per-lane source offsets, bias image.
#define SR355_TASK_SETUP(ORD)                                                                         \
    {                                                                                                 \
        const int p_ = (ORD) / n_my;                                                                  \
        const int i_ = (ORD)-p_ * n_my;                                                               \
        if (p_ != it_p) {                                                                             \
            /* the pass's cursor record, brought into the LDS when the cursor entered the pass before (landed: a   */ \
            /* stage boundary lies in between); then the request for the next pass's record                        */ \
            chain_v4i r0_, r1_, r2_, r3_;                                                             \
            const unsigned ra_ = (unsigned)(size_t)(smem + G::CUR_OFF + (p_ & 1) * 256);              \
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t" \
                         "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"                      \
                         : "=&v"(r0_), "=&v"(r1_), "=&v"(r2_), "=&v"(r3_)                             \
                         : "v"(ra_)                                                                   \
                         : "memory");                                                                 \
            it_perm_lo = (unsigned)__builtin_amdgcn_readfirstlane(r3_[0]);                            \
            it_perm_hi = (unsigned)__builtin_amdgcn_readfirstlane(r3_[1]);                            \
            w_lo = __builtin_amdgcn_readfirstlane(r0_[0]), w_hi = __builtin_amdgcn_readfirstlane(r0_[1]); \
            cur_in_lo = __builtin_amdgcn_readfirstlane(r0_[2]), cur_in_hi = __builtin_amdgcn_readfirstlane(r0_[3]); \
            cur_b_lo = __builtin_amdgcn_readfirstlane(r1_[0]), cur_b_hi = __builtin_amdgcn_readfirstlane(r1_[1]); \
            it_S = __builtin_amdgcn_readfirstlane(r1_[2]);                                            \
            cur_cout_pad = __builtin_amdgcn_readfirstlane(r1_[3]);                                    \
            it_need = __builtin_amdgcn_readfirstlane(r2_[0]), it_dep = __builtin_amdgcn_readfirstlane(r2_[1]); \
            it_need2 = __builtin_amdgcn_readfirstlane(r2_[2]), it_dep2 = __builtin_amdgcn_readfirstlane(r2_[3]); \
            it_p = p_;                                                                                \
            w_stage_stride = RSU * 9 * cur_cout_pad * 16;                                             \
            w_len = w_stage_stride * it_S;                                                            \
            if (wave == 0 && p_ + 1 < c.n_passes) SR355_CURSOR_DMA(p_ + 1);                           \
        }                                                                                             \
        const int cout_pad_ = cur_cout_pad;                                                           \
        const bool new_tile_ = i_ != it_i || SR355_EXP_SKIP2_V;                                       \
        if (new_tile_) {                                                                              \
            const int t_ = first + i_ * nwg;                                                          \
            it_tx = t_ % c.tiles_x;                                                                   \
            const int tr_ = t_ / c.tiles_x;                                                           \
            it_ty = tr_ % c.tiles_y;                                                                  \
            it_n = tr_ / c.tiles_y;                                                                   \
            it_i = i_;                                                                                \
        }                                                                                             \
        const unsigned long long ib_ = (((unsigned long long)(unsigned)cur_in_hi << 32) | (unsigned)cur_in_lo) + \
                                       (size_t)it_n * img_bytes;                                      \
        in_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)ib_);                                   \
        in_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib_ >> 32));                           \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) {                                         \
            const int j = k * WAVES + wave;                                                           \
            if (j < G::NIN_I) {                                                                       \
                if (new_tile_) {                                                                      \
                    const int i = j * 64 + lane;                                                      \
                    const int u = i / G::NPIX;                                                        \
                    const int p = i - u * G::NPIX;                                                    \
                    const int ty = p / TWP;                                                           \
                    const int tx = p - ty * TWP;                                                      \
                    const int y = it_ty * G::TH - 1 + ty, x = it_tx * 32 - 1 + tx;                    \
                    const bool valid = (i < G::NIN) && ((unsigned)y < (unsigned)H) && ((unsigned)x < (unsigned)W); \
                    voff[k] = valid ? ((y * c.in_units + u) * W + x) * 16 : OOB;                      \
                }                                                                                     \
                if (SR355_EXP_SKIP2_V && p_ % 6 == 5) voff[k] = OOB;                                  \
            } else if (j < G::JTOT) {                                                                 \
                const int jw = (j - G::NIN_I) * 64 + lane;                                            \
                const int u = jw / (9 * G::CO);                                                       \
                const int rem = jw - u * 9 * G::CO;                                                   \
                const int tap = rem / G::CO;                                                          \
                const int co = rem - tap * G::CO;                                                     \
                voff[k] = ((u * 9 + tap) * cout_pad_ + co) * 16;                                      \
            } else {                                                                                  \
                voff[k] = OOB;                                                                        \
            }                                                                                         \
        }                                                                                             \
        if (wave == 0) {                                                                              \
            const __amdgpu_buffer_rsrc_t b_rsrc = SR355_RSRC(cur_b_lo, cur_b_hi, G::CO * 4);          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void *)(smem + G::BIAS_OFF + ((ORD)&1) * 256), 4,  \
                                                     lane * 4, 0, 0, 0);                              \
        }                                                                                             \
    }

    // One stage of DMA into ring slot SLOT, without a branch (see conv3x3_fine): with LIVE false the same
    // instructions go out with out-of-range sources (zeros into a slot nobody reads).
#define SR355_ISSUE_DMA(SLOT, LIVE)                                                                   \
    {                                                                                                 \
        const bool live_ = (LIVE);                                                                    \
        const int ps_ = (int)(((it_stage < 8 ? it_perm_lo : it_perm_hi) >> ((it_stage & 7) * 4)) & 15u); \
        const int soff_in_ = __builtin_amdgcn_readfirstlane(ps_ * in_stage_stride);                   \
        const int soff_w_ = __builtin_amdgcn_readfirstlane(ps_ * w_stage_stride);                     \
        const int dst0_ = (SLOT)*G::STAGE_BYTES + wave * 1024;                                        \
        _Pragma("unroll") for (int k = 0; k < G::KMAX; ++k) {                                         \
            const int j = k * WAVES + wave;                                                           \
            const bool all_in_ = k * WAVES + WAVES - 1 < G::NIN_I;                                    \
            const bool all_w_ = k * WAVES >= G::NIN_I && k * WAVES + WAVES - 1 < G::JTOT;             \
            const bool is_in_ = all_in_ || (!all_w_ && j < G::NIN_I);                                 \
            const bool exists_ = all_in_ || all_w_ || j < G::JTOT;                                    \
            const int lo_ = is_in_ ? in_lo : w_lo, hi_ = is_in_ ? in_hi : w_hi;                       \
            const int len_ = is_in_ ? in_len : w_len, so_ = is_in_ ? soff_in_ : soff_w_;              \
            const int dst_ = exists_ ? dst0_ + k * WAVES * 1024 : G::DUMMY_OFF;                       \
            const int vo_ = (live_ && exists_ && !SR355_EXP_NOLOAD_V) ? SR355_EXP_L2WRAP(voff[k], is_in_) : OOB;                \
            if (all_in_ && SR355_EXP_IN_AUX != 0)      /* (folds after unrolling) */                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(SR355_RSRC(lo_, hi_, len_), (lds_void *)(smem + dst_), 16,  \
                                                         vo_, so_, 0, SR355_EXP_IN_AUX);              \
            else                                                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(SR355_RSRC(lo_, hi_, len_), (lds_void *)(smem + dst_), 16,  \
                                                         vo_, so_, 0, 0);                             \
        }                                                                                             \
    }

    // Wave 0 waits until NEED passes are complete on the neighbours of the cursor's tile, then acquires.
#define SR355_AWAIT_NEIGHBOURS(NEED)                                                                  \
    if (wave == 0) {                                                                                  \
        const int dy_ = lane / 3 - 1, dx_ = lane % 3 - 1;                                             \
        const int ty_ = it_ty + dy_, tx_ = it_tx + dx_;                                               \
        const bool polls_ = lane < 9 && lane != 4 && (unsigned)ty_ < (unsigned)c.tiles_y &&           \
                            (unsigned)tx_ < (unsigned)c.tiles_x;                                      \
        gu32 *f_ = (gu32 *)c.flags + ((size_t)it_n * c.tiles_y + (polls_ ? ty_ : it_ty)) * c.tiles_x + \
                   (polls_ ? tx_ : it_tx);                                                            \
        const unsigned long long t0_ = wall_clock64();                                                \
        while (!gave_up && !SR355_EXP_NOPOLL_V) {                                                     \
            const unsigned v_ = polls_ ? __hip_atomic_load(f_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; \
            if (__all(!polls_ || v_ >= (unsigned)(NEED))) break;                                      \
            __builtin_amdgcn_s_sleep(2);                                                              \
            if (wall_clock64() - t0_ > 200000000ull) {   /* 2 s at 100 MHz: a peer is not resident */ \
                if (lane == 0) __hip_atomic_store((gu32 *)c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
                gave_up = true;     /* no further waits in this workgroup: the output is invalid anyway */ \
            }                                                                                         \
        }                                                                                             \
        asm volatile("; chain acquire" ::: "memory");                                                 \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                            \
    }

    bool gave_up = false;
#ifdef SR355_TRACE
    unsigned long long tr_[12] = {};
    const unsigned long long tr_start_ = wall_clock64();
    const unsigned long long tr_cycles_ = clock64();
#endif
    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][nb][e] = 0.f;

    const int lane_in = (h * G::NPIX + wave * NB * TWP + r) * 16;
    const int lane_w = G::W_OFF + (h * 9 * G::CO + r) * 16;

#define SR355_LOAD_FRAGS(WF, XF, KX)                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                              \
            WF[ky] = *reinterpret_cast<const bf16x8 *>(pw + ((ky * 3 + (KX)) * G::CO) * 16);          \
        _Pragma("unroll") for (int rho = 0; rho < NB + 2; ++rho)                                      \
            if (SR355_EXP_KEEP(KX)) XF[rho] = *reinterpret_cast<const bf16x8 *>(pin + (rho * TWP + (KX)) * 16); \
            else XF[rho] = exp_prev_[rho];                                                            \
        _Pragma("unroll") for (int rho = 0; rho < NB + 2; ++rho) exp_prev_[rho] = XF[rho];            \
    }
#ifndef SR355_EXP_IN_AUX   /* cache-policy bits of the activation loads (timing experiments: 2 = nt, 1 = sc0, 16 = sc1) */
#define SR355_EXP_IN_AUX 0
#endif
#ifdef SR355_EXP_NOLOAD   /* timing experiments only: every DMA reads out of range (zeros, no fetch) */
#define SR355_EXP_NOLOAD_V true
#else
#define SR355_EXP_NOLOAD_V false
#endif
#ifdef SR355_EXP_L2LOAD   /* timing experiments only: every activation load wraps into the first 128 KB of its image (L2 hits) */
#define SR355_EXP_L2WRAP(V, IS_IN) (((IS_IN) && (V) != OOB) ? ((V)&0x1fff0) : (V))
#else
#define SR355_EXP_L2WRAP(V, IS_IN) (V)
#endif
#ifdef SR355_EXP_SKIP2    /* timing experiments only: the second pass of a conv5 fetches no activations */
#define SR355_EXP_SKIP2_V true
#else
#define SR355_EXP_SKIP2_V false
#endif
#ifdef SR355_EXP_NOPOLL   /* timing experiments only: wrong results */
#define SR355_EXP_NOPOLL_V true
#else
#define SR355_EXP_NOPOLL_V false
#endif
#ifdef SR355_EXP_READS
#define SR355_EXP_KEEP(KX) ((KX) == 0)
#else
#define SR355_EXP_KEEP(KX) true
#endif
#define SR355_MMA_FRAGS(WF, XF)                                                                       \
    {                                                                                                 \
        _Pragma("unroll") for (int rho = 0; rho < NB + 2; ++rho)                                      \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                        \
                const int nb = rho - ky;                                                              \
                if (nb >= 0 && nb < NB)                                                               \
                    acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ky], XF[rho], acc[0][nb], 0, 0, 0); \
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

    // A cursor that stands in front of a pass whose first stage needs the pass this workgroup is still computing
    // cannot run ahead of the workgroup's own epilogue: it is held there and restarted after the epilogue.
#define SR355_CURSOR_HELD(CUR_P) (it_ord < n_ord && it_stage == 0 && it_dep == 0 && it_need > (CUR_P))

    // ---- prologue: first task, first stage ----
    if (wave == 0) SR355_CURSOR_DMA(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SR355_TASK_SETUP(0);
    if (SR355_CURSOR_HELD(-1) && it_need > 0) {   // (a chain whose first pass reads another launch's output has need = 0)
        SR355_AWAIT_NEIGHBOURS(it_need);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    SR355_ISSUE_DMA(0, true);
    ++it_stage;    // (every pass has at least two stages)

    int slot = 0, issue_slot = 1;
    int pub_tile = -1;
    unsigned pub_val = 0;
    int ct_i = -1, ct_tx = 0, ct_ty = 0, ct_n = 0;
    for (int ord = 0; ord < n_ord; ++ord) {
        const int p = ord / n_my;
        if (CHAIN_WG_PER_CU == 3) {
            const int pr = (p + prio_class) % 3;
            if (pr == 0) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        } else if (prio_alternates) {   // (measured: per pass beats per stage and per two passes; this parity balances the halves)
            if (p & 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        }
        {
            const int i = ord - p * n_my;
            if (i != ct_i) {
                const int t = first + i * nwg;
                ct_tx = t % c.tiles_x;
                const int tr = t / c.tiles_x;
                ct_ty = tr % c.tiles_y;
                ct_n = tr / c.tiles_y;
                ct_i = i;
            }
        }
        // the pass descriptor, fetched now: its scalar-cache misses hide behind the stages
        CHAIN_T0(td_);
        ConvArgs a;
        {
            typedef __attribute__((address_space(4))) const unsigned cword_t;
            cword_t *src = (cword_t *)&passes[p].a;
            static_assert(sizeof(ConvArgs) % 4 == 0, "descriptor is whole words");
            unsigned words[sizeof(ConvArgs) / 4];
#pragma unroll
            for (unsigned i = 0; i < sizeof(ConvArgs) / 4; ++i) words[i] = src[i];
            __builtin_memcpy(&a, words, sizeof(a));     // (not a cast: the fields are floats and pointers)
        }
        const int cb_co = passes[p].cb_co;
        const int S = a.n_stages;
        CHAIN_ADD(9, td_);
        for (int s = 0; s < S; ++s) {
            // the stage the cursor requests in this iteration is the first one that reads the previous pass's output
            // on other tiles: wait for the neighbours and acquire before the boundary below
            const bool live = it_ord < n_ord && !SR355_CURSOR_HELD(p);
            const bool dep1 = it_stage == it_dep && it_need > 0, dep2 = it_stage == it_dep2 && it_need2 > 0;
            if (live && (dep1 || dep2)) {
                CHAIN_T0(tp_);
                const int need_now = dep2 ? it_need2 : it_need;
                SR355_AWAIT_NEIGHBOURS(need_now);
                CHAIN_ADD(0, tp_);
                CHAIN_T0(tq_);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CHAIN_ADD(5, tq_);
            }
            // the requested stage has landed, this wave's stores of the previous epilogue have drained and the
            // acquire, if any, has completed; after the barrier that holds for every wave
            CHAIN_T0(tw_);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CHAIN_ADD(s == 0 ? 6 : 1, tw_);
            CHAIN_T0(tb_);
            __builtin_amdgcn_s_barrier();
            CHAIN_ADD(2, tb_);
            if (pub_tile >= 0) {
                if (tid == 0)
                    __hip_atomic_store((gu32 *)c.flags + pub_tile, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pub_tile = -1;
            }
            const unsigned char *buf = smem + slot * G::STAGE_BYTES;
            const unsigned char *pin = buf + lane_in;
            const unsigned char *pw = buf + lane_w;
            bf16x8 wfA[3], xfA[NB + 2], wfB[3], xfB[NB + 2], exp_prev_[NB + 2];
            SR355_LOAD_FRAGS(wfA, xfA, 0);
            __builtin_amdgcn_sched_barrier(0);
            // column 0: its MFMAs, the reads of column 1 and this stage's DMA issue, one of each at a time
            SR355_ISSUE_DMA(issue_slot, live);
            SR355_LOAD_FRAGS(wfB, xfB, 1);
            SR355_MMA_FRAGS(wfA, xfA);
            SR355_INTERLEAVE(G::N_READS, G::KMAX, 0);
            __builtin_amdgcn_sched_barrier(0);
            // column 1 under the reads of column 2
            SR355_LOAD_FRAGS(wfA, xfA, 2);
            SR355_MMA_FRAGS(wfB, xfB);
            SR355_INTERLEAVE(G::N_READS, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
#ifndef SR355_EXP_NOMFMA2   /* timing experiments only: a third of the MFMAs dropped */
            SR355_MMA_FRAGS(wfA, xfA);
#endif
            __builtin_amdgcn_sched_barrier(0);
            issue_slot ^= 1;
            slot ^= 1;
            // advance the cursor (branches, kept out of the MFMA blocks)
            if (live && ++it_stage == it_S) {
                CHAIN_T0(ta_);
                it_stage = 0;
                if (++it_ord < n_ord) SR355_TASK_SETUP(it_ord);
                CHAIN_ADD(8, ta_);
            }
        }
        // ---- epilogue of task ord: write-through stores, published at the next stage boundary ----
        CHAIN_T0(te_);
#ifdef SR355_EXP_NOEPI   /* timing experiments only: wrong results */
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) _Pragma("unroll") for (int e = 0; e < 16; ++e) {
            asm volatile("" ::"v"(acc[0][nb][e]));
            acc[0][nb][e] = 0.f;
        }
#else
        {
            chain_epilogue(a, acc, smem + G::BIAS_OFF + (ord & 1) * 256, ct_n, ct_ty * G::TH + wave * NB, ct_tx * 32 + r,
                           cb_co, h);
        }
#endif
        CHAIN_ADD(3, te_);
        pub_tile = (ct_n * c.tiles_y + ct_ty) * c.tiles_x + ct_tx;
        pub_val = (unsigned)(p + 1);
        if (SR355_CURSOR_HELD(p) && it_ord == ord + 1) {
            CHAIN_T0(tr0_);
            // restart: publish this pass, wait for the neighbours, then request the next pass's first stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0)
                __hip_atomic_store((gu32 *)c.flags + pub_tile, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pub_tile = -1;
            SR355_AWAIT_NEIGHBOURS(it_need);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            SR355_ISSUE_DMA(slot, true);      // the slot the next iteration reads
            ++it_stage;
            CHAIN_ADD(4, tr0_);
        }
    }
    // the last pass's stores and the padding DMA drain before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SR355_TRACE
    if (c.trace && tid == 0) {
        tr_[7] = wall_clock64() - tr_start_;
        tr_[10] = clock64() - tr_cycles_;      // shader-clock cycles over the same interval as tr_[7]
        tr_[11] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |   // XCC_ID
                  (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                        // HW_ID
        for (int k = 0; k < 12; ++k) c.trace[blockIdx.x * 12 + k] = tr_[k];
    }
#endif
#undef SR355_RSRC
#undef SR355_CURSOR_DMA
#undef SR355_TASK_SETUP
#undef SR355_ISSUE_DMA
#undef SR355_AWAIT_NEIGHBOURS
#undef SR355_LOAD_FRAGS
#undef SR355_MMA_FRAGS
#undef SR355_INTERLEAVE
#undef SR355_CURSOR_HELD
}

// Workgroups of the chain kernel that are resident at once on the current device (two per CU by its LDS use).
int chain_resident_workgroups(int *out) {
    static std::once_flag once[64];
    static int slots[64] = {};
    static int rc_of[64] = {};
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) {
        set_error("conv3x3 chain: device index %d out of range", dev);
        return SR355_ERR_UNSUPPORTED;
    }
    std::call_once(once[dev], [dev] {
        rc_of[dev] = SR355_ERR_HIP;
        if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_chain),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES),
                      "hipFuncSetAttribute"))
            return;
        int per_cu = 0, cu = 0;
        if (check_hip(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_chain, 256, G::LDS_BYTES),
                      "hipOccupancyMaxActiveBlocksPerMultiprocessor") ||
            check_hip(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute"))
            return;
        if (per_cu > CHAIN_WG_PER_CU) per_cu = CHAIN_WG_PER_CU;
        slots[dev] = per_cu * cu;
        rc_of[dev] = SR355_OK;
    });
    if (rc_of[dev]) return rc_of[dev];
    *out = slots[dev];
    return SR355_OK;
}

bool chain_supported(const ConvArgs &a, int cin_units) {
    return a.ups == 0 && a.zins == 0 && !a.f16_in && a.stride != 2 && (cin_units % RSU) == 0 && cin_units >= 2 * RSU &&
           (a.cout_pad == 32 || a.cout_pad == 64) && a.cout_real == a.cout_pad && !a.out_nchw && !a.out_u8 && a.out_act && !a.mask_act &&
           a.slope >= 0.f && a.slope <= 1.f;
}

int launch_conv3x3_chain(const ChainArgs &c0, int max_workgroups, hipStream_t stream) {
    ChainArgs c = c0;
    c.tiles_x = (c.W + 31) / 32;
    c.tiles_y = (c.H + G::TH - 1) / G::TH;
    const long long n_tiles = (long long)c.N * c.tiles_x * c.tiles_y;
    if (n_tiles <= 0 || n_tiles > 0x7fffffffLL || c.n_passes < 1 || !c.passes || !c.flags || !c.err) {
        set_error("conv3x3 chain: bad arguments (%lld tiles, %d passes)", n_tiles, c.n_passes);
        return SR355_ERR_INVALID;
    }
    int slots = 0;
    if (int rc = chain_resident_workgroups(&slots)) return rc;
    if (max_workgroups > 0 && slots > max_workgroups) slots = max_workgroups;
    if (max_workgroups < 0) slots = -max_workgroups;      // forced grid (may exceed what is resident: the waits then give up)
    if (slots < 1) {
        set_error("conv3x3 chain: no resident workgroup slot");
        return SR355_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)(n_tiles < slots ? n_tiles : slots), 1, 1);
    hipLaunchKernelGGL(conv3x3_chain, grid, dim3(256), G::LDS_BYTES, stream, c);
    return check_hip(hipGetLastError(), "conv3x3 chain launch");
}

}  // namespace sr355
