// One dense block (five 3x3 convs over a growing channel stack) as ONE launch, for images at most 32 pixels wide.
//
// Replaces the five nn.Conv2d + torch.cat + LeakyReLU calls of ResidualDenseBlock.forward
// (/root/reference/esrgan/models/esrgan_model.py:18-24) and, on the transposed weights and the mirrored gradient
// stack, the five data-gradient convs autograd runs for the same lines (/root/reference/esrgan/train.py:161, :220).
//
// At training-patch sizes (16 patches of 32x32) one conv is 16k pixels: a chain of 345 dependent launches of a few
// microseconds each, none of which fills the chip.  Here a workgroup owns a band of two output rows of one image
// and runs the whole block for it:
//   * the block input x (64 channels) of the band plus a 5-row halo is loaded once into LDS (LDS-DMA, out-of-image
//     pixels read as zeros);
//   * growth output k (k = 1..4) is computed on the band plus a (5-k)-row halo and kept in LDS in MFMA operand
//     order: the halo rows are recomputed by the neighbouring bands instead of being exchanged between workgroups
//     (the launches this replaces under-fill the chip several times over, so the recomputation is free);
//   * the packed weights of the five convs stream through a five-slot LDS ring as one flat sequence of 9216-byte
//     chunks (two 8-channel units x 9 taps x 32 outputs; for conv5 one unit x 9 taps x 64 outputs), four chunks
//     in flight behind counted `s_waitcnt vmcnt`;
//   * conv5 ends in the same fused epilogue as the stand-alone kernels (bias, weighted residuals, fp32 carriers,
//     sign mask), the growth outputs of the band's own rows are copied from LDS to the block buffer for the
//     backward pass.
// The accumulation order per output (stage, tap column, tap row) is that of the stand-alone kernels, so the results
// are bit-identical to the five-launch path.
#include <stdlib.h>

#include <mutex>

#include "conv_epilogue.h"
#include "sr355_internal.h"

namespace sr355 {

#ifdef SR355_TRACE
#define DB_STAMP(K)                                                                                           \
    if (a.trace && threadIdx.x == 0) {                                                                        \
        a.trace[blockIdx.x * 32 + (K)] = __builtin_readcyclecounter();                                        \
        a.trace[blockIdx.x * 32 + 16 + (K)] = wall_clock64();                                                 \
    }
#define DB_DEBUG(BIT) (a.debug & (BIT))   // (tools/bench_dense only)
#else
#define DB_STAMP(K)
#define DB_DEBUG(BIT) false
#endif

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int DB_OOB = 0x7fffff00;   // byte offset beyond any descriptor range: the DMA writes zeros
constexpr int DB_TH = 2;             // output rows per workgroup
constexpr int DB_P = 34;             // pixels per LDS row: 32 + one zero column on each side
constexpr int DB_ROWB = DB_P * 16;   // bytes of one 8-channel unit of one row

// resident arrays: k = 0 the block input (8 units), k = 1..4 growth output k (4 units); array k covers the band
// plus a (5-k)-row halo above and below
constexpr int rows_of(int k) { return DB_TH + 2 * (5 - k); }
constexpr int units_of(int k) { return k == 0 ? 8 : 4; }
constexpr int base_of(int k) {
    int b = 0;
    for (int i = 0; i < k; ++i) b += units_of(i) * rows_of(i) * DB_ROWB;
    return b;
}
constexpr int DB_RES_BYTES = base_of(5);
constexpr int DB_CHUNK = 9216;
constexpr int DB_SLOTS = 5;
constexpr int DB_RING_OFF = DB_RES_BYTES;
constexpr int DB_BIAS_OFF = DB_RING_OFF + DB_SLOTS * DB_CHUNK;   // five areas of 256 bytes (one 64-lane DMA each)
constexpr int DB_DUMMY_OFF = DB_BIAS_OFF + 5 * 256;                 // landing area of the padding DMA
constexpr int DB_LDS_BYTES = DB_DUMMY_OFF + 1024;
constexpr int DB_X_ITEMS = units_of(0) * rows_of(0) * DB_P;      // 16-byte items of the input tile
constexpr int DB_X_PIECES = (DB_X_ITEMS + 63) / 64;
constexpr int DB_X_K = (DB_X_PIECES + 3) / 4;                    // DMA instructions per wave for the input tile
static_assert(DB_LDS_BYTES <= 160 * 1024, "dense block image does not fit the LDS");
static_assert(DB_RES_BYTES == 113152, "resident image");

// rows [i0, i0 + nb) of a step with `rows` output rows that wave `wv` of four computes
__device__ __forceinline__ void wave_rows(int rows, int wv, int &i0, int &nb) {
    const int base = rows >> 2, extra = rows & 3;
    i0 = wv * base + (wv < extra ? wv : extra);
    nb = base + (wv < extra ? 1 : 0);
}
}  // namespace

// LDS reads of data that plain stores or long-landed DMA wrote, issued through inline asm: the compiler orders every
// LDS read it can see behind ALL LDS-DMA still in flight (s_waitcnt vmcnt(0)), which would drain the weight ring at
// each use.  The callers guarantee by their own waits and barriers that the bytes read are in place.
__device__ __forceinline__ void db_lds_read16x4(const unsigned char *p, int stride, v4u (&v)[4]) {
    const unsigned b = (unsigned)(size_t)p;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(b), "v"(b + stride), "v"(b + 2 * stride), "v"(b + 3 * stride)
                 : "memory");
}
__device__ __forceinline__ uint4 db_lds_read16(const unsigned char *p) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}

// Issue cursor of the weight ring (every field wave-uniform).
struct DbRing {
    int conv, local, left, slot;   // conv being streamed, next chunk of it, chunks left in it, slot to fill next
    int w_lo, w_hi, w_len;         // base and byte length of that conv's packed weights
};

// One chunk: nine 1 KiB pieces, piece p by wave p % 4; a wave without a third piece sends an out-of-range load to the
// dummy area so that every wave issues three DMA instructions per chunk (the vmcnt arithmetic below counts on it).
// Past the last chunk the same instructions go out with out-of-range sources (zeros into a free slot).
__device__ __forceinline__ void db_issue_chunk(DbRing &g, const DenseArgs &a, unsigned char *smem, int wave,
                                               const int (&pvoff)[3]) {
    if (DB_DEBUG(8)) return;
    const bool live = g.conv < 5 && !DB_DEBUG(1);
    const int so = __builtin_amdgcn_readfirstlane(g.local * DB_CHUNK);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool exists = k < 2 || wave == 0;
        const int dst = exists ? DB_RING_OFF + g.slot * DB_CHUNK + (wave + 4 * k) * 1024 : DB_DUMMY_OFF;
        const int vo = live ? pvoff[k] : DB_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(g.w_hi) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane(g.w_lo)),
                0, __builtin_amdgcn_readfirstlane(g.w_len), 0x00020000),
            (lds_void *)(smem + dst), 16, vo, so, 0, 0);
    }
    g.slot = g.slot + 1 == DB_SLOTS ? 0 : g.slot + 1;
    ++g.local;
    if (live && --g.left == 0) {
        ++g.conv;
        g.local = 0;
        if (g.conv < 5) {
            const unsigned long long p = (unsigned long long)a.w[g.conv];
            g.w_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
            g.w_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
            g.left = g.conv == 4 ? 24 : 4 + 2 * g.conv;
            g.w_len = g.left * DB_CHUNK;
        }
    }
}

// MFMA = 0x008, VMEM read = 0x020, DS read = 0x100: one MFMA, then its share of the LDS reads and of the DMA
// instructions issued in its shadow
#define DB_INTERLEAVE(N_MFMA, N_DS, N_VMEM, SYNC) DB_INTERLEAVE_V(N_MFMA, N_DS, N_VMEM, 0, SYNC)
// ... and N_VALU vector instructions per MFMA (a deferred epilogue's arithmetic; VALU = 0x002, DS write = 0x200)
#define DB_INTERLEAVE_V(N_MFMA, N_DS, N_VMEM, N_VALU, SYNC)                                                   \
    {                                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < (N_MFMA); ++i_) {                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SYNC);                                             \
            if ((N_VALU) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (N_VALU), SYNC);                    \
            _Pragma("unroll") for (int j_ = 0; j_ < ((N_DS) + (N_MFMA)-1) / (N_MFMA); ++j_)                   \
                if (i_ * (((N_DS) + (N_MFMA)-1) / (N_MFMA)) + j_ < (N_DS))                                    \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, SYNC);                                     \
            if (i_ < (N_VMEM)) __builtin_amdgcn_sched_group_barrier(0x020, 1, SYNC);                          \
        }                                                                                                     \
    }

// What the device functions below share.
struct DbCtx {
    const DenseArgs &a;
    unsigned char *smem;
    int tid, wave, r, h, n, y0, img_bytes;
    int pvoff[3];
    float slope, mslope;   // copies of the launch arguments the epilogues use (not re-read from the argument segment)
    int H, W;
};

// Epilogue of one row of growth step J: bias, LeakyReLU (as max(v, slope v): the launcher admits slopes in [0, 1]
// only), sign gate, zero outside the image, bf16 into the step's LDS array.  `row` is the row's index in the array.
template <int J, int Q0, int Q1, bool GATED>
__device__ __forceinline__ void db_growth_quads(const f32x16 &acc, const v2u (&mraw)[4], const v4u (&braw)[4], int row,
                                                const DbCtx &c) {
    const float slope = c.slope, mslope = c.mslope;
    const int y = c.y0 - (5 - J) + row;
    const unsigned keep = (y >= 0 && y < c.H && c.r < c.W) ? 0xffffffffu : 0u;
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const float4 bv = __builtin_bit_cast(float4, braw[q]);
        const float v[4] = {acc[4 * q + 0] + bv.x, acc[4 * q + 1] + bv.y, acc[4 * q + 2] + bv.z, acc[4 * q + 3] + bv.w};
        const bf16x4 mv = __builtin_bit_cast(bf16x4, mraw[q]);
        bf16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = __builtin_fmaxf(v[j], v[j] * slope);
            if constexpr (GATED) t = (float)mv[j] > 0.f ? t : t * mslope;
            pk[j] = (__bf16)t;
        }
        v2u o = __builtin_bit_cast(v2u, pk);
        o.x &= keep, o.y &= keep;
        *reinterpret_cast<v2u *>(c.smem + base_of(J) + ((q * rows_of(J) + row) * DB_P + c.r + 1) * 16 + c.h * 8) = o;
    }
}

// One conv of the block for this wave's NB consecutive output rows [i0, i0 + NB) and one block of 32 outputs.
//   J : step 1..5; it reads 4 two-unit stages of the input array (K = 0) and 2 of each growth array K = 1..J-1
//   CO: outputs per row of a weight chunk.  32: a chunk is a whole stage [2 units][9 taps][32][16 B].  64 (conv5):
//       a chunk is one unit [9 taps][64][16 B], a stage reads two chunks and this wave the outputs [32 mb, 32 mb + 32)
//   JP, NBP: the previous growth step and the rows of it this wave holds in `pacc` (NBP = 0: none).  That step's
//       epilogue runs here, one row per stage under the MFMAs of stages 0..NBP-1, instead of between the two convs
//       where nothing would cover it: stages 0..3 of a step read only the input array and older growth arrays.  From
//       the barrier of stage 3 on every wave's rows of growth array JP are in LDS; the band's own rows are copied
//       to the block buffer in that stage.
// Per stage: wait for its chunk(s), barrier (the chunk has landed for every wave; every wave is done with the
// previous stage's slot), refill the freed slot(s), then three tap columns.  The loop is software-pipelined so that
// the wave alone keeps its SIMD's matrix pipe fed: the fragments of column kx + 1 are read from LDS under the MFMAs of
// column kx, the last column of a stage is multiplied after the next stage's barrier (under the latency of that
// stage's first weight fragments), and the first activation fragments of the next stage (resident data, independent
// of the ring) are read before its barrier.  Accumulation order per output: stage, tap column, tap row.
template <int J, int NB, int CO, int JP, int NBP, bool GATED>
__device__ __forceinline__ void db_conv(f32x16 (&acc)[NB], const f32x16 (&pacc)[NBP > 0 ? NBP : 1],
                                        const v2u (&praw)[3][4], int i0p, DbRing &ring, int &c_slot, const DbCtx &c,
                                        int i0, int mb) {
    constexpr int S = 2 + 2 * J;
    constexpr int NX = NB + 2;
    constexpr int COPY_STAGE = 3;
    static_assert(NBP == 0 || (NBP <= COPY_STAGE && S > COPY_STAGE + 1), "the deferred epilogue must finish before its array is read");
    const DenseArgs &a = c.a;
    unsigned char *smem = c.smem;
    const int r = c.r, h = c.h;
    bf16x8 w0[3], x0[NX], w1[3], x1[NX], w2[3], x2[NX];
    v4u braw[4] = {};   // the previous step's 16 bias values of this lane: channels q*8 + h*4 + j
    if constexpr (NBP > 0) db_lds_read16x4(smem + DB_BIAS_OFF + (JP - 1) * 256 + h * 16, 32, braw);
    const int lane16 = r * 16;
    // activation fragments of stage t: unit 2 ss + h of source K, rows i0 + (J - K) - 1 + rho
    auto stage_ptr = [&](int t) -> const unsigned char * {
        const int K = t < 4 ? 0 : 1 + ((t - 4) >> 1);
        const int ss = t < 4 ? t : ((t - 4) & 1);
        const int rows_k = 12 - 2 * K;
        const int base_k = K == 0 ? 0 : K == 1 ? base_of(1) : K == 2 ? base_of(2) : K == 3 ? base_of(3) : base_of(4);
        const int soff = base_k + ((ss * 2) * rows_k + i0 + J - K - 1) * DB_ROWB;
        return smem + soff + lane16 + (h ? rows_k * DB_ROWB : 0);
    };
#define DB_LOAD_W(WF, KX)                                                                                     \
    _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) WF[ky] = *reinterpret_cast<const bf16x8 *>(pw + ((ky * 3 + (KX)) * CO) * 16);
#define DB_LOAD_X(XF, KX)                                                                                     \
    _Pragma("unroll") for (int rho = 0; rho < NX; ++rho) XF[rho] = *reinterpret_cast<const bf16x8 *>(pa + (rho * DB_P + (KX)) * 16);
#define DB_MMA(WF, XF)                                                                                        \
    _Pragma("unroll") for (int rho = 0; rho < NX; ++rho) _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {   \
        const int nb_ = rho - ky;                                                                             \
        if (nb_ >= 0 && nb_ < NB) acc[nb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[ky], XF[rho], acc[nb_], 0, 0, 0); \
    }
    const unsigned char *pa = stage_ptr(0);
    DB_LOAD_X(x0, 0);
#pragma unroll
    for (int t = 0; t < S; ++t) {
        // every LDS read of the previous stage's weight slot has returned before the barrier lets it be refilled
        // (the NX youngest LDS operations are the activation prefetch above)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NX) : "memory");
        // refills of this stage: one chunk (two per conv5 stage after the first), issued further down under MFMAs
        if constexpr (CO == 32) {
            // three younger chunks may be in flight (three DMA instructions each per wave)
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else if (t == 0) {
            // entering conv5: chunks c + 2, c + 3 are in flight behind the two this stage reads
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            // steady state: one younger chunk in flight, two slots free after the barrier
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if (!DB_DEBUG(16)) __builtin_amdgcn_s_barrier();   // (bit 16, timing builds only: what the stage barriers cost)
        const unsigned char *pw;
        if constexpr (CO == 32) {
            pw = smem + DB_RING_OFF + c_slot * DB_CHUNK + (h * 9 * 32 + r) * 16;
            c_slot = c_slot + 1 == DB_SLOTS ? 0 : c_slot + 1;
        } else {
            const int slot_b = c_slot + 1 == DB_SLOTS ? 0 : c_slot + 1;
            pw = smem + DB_RING_OFF + (h ? slot_b : c_slot) * DB_CHUNK + (mb * 32 + r) * 16;
            c_slot = slot_b + 1 == DB_SLOTS ? 0 : slot_b + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        DB_LOAD_W(w0, 0);
        if (t > 0) { DB_MMA(w2, x2); }
        __builtin_amdgcn_sched_barrier(0);
        // column 0: its MFMAs, the reads of column 1, the refill of the freed slot and one row of the previous
        // step's epilogue, interleaved (an LDS-DMA instruction costs its wave 60-180 issue cycles and the epilogue
        // some sixty vector instructions: in the shadow of the MFMAs, not in front of them)
        db_issue_chunk(ring, a, smem, c.wave, c.pvoff);
        DB_LOAD_W(w1, 1);
        DB_LOAD_X(x1, 1);
        DB_MMA(w0, x0);
        constexpr int SIDE_VALU = (50 + 3 * NB - 1) / (3 * NB);   // half a row's epilogue over the column's MFMAs
        if (NBP > 0 && t < NBP) {
            db_growth_quads<JP, 0, 2, GATED>(pacc[t < NBP ? t : 0], praw[t < 3 ? t : 0], braw, i0p + t, c);
            DB_INTERLEAVE_V(3 * NB, 3 + NX, 3, SIDE_VALU, 0);
        } else {
            DB_INTERLEAVE(3 * NB, 3 + NX, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // column 1 under the reads of column 2 (conv5: the stage's second refill; the other half of the epilogue row)
        const bool second_issue = CO == 64 && t > 0;
        if (second_issue) db_issue_chunk(ring, a, smem, c.wave, c.pvoff);
        DB_LOAD_W(w2, 2);
        DB_LOAD_X(x2, 2);
        DB_MMA(w1, x1);
        if (NBP > 0 && t < NBP) {
            db_growth_quads<JP, 2, 4, GATED>(pacc[t < NBP ? t : 0], praw[t < 3 ? t : 0], braw, i0p + t, c);
            if (second_issue) {
                DB_INTERLEAVE_V(3 * NB, 3 + NX, 3, SIDE_VALU, 1);
            } else {
                DB_INTERLEAVE_V(3 * NB, 3 + NX, 0, SIDE_VALU, 1);
            }
        } else if (second_issue) {
            DB_INTERLEAVE(3 * NB, 3 + NX, 3, 1);
        } else {
            DB_INTERLEAVE(3 * NB, 3 + NX, 0, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NBP > 0) {
            if (t == COPY_STAGE && a.growth_out) {
                // growth array JP is complete since this stage's barrier: the band's own rows go to the block buffer,
                // one 16-byte item a thread (4 units x 2 rows x 32 pixels)
                const int u = c.tid >> 6, row = (c.tid >> 5) & 1, px = c.tid & 31;
                const int y = c.y0 + row;
                const uint4 val = db_lds_read16(smem + base_of(JP) + ((u * rows_of(JP) + (5 - JP) + row) * DB_P + px + 1) * 16);
                if (y < c.H && px < c.W)
                    *reinterpret_cast<uint4 *>(a.growth_out + (size_t)c.n * c.img_bytes +
                                               ((size_t)(y * a.units + 8 + 4 * (JP - 1) + u) * c.W + px) * 16) = val;
            }
        }
        if (t + 1 < S) {
            pa = stage_ptr(t + 1);
            DB_LOAD_X(x0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    DB_MMA(w2, x2);
#undef DB_LOAD_W
#undef DB_LOAD_X
#undef DB_MMA
}

// The five convs for a wave that holds NB1..NB4 rows of the growth steps (waves 0, 1: 3 2 2 1; waves 2, 3: 2 2 1 1).
template <int NB1, int NB2, int NB3, int NB4, bool GATED>
__device__ __forceinline__ void db_run(DbRing &ring, const DbCtx &c, const v2u (&mraw)[4][3][4]) {
    const int wave = c.wave;
    [[maybe_unused]] const DenseArgs &a = c.a;
    int c_slot = 0;   // ring slot of the chunk consumed next
    int i01, i02, i03, i04, nb_;
    wave_rows(rows_of(1), wave, i01, nb_);
    wave_rows(rows_of(2), 3 - wave, i02, nb_);
    wave_rows(rows_of(3), wave, i03, nb_);
    wave_rows(rows_of(4), 3 - wave, i04, nb_);
#define DB_ZERO(ACC, N)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < (N); ++i) _Pragma("unroll") for (int e = 0; e < 16; ++e) ACC[i][e] = 0.f;
    f32x16 acc1[NB1], acc2[NB2], acc3[NB3], acc4[NB4], acc5[1];
    DB_ZERO(acc1, NB1);
    db_conv<1, NB1, 32, 0, 0, GATED>(acc1, acc5, mraw[0], 0, ring, c_slot, c, i01, 0);   // (acc5, mraw[0]: unused)
    DB_STAMP(3);
    DB_ZERO(acc2, NB2);
    db_conv<2, NB2, 32, 1, NB1, GATED>(acc2, acc1, mraw[0], i01, ring, c_slot, c, i02, 0);
    DB_STAMP(5);
    DB_ZERO(acc3, NB3);
    db_conv<3, NB3, 32, 2, NB2, GATED>(acc3, acc2, mraw[1], i02, ring, c_slot, c, i03, 0);
    DB_STAMP(7);
    DB_ZERO(acc4, NB4);
    db_conv<4, NB4, 32, 3, NB3, GATED>(acc4, acc3, mraw[2], i03, ring, c_slot, c, i04, 0);
    DB_STAMP(9);
    // step 5: 192 -> 64 on the band's two rows; wave = (row, 32-output block)
    const int row5 = wave & 1, mb5 = wave >> 1;
    DB_ZERO(acc5, 1);
    db_conv<5, 1, 64, 4, NB4, GATED>(acc5, acc4, mraw[3], i04, ring, c_slot, c, row5, mb5);
    DB_STAMP(10);
    epi_f32x16 acc_t[1][1];
    acc_t[0][0] = acc5[0];
    conv_epilogue_tile<1, 1>(c.a.tail, acc_t, c.smem + DB_BIAS_OFF + 4 * 256 + mb5 * 128, c.n, c.y0 + row5, c.r, mb5 * 32,
                             c.h, true);
#undef DB_ZERO
}

// GATED: the data-gradient form (sign gates on steps 1..4); two kernels rather than one with both paths, which ran
// 1.5 x slower in either form.
// `buf`, `w0`, H, W, units, n_bands repeat a.buf, a.w[0], a.H, a.W, a.units, a.n_bands as kernel arguments of their own:
// the dispatcher preloads the first kernel arguments into scalar registers (the build passes
// -amdgpu-kernarg-preload-count for this file), so the input tile's and the first weight chunks' requests go out without
// waiting for the argument block to arrive from memory.
template <bool GATED>
__device__ __forceinline__ void dense_block_body(const unsigned char *buf_hot, const unsigned char *w0_hot, int H_hot,
                                                 int W_hot, int units_hot, int n_bands_hot, const DenseArgs &a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = H_hot, W = W_hot;
    const int n = blockIdx.x / n_bands_hot;
    const int y0 = (blockIdx.x - n * n_bands_hot) * DB_TH;
    const int img_bytes = H * units_hot * W * 16;
    DB_STAMP(0);

#define DB_RSRC(PTR, LEN)                                                                                     \
    __builtin_amdgcn_make_buffer_rsrc(                                                                        \
        (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(                               \
                      (int)(unsigned)((unsigned long long)(PTR) >> 32))                                       \
                  << 32) |                                                                                    \
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(PTR))),         \
        0, __builtin_amdgcn_readfirstlane(LEN), 0x00020000)

    // ---- sign gates (data-gradient use): the saved activations at this wave's output pixels of the four growth steps,
    // 4 channels x 4 quads per row.  Issued first: they come from HBM (written a whole forward pass ago) while the
    // input tile was written by the launch before, and the sixty DMA instructions of the prologue take two
    // microseconds to issue; the wait on them is the wait on the input tile, and no later use of them waits on the
    // weight ring.
    v2u mraw[4][3][4] = {};
    if constexpr (GATED) {
        const __amdgpu_buffer_rsrc_t rs_mk =
            DB_RSRC(a.mask + (size_t)n * ((size_t)H * a.mask_units * W * 16), H * a.mask_units * W * 16);
        const int q_step = __builtin_amdgcn_readfirstlane(W * 16);   // one unit further in the mask buffer
#pragma unroll
        for (int J = 1; J <= 4; ++J) {
            int i0, nb;
            wave_rows(rows_of(J), (J & 1) ? wave : 3 - wave, i0, nb);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < nb) {   // (wave-uniform)
                    const int y = y0 - (5 - J) + i0 + i;
                    const bool ok = y >= 0 && y < H && r < W;
                    const int vo = ok ? ((y * a.mask_units + a.mask_off[J - 1]) * W + r) * 16 + h * 8 : DB_OOB;
#pragma unroll
                    for (int q = 0; q < 4; ++q) mraw[J - 1][i][q] = __builtin_amdgcn_raw_buffer_load_b64(rs_mk, vo, q * q_step, 0);
                }
            }
        }
    }

    // ---- zero columns left and right of the growth arrays (the input array's are written by its DMA)
    {
        constexpr int G_UNIT_ROWS = 4 * (rows_of(1) + rows_of(2) + rows_of(3) + rows_of(4));
        if (tid < 2 * G_UNIT_ROWS) {
            const int ur = tid >> 1, side = tid & 1;
            constexpr int G_BASE = base_of(1);
            *reinterpret_cast<uint4 *>(smem + G_BASE + ur * DB_ROWB + side * (DB_P - 1) * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    // ---- input tile: 8 units x 12 rows x 34 pixels, piece j = k*4 + wave of 64 items each
    {
        const __amdgpu_buffer_rsrc_t rs_in = DB_RSRC(buf_hot + (size_t)n * img_bytes, img_bytes);
#pragma unroll
        for (int k = 0; k < DB_X_K; ++k) {
            const int j = k * 4 + wave;
            const int i = j * 64 + lane;
            const int u = i / (rows_of(0) * DB_P);
            const int p = i - u * (rows_of(0) * DB_P);
            const int row = p / DB_P;
            const int px = p - row * DB_P;
            const int y = y0 - 5 + row, x = px - 1;
            const bool valid = i < DB_X_ITEMS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int vo = valid ? ((y * units_hot + u) * W + x) * 16 : DB_OOB;
            const int dst = j < DB_X_PIECES ? j * 1024 : DB_DUMMY_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void *)(smem + dst), 16, vo, 0, 0, 0);
        }
    }
    // ---- biases: conv k at DB_BIAS_OFF + 256 k (lanes beyond the conv's outputs read out of range: zeros); every
    // wave issues all five (identical bytes) so that the DMA counts of the waves stay equal
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int nfl = k < 4 ? 32 : 64;
        const __amdgpu_buffer_rsrc_t rs_b = DB_RSRC(a.bias[k], nfl * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_void *)(smem + DB_BIAS_OFF + k * 256), 4, lane * 4, 0, 0, 0);
    }

    // ---- weight ring: flat chunk sequence over the five convs, four chunks issued ahead
    DbRing ring;
    ring.conv = 0, ring.local = 0, ring.left = 4, ring.slot = 0;
    ring.w_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)w0_hot);
    ring.w_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)w0_hot >> 32));
    ring.w_len = 4 * DB_CHUNK;
    DbCtx ctx = {a, smem, tid, wave, r, h, n, y0, img_bytes, {}, a.slope, a.mask_slope, H, W};
#pragma unroll
    for (int k = 0; k < 3; ++k) ctx.pvoff[k] = (wave + 4 * k) < 9 ? (wave + 4 * k) * 1024 + lane * 16 : DB_OOB;
#pragma unroll
    for (int k = 0; k < 4; ++k) db_issue_chunk(ring, a, smem, wave, ctx.pvoff);

    // ---- pull the block's packed weights (468 KiB) into this XCD's L2 ahead of the ring (behind the input tile's
    // and the ring's first requests: nothing waits for these).  Consecutive dense blocks
    // use different weights, so the ring's first touch of every line would miss L2.  Workgroups b and b + 8 usually
    // share an XCD (round-robin placement; a speed assumption only): the waves of such a group touch disjoint 1 KiB
    // pieces, one dword per 16 bytes, and drop the result.
    {
        const int per_xcd = (gridDim.x + 7) >> 3;
        const int gw = ((blockIdx.x >> 3) % per_xcd) * 4 + wave, stride = per_xcd * 4;
#pragma unroll 1
        for (int p = gw, it = 0; p < 468 && it < 16 && !DB_DEBUG(4); p += stride, ++it) {
            const int j = p < 36 ? 0 : p < 90 ? 1 : p < 162 ? 2 : p < 252 ? 3 : 4;
            const int start = j == 0 ? 0 : j == 1 ? 36 : j == 2 ? 90 : j == 3 ? 162 : 252;
            const int len = (j == 4 ? 216 : 36 + 18 * j) * 1024;
            // (an LDS-DMA into the dummy area: a load with a register destination would have to keep that register
            // reserved until it lands)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(DB_RSRC(a.w[j], len), (lds_void *)(smem + DB_DUMMY_OFF), 4,
                                                     (p - start) * 1024 + lane * 16, 0, 0, 0);
        }
    }


    // the input tile, the zero columns (and the gates) are in place before the first activation fragments are read
    // (the fragment prefetch of a step runs ahead of that step's first ring barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DB_STAMP(1);

    if (wave < 2)
        db_run<3, 2, 2, 1, GATED>(ring, ctx, mraw);
    else
        db_run<2, 2, 1, 1, GATED>(ring, ctx, mraw);
    // the padding chunks still in flight must land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DB_STAMP(11);
#undef DB_RSRC
}

__global__ __launch_bounds__(256, 1) void dense_block_band(const unsigned char *buf, const unsigned char *w0, int H, int W,
                                                           int units, int n_bands, const DenseArgs a) {
    dense_block_body<false>(buf, w0, H, W, units, n_bands, a);
}
__global__ __launch_bounds__(256, 1) void dense_block_band_gated(const unsigned char *buf, const unsigned char *w0, int H,
                                                                 int W, int units, int n_bands, const DenseArgs a) {
    dense_block_body<true>(buf, w0, H, W, units, n_bands, a);
}

bool dense_block_supported(int nf, int gc, int dtype, int W) {
    return dtype == SR355_DTYPE_BF16 && nf == 64 && gc == 32 && W >= 1 && W <= 32;
}

int launch_dense_block(const DenseArgs &a0, hipStream_t stream) {
    DenseArgs a = a0;
    if (!(a.slope >= 0.f && a.slope <= 1.f)) {
        set_error("dense block: the LeakyReLU slope %g is outside [0, 1]", (double)a.slope);
        return SR355_ERR_UNSUPPORTED;
    }
    if (a.units != 24 || a.W < 1 || a.W > 32 || a.H < 1 || a.N < 1 || !a.buf || a.tail.out_nchw || a.tail.out_u8 ||
        a.tail.cout_real != 64) {
        set_error("dense block: needs a 24-unit bf16 block buffer, images at most 32 pixels wide and a 64-channel "
                  "blocked / fp32 output (got units=%d N=%d H=%d W=%d)", a.units, a.N, a.H, a.W);
        return SR355_ERR_UNSUPPORTED;
    }
    a.n_bands = (a.H + DB_TH - 1) / DB_TH;
    const long long grid = (long long)a.N * a.n_bands;
    if (grid > 0x7fffffffLL || (size_t)a.H * a.units * a.W * 16 >= ((size_t)1 << 31)) {
        set_error("dense block: shape out of range");
        return SR355_ERR_INVALID;
    }
    a.tail.N = a.N, a.tail.H = a.H, a.tail.W = a.W;
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    if (dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_block_band),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS_BYTES);
        if (attr_err[dev] == hipSuccess)
            attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_block_band_gated),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS_BYTES);
    });
    if (check_hip(attr_err[dev], "hipFuncSetAttribute(dense_block_band)")) return SR355_ERR_HIP;
    if (a.mask)
        hipLaunchKernelGGL(dense_block_band_gated, dim3((unsigned)grid), dim3(256), DB_LDS_BYTES, stream,
                           (const unsigned char *)a.buf, (const unsigned char *)a.w[0], a.H, a.W, a.units, a.n_bands, a);
    else
        hipLaunchKernelGGL(dense_block_band, dim3((unsigned)grid), dim3(256), DB_LDS_BYTES, stream,
                           (const unsigned char *)a.buf, (const unsigned char *)a.w[0], a.H, a.W, a.units, a.n_bands, a);
    return check_hip(hipGetLastError(), "dense block launch");
}

}  // namespace sr355
