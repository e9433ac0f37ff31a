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
#define DB_DEBUG(BIT) (a.debug & (BIT))
#else
#define DB_STAMP(K)
#define DB_DEBUG(BIT) false
#endif

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ void db_lds_read16x4(const unsigned char *p, int stride, uint4 (&v)[4]) {
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
#define DB_INTERLEAVE(N_MFMA, N_DS, N_VMEM, SYNC)                                                             \
    {                                                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < (N_MFMA); ++i_) {                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SYNC);                                             \
            _Pragma("unroll") for (int j_ = 0; j_ < ((N_DS) + (N_MFMA)-1) / (N_MFMA); ++j_)                   \
                if (i_ * (((N_DS) + (N_MFMA)-1) / (N_MFMA)) + j_ < (N_DS))                                    \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, SYNC);                                     \
            if (i_ < (N_VMEM)) __builtin_amdgcn_sched_group_barrier(0x020, 1, SYNC);                          \
        }                                                                                                     \
    }

// One conv of the block for this wave's NB consecutive output rows [i0, i0 + NB) and one block of 32 outputs.
//   J : step 1..5; it reads 4 two-unit stages of the input array (K = 0) and 2 of each growth array K = 1..J-1
//   CO: outputs per row of a weight chunk.  32: a chunk is a whole stage [2 units][9 taps][32][16 B].  64 (conv5):
//       a chunk is one unit [9 taps][64][16 B], a stage reads two chunks and this wave the outputs [32 mb, 32 mb + 32)
// Per stage: wait for its chunk(s), barrier (the chunk has landed for every wave; every wave is done with the
// previous stage's slot), refill the freed slot(s), then three tap columns.  The loop is software-pipelined so that
// the wave alone keeps its SIMD's matrix pipe fed: the fragments of column kx + 1 are read from LDS under the MFMAs of
// column kx, the last column of a stage is multiplied after the next stage's barrier (under the latency of that
// stage's first weight fragments), and the first activation fragments of the next stage (resident data, independent
// of the ring) are read before its barrier.  Accumulation order per output: stage, tap column, tap row.
template <int J, int NB, int CO>
__device__ __forceinline__ void db_conv(f32x16 (&acc)[NB], DbRing &ring, int &c_slot, const DenseArgs &a,
                                        unsigned char *smem, int wave, int r, int h, const int (&pvoff)[3], int i0,
                                        int mb) {
    constexpr int S = 2 + 2 * J;
    constexpr int NX = NB + 2;
    bf16x8 w0[3], x0[NX], w1[3], x1[NX], w2[3], x2[NX];
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
    for (int t = 0; t < S; ++t) {
        // every LDS read of the previous stage's weight slot has returned before the barrier lets it be refilled
        // (the NX youngest reads are the activation prefetch above)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NX) : "memory");
        // refills of this stage: one chunk (two per conv5 stage after the first), issued further down under MFMAs
        int n_issue = 1;
        if constexpr (CO == 32) {
            // three younger chunks may be in flight (three DMA instructions each per wave)
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else if (t == 0) {
            // entering conv5: chunks c + 2, c + 3 are in flight behind the two this stage reads
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            // steady state: one younger chunk in flight, two slots free after the barrier
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            n_issue = 2;
        }
        __builtin_amdgcn_s_barrier();
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
        // column 0: its MFMAs, the reads of column 1 and the refill of the freed slot, one of each at a time (an
        // LDS-DMA instruction costs its wave 60-180 issue cycles: in the shadow of an MFMA, not in front of it)
        db_issue_chunk(ring, a, smem, wave, pvoff);
        DB_LOAD_W(w1, 1);
        DB_LOAD_X(x1, 1);
        DB_MMA(w0, x0);
        DB_INTERLEAVE(3 * NB, 3 + NX, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CO == 64) {
            if (n_issue == 2) {
                db_issue_chunk(ring, a, smem, wave, pvoff);
                DB_LOAD_W(w2, 2);
                DB_LOAD_X(x2, 2);
                DB_MMA(w1, x1);
                DB_INTERLEAVE(3 * NB, 3 + NX, 3, 1);
            } else {
                DB_LOAD_W(w2, 2);
                DB_LOAD_X(x2, 2);
                DB_MMA(w1, x1);
                DB_INTERLEAVE(3 * NB, 3 + NX, 0, 1);
            }
        } else {
            DB_LOAD_W(w2, 2);
            DB_LOAD_X(x2, 2);
            DB_MMA(w1, x1);
            DB_INTERLEAVE(3 * NB, 3 + NX, 0, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
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

// Epilogue of a growth step: bias, LeakyReLU, sign gate, zero outside the image, bf16 into the step's LDS array.
template <int J, int NB>
__device__ __forceinline__ void db_growth_store(const f32x16 (&acc)[NB], const DenseArgs &a, unsigned char *smem,
                                                const unsigned (&mbits)[3], int i0, int y0, int r, int h) {
    const float slope = a.slope, mslope = a.mask_slope;
    const bool gated = a.mask != nullptr;
    uint4 braw[4];   // this lane's 16 bias values: channels q*8 + h*4 + j
    db_lds_read16x4(smem + DB_BIAS_OFF + (J - 1) * 256 + h * 16, 32, braw);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int y = y0 - (5 - J) + i0 + i;
        const bool ok = y >= 0 && y < a.H && r < a.W;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = __builtin_bit_cast(float4, braw[q]);
            const float v[4] = {acc[i][4 * q + 0] + bv.x, acc[i][4 * q + 1] + bv.y, acc[i][4 * q + 2] + bv.z,
                                acc[i][4 * q + 3] + bv.w};
            bf16x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j] > 0.f ? v[j] : v[j] * slope;
                if (gated) t = ((mbits[i] >> (4 * q + j)) & 1u) ? t : t * mslope;
                pk[j] = (__bf16)(ok ? t : 0.f);
            }
            *reinterpret_cast<bf16x4 *>(smem + base_of(J) + ((q * rows_of(J) + i0 + i) * DB_P + r + 1) * 16 + h * 8) = pk;
        }
    }
}

template <int J>
__device__ __forceinline__ void db_growth_step(DbRing &ring, int &c_slot, const DenseArgs &a, unsigned char *smem,
                                               int tid, int wave, int r, int h, const int (&pvoff)[3],
                                               const unsigned (&mbits)[3], int n, int y0, int img_bytes) {
    int i0, nb;
    wave_rows(rows_of(J), (J & 1) ? wave : 3 - wave, i0, nb);
    // rows per wave: step 1 -> 3 3 2 2, step 2 -> 2 2 2 2, step 3 -> 2 2 1 1, step 4 -> 1 1 1 1
    constexpr int NB_HI = (rows_of(J) + 3) / 4, NB_LO = rows_of(J) / 4;
    if (nb == NB_HI) {
        f32x16 acc[NB_HI];
#pragma unroll
        for (int i = 0; i < NB_HI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        db_conv<J, NB_HI, 32>(acc, ring, c_slot, a, smem, wave, r, h, pvoff, i0, 0);
        DB_STAMP(2 * J);
        db_growth_store<J, NB_HI>(acc, a, smem, mbits, i0, y0, r, h);
    } else if constexpr (NB_LO != NB_HI) {
        f32x16 acc[NB_LO];
#pragma unroll
        for (int i = 0; i < NB_LO; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        db_conv<J, NB_LO, 32>(acc, ring, c_slot, a, smem, wave, r, h, pvoff, i0, 0);
        DB_STAMP(2 * J);
        db_growth_store<J, NB_LO>(acc, a, smem, mbits, i0, y0, r, h);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (a.growth_out) {   // the band's own rows go to the block buffer: 4 units x 2 rows x 32 pixels, one item a thread
        const int u = tid >> 6, row = (tid >> 5) & 1, px = tid & 31;
        const int y = y0 + row;
        if (y < a.H && px < a.W) {
            const uint4 val = db_lds_read16(smem + base_of(J) + ((u * rows_of(J) + (5 - J) + row) * DB_P + px + 1) * 16);
            *reinterpret_cast<uint4 *>(a.growth_out + (size_t)n * img_bytes +
                                       ((size_t)(y * a.units + 8 + 4 * (J - 1) + u) * a.W + px) * 16) = val;
        }
    }
}

__global__ __launch_bounds__(256, 1) void dense_block_band(const DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = a.H, W = a.W;
    const int n = blockIdx.x / a.n_bands;
    const int y0 = (blockIdx.x - n * a.n_bands) * DB_TH;
    const int img_bytes = H * a.units * W * 16;
    DB_STAMP(0);

#define DB_RSRC(PTR, LEN)                                                                                     \
    __builtin_amdgcn_make_buffer_rsrc(                                                                        \
        (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(                               \
                      (int)(unsigned)((unsigned long long)(PTR) >> 32))                                       \
                  << 32) |                                                                                    \
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(PTR))),         \
        0, __builtin_amdgcn_readfirstlane(LEN), 0x00020000)

    // ---- pull the block's packed weights (468 KiB) into this XCD's L2 ahead of the ring.  Consecutive dense blocks
    // use different weights, so the ring's first touch of every line would miss L2 and the four chunks it keeps in
    // flight would bound the whole kernel by the latency of the next cache level.  Workgroups b and b + 8 usually
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

    // ---- zero columns left and right of the growth arrays (the input array's are written by its DMA)
    {
        constexpr int G_UNIT_ROWS = 4 * (rows_of(1) + rows_of(2) + rows_of(3) + rows_of(4));
        if (tid < 2 * G_UNIT_ROWS) {
            const int ur = tid >> 1, side = tid & 1;
            *reinterpret_cast<uint4 *>(smem + base_of(1) + ur * DB_ROWB + side * (DB_P - 1) * 16) = make_uint4(0, 0, 0, 0);
        }
    }

    // ---- input tile: 8 units x 12 rows x 34 pixels, piece j = k*4 + wave of 64 items each
    {
        const __amdgpu_buffer_rsrc_t rs_in = DB_RSRC(a.buf + (size_t)n * img_bytes, img_bytes);
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
            const int vo = valid ? ((y * a.units + u) * W + x) * 16 : DB_OOB;
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
    ring.w_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)a.w[0]);
    ring.w_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)a.w[0] >> 32));
    ring.w_len = 4 * DB_CHUNK;
    int pvoff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pvoff[k] = (wave + 4 * k) < 9 ? (wave + 4 * k) * 1024 + lane * 16 : DB_OOB;
#pragma unroll
    for (int k = 0; k < 4; ++k) db_issue_chunk(ring, a, smem, wave, pvoff);
    int c_slot = 0;   // ring slot of the chunk consumed next

    // ---- sign masks (data-gradient use): bit (4q + j) of mbits[J-1][i] = the saved activation of channel
    // q*8 + h*4 + j at this lane's pixel of row i of step J is positive.  Loaded behind the prologue's DMA and turned into
    // bits before the first barrier: the wait on them is the wait on the input tile, and no later wait on them drains
    // the weight ring.
    unsigned mbits[4][3] = {};
    if (a.mask) {
        const __amdgpu_buffer_rsrc_t rs_mk =
            DB_RSRC(a.mask + (size_t)n * ((size_t)H * a.mask_units * W * 16), H * a.mask_units * W * 16);
        v2u raw[4][3][4];
#pragma unroll
        for (int J = 1; J <= 4; ++J) {
            int i0, nb;
            wave_rows(rows_of(J), (J & 1) ? wave : 3 - wave, i0, nb);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int y = y0 - (5 - J) + i0 + i;
                const bool ok = i < nb && y >= 0 && y < H && r < W;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int vo = ok ? ((y * a.mask_units + a.mask_off[J - 1] + q) * W + r) * 16 + h * 8 : DB_OOB;
                    raw[J - 1][i][q] = __builtin_amdgcn_raw_buffer_load_b64(rs_mk, vo, 0, 0);
                }
            }
        }
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                unsigned bits = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x4 mv = __builtin_bit_cast(bf16x4, raw[J][i][q]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bits |= ((float)mv[j] > 0.f ? 1u : 0u) << (4 * q + j);
                }
                mbits[J][i] = bits;
            }
    }


    // the input tile and the zero columns are in place before the first activation fragments are read (the
    // fragment prefetch of a step runs ahead of that step's first ring barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (in a gradient launch the mask loads above are the youngest)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DB_STAMP(1);

    db_growth_step<1>(ring, c_slot, a, smem, tid, wave, r, h, pvoff, mbits[0], n, y0, img_bytes);
    DB_STAMP(3);
    db_growth_step<2>(ring, c_slot, a, smem, tid, wave, r, h, pvoff, mbits[1], n, y0, img_bytes);
    DB_STAMP(5);
    db_growth_step<3>(ring, c_slot, a, smem, tid, wave, r, h, pvoff, mbits[2], n, y0, img_bytes);
    DB_STAMP(7);
    db_growth_step<4>(ring, c_slot, a, smem, tid, wave, r, h, pvoff, mbits[3], n, y0, img_bytes);
    DB_STAMP(9);

    // ---- step 5: 192 -> 64 on the band's two rows; wave = (row, 32-output block)
    {
        const int row5 = wave & 1, mb5 = wave >> 1;
        f32x16 acc5[1];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc5[0][e] = 0.f;
        db_conv<5, 1, 64>(acc5, ring, c_slot, a, smem, wave, r, h, pvoff, row5, mb5);
        DB_STAMP(10);
        epi_f32x16 acc_t[1][1];
        acc_t[0][0] = acc5[0];
        conv_epilogue_tile<1, 1>(a.tail, acc_t, smem + DB_BIAS_OFF + 4 * 256 + mb5 * 128, n, y0 + row5, r, mb5 * 32, h, true);
    }
    // the padding chunks still in flight must land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DB_STAMP(11);
#undef DB_RSRC
}

bool dense_block_supported(int nf, int gc, int dtype, int W) {
    return dtype == SR355_DTYPE_BF16 && nf == 64 && gc == 32 && W >= 1 && W <= 32;
}

int launch_dense_block(const DenseArgs &a0, hipStream_t stream) {
    DenseArgs a = a0;
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
    });
    if (check_hip(attr_err[dev], "hipFuncSetAttribute(dense_block_band)")) return SR355_ERR_HIP;
    hipLaunchKernelGGL(dense_block_band, dim3((unsigned)grid), dim3(256), DB_LDS_BYTES, stream, a);
    return check_hip(hipGetLastError(), "dense block launch");
}

}  // namespace sr355
