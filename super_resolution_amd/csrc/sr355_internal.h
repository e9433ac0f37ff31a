// Internal declarations shared by the translation units of libsr355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sr355.h"

namespace sr355 {

// Arguments of one conv3x3 launch, passed by value to the kernel.
struct ConvArgs {
    const unsigned char *in;     // blocked activations
    const unsigned char *w;      // packed weights [cin_units][9][cout_pad][16 B]
    const float *bias;           // [cout_pad]
    unsigned char *out_act;      // blocked, may be null
    float *out_f32;              // fp32 trunk [N][H][f32_units][W][4], may be null
    float *out_nchw;             // [N][cout_real][H][W], may be null
    int f16_in;                  // bf16 path only: input and weights hold IEEE fp16 (same 16-byte units, f16 MFMA)
    int f16_out;                 // bf16 path only: out_act is stored as fp16 (it feeds an f16_in conv)
    unsigned char *out_u8;       // uint8 image [N][H][W][cout_real] = trunc(255 * clamp(v, 0, 1)), may be null
    const unsigned char *res_act; // blocked bf16 activations [N][H][res_act_units][W][16 B] whose channels [0, cout) are
                                 // the first residual instead of res1 (same weights res1_w / res1_scale), may be null
    int res_act_units;
    const float *res1;           // fp32 trunk, may be null
    const float *res2;           // fp32 trunk, may be null
    float res1_scale, res2_scale; // v = resK_w * resK + resK_scale * v
    float res1_w, res2_w;
    const unsigned char *mask_act; // blocked activations whose sign gates v (LeakyReLU backward), may be null
    int mask_units, mask_unit_off;
    float mask_slope;
    float slope;                 // LeakyReLU slope, 1.0f = identity
    int N, H, W;                 // output dims
    int in_units;                // units per pixel of the input buffer
    int n_stages;                // stages to run (2 * n_real_stages with split weights)
    int n_real_stages;           // cin_units / SU: stages of real input
    int ups;                     // 0 | 1
    int zins;                    // with ups = 1: the input is the zero-inserted x2 image (value at even y, x only, zeros
                                 // elsewhere; input dims (H+1)/2 x (W+1)/2): the data gradient of a stride-2 conv
    int stride;                  // 1 | 2 (register-staged kernel only); stride 2 reads an in_H x in_W input
    int in_H, in_W;
    int out_units, out_unit_off; // geometry of out_act
    int cout_real, cout_pad;
    int f32_units;               // units per pixel of the fp32 trunk tensors (cout_real / 4)
    int tiles_x, tiles_y;
    int rotate;                  // ring kernel: 1 = per-tile rotation of the stage order
    int debug;                   // timing experiments only: bit 0 = skip reads, bit 1 = skip stores
    unsigned long long *trace;   // timing experiments only: 16 wall-clock stamps per workgroup, normally null
};

// Arguments of one fused dense-block launch (dense_block.hip): five convs over one 24-unit block buffer.
struct DenseArgs {
    const unsigned char *buf;      // [N][H][24][W][16 B] bf16: units [0, 8) hold the block input
    unsigned char *growth_out;     // where steps 1..4 store their 32 channels each (units 8..23 of a buffer of the
                                   // same geometry, normally `buf` itself), or null: kept in LDS only
    const unsigned char *w[5];     // packed weights of conv1..conv5 ([cin_units][9][cout_pad][16 B])
    const float *bias[5];          // 32, 32, 32, 32, 64 floats
    const unsigned char *mask;     // data-gradient use: saved activations whose sign gates steps 1..4, or null
    int mask_units, mask_off[4];   // units per pixel of `mask`, unit offset of the gate of step k + 1
    float mask_slope;
    float slope;                   // LeakyReLU slope of steps 1..4 (1.0f = identity)
    int N, H, W, units;
    int n_bands;                   // set by the launcher
    int debug;                     // timing experiments only (builds with SR355_TRACE): bit 0 = no weight traffic,
                                   // bit 2 = no L2 prefetch
    unsigned long long *trace;     // timing experiments only: 8 shader-clock + 8 wall-clock stamps per workgroup
    ConvArgs tail;                 // conv5's epilogue: residuals, carriers, outputs (cout 64); N/H/W set by the launcher
};

// One pass of the chain kernel (conv3x3_chain.hip): a conv restricted to one block of 32 output channels.
// What the DMA cursor needs of a pass, 64 bytes in front of its descriptor: the kernel brings the next pass's record
// into the LDS one pass ahead (one dword per lane of a 16-lane load) instead of waiting on scalar loads when it gets there.
struct ChainCursor {
    unsigned long long w, in, bias;   // = a.w, a.in, a.bias
    int n_stages, cout_pad;           // = a.n_stages, a.cout_pad
    int need, dep_stage, need2, dep2_stage;
    unsigned perm_lo, perm_hi;        // = ChainPass::perm
    int pad[2];
};
static_assert(sizeof(ChainCursor) == 64, "one dword per lane of a quarter-wave");
struct ChainPass {
    ChainCursor cur;  // filled by launch_conv3x3_chain's caller through chain_fill_cursor
    ConvArgs a;       // w and bias already point at the block; n_stages set; N/H/W/in_units as in ChainArgs
    int cb_co;        // first output channel of the block
    int need;         // passes that must be complete on the neighbouring tiles before stage dep_stage is requested
    int dep_stage;    // first stage that reads the previous conv's output: n_stages = none, 0 = all, otherwise >= 2
    int need2;        // a second such point (a conv whose input was written by two passes: the first one's channels are
    int dep2_stage;   // needed from stage 0, the second one's from dep2_stage >= 2); need2 = 0: none
    // Order in which the pass walks its input-channel stages: nibble i = the stage requested i-th (at most 16 stages).
    // A dense block's convs read the same leading channels one after the other; walking them in the reverse of the
    // previous pass's order makes the first stages of a pass hit the lines the XCD's L2 touched last.
    unsigned long long perm;
    int pad[1];
};
inline void chain_fill_cursor(ChainPass &ps) {
    ps.cur.w = (unsigned long long)ps.a.w, ps.cur.in = (unsigned long long)ps.a.in;
    ps.cur.bias = (unsigned long long)ps.a.bias;
    ps.cur.n_stages = ps.a.n_stages, ps.cur.cout_pad = ps.a.cout_pad;
    ps.cur.need = ps.need, ps.cur.dep_stage = ps.dep_stage, ps.cur.need2 = ps.need2, ps.cur.dep2_stage = ps.dep2_stage;
    ps.cur.perm_lo = (unsigned)ps.perm, ps.cur.perm_hi = (unsigned)(ps.perm >> 32);
    ps.cur.pad[0] = ps.cur.pad[1] = 0;
}
// Rows a wave computes of the chain kernel's tile (tile = 4 waves x CHAIN_NB rows x 32 pixels) and the workgroups of
// it a CU holds: 4 rows -> 62 KB of LDS, two per CU; 3 rows -> 51 KB, three per CU.
#ifndef SR355_CHAIN_NB
#define SR355_CHAIN_NB 4
#endif
constexpr int CHAIN_NB = SR355_CHAIN_NB;
constexpr int CHAIN_TILE_ROWS = 4 * CHAIN_NB;
constexpr int CHAIN_WG_PER_CU = CHAIN_NB >= 4 ? 2 : 3;
struct ChainArgs {
    const ChainPass *passes;   // device memory, written before the launch
    int n_passes;
    unsigned *flags;           // one word per tile, zero at launch: passes complete on that tile
    unsigned *err;             // raised when a wait gave up
    int N, H, W, in_units;
    int tiles_x, tiles_y;      // set by the launcher
    unsigned long long *trace; // timing builds only (SR355_TRACE): eight words per workgroup, normally null
};

// Timing experiments (skip the loads / the stores of a conv) exist in builds with SR355_TRACE only (csrc/tools); the
// library proper has no such switch.
#ifdef SR355_TRACE
#define SR355_DBG(A, BIT) ((A).debug & (BIT))
#else
#define SR355_DBG(A, BIT) false
#endif

void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what);

// Launch the MFMA conv.  cin_units must be even; returns SR355_* code.
int launch_conv3x3(const ConvArgs &a, int dtype, int cin_units, int weight_split, hipStream_t stream);

// Persistent LDS-DMA ring variant (bf16, large images); same contract as launch_conv3x3.
int launch_conv3x3_ring(const ConvArgs &a, int cin_units, int weight_split, hipStream_t stream);

// Ring kernel with a branch-free, MFMA-interleaved stage body; n_stages, n_real_stages and debug must be set.
int launch_conv3x3_fine(const ConvArgs &a, hipStream_t stream);
int launch_conv3x3_fine_small(const ConvArgs &a, hipStream_t stream);

// A chain of convs over one geometry as one persistent launch (large images).  `max_workgroups` > 0 caps the grid.
bool chain_supported(const ConvArgs &a, int cin_units);
int chain_resident_workgroups(int *out);
int launch_conv3x3_chain(const ChainArgs &c, int max_workgroups, hipStream_t stream);

// Fused dense block for images at most 32 pixels wide (bf16, 64 features, 32 growth channels).
bool dense_block_supported(int nf, int gc, int dtype, int W);
int launch_dense_block(const DenseArgs &a, hipStream_t stream);

int launch_nchw_to_blocked(const float *src, int N, int C, int H, int W, int dtype, void *dst,
                           int dst_units, int split, hipStream_t stream);
int launch_u8hwc_to_blocked(const unsigned char *src, int N, int C, int H, int W, int dtype, void *dst,
                            int dst_units, int split, hipStream_t stream);
int launch_adam_flat(float *p, const float *g, float *m, float *v, size_t n, float beta1, float beta2, float omb1,
                     float omb2, float eps, float weight_decay, float step_size, float bc2_sqrt, hipStream_t stream);
int launch_blocked_to_nchw(const void *src, int src_units, int unit_off, int dtype, int N, int C,
                           int H, int W, float *dst, hipStream_t stream);
int launch_pack_weight(const float *w, const float *bias, int cout, int cin, int dtype, int split,
                       void *wp, float *bp, hipStream_t stream);

// Batched weight packing: up to PACK_BATCH jobs travel in the kernel arguments of one launch
// (a training step repacks every conv after the optimizer update: ~350 forward and ~1000 transposed packs).
constexpr int PACK_BATCH = 64;
struct PackJob {      // one launch_pack_weight call
    const float *w, *bias;
    void *wp;
    float *bp;
    int cout, cin, split;
    int f16;              // bf16 path: store IEEE fp16 instead of bf16 (no rounding residue)
};
struct PackTJob {     // one launch_pack_weight_t call
    const float *w;
    void *wp;
    int Co, Ci, ci_lo, n_out, unit_off, cout_pad;
    float scale;
};
int launch_pack_weight_batch(const PackJob *jobs, int n, int dtype, hipStream_t stream);
// All jobs in one launch each: the table is copied to `table_dev` (n * sizeof(job) bytes of device memory) first.
int launch_pack_weight_table(const PackJob *jobs_host, int n, int dtype, void *table_dev, hipStream_t stream);
int launch_pack_weight_t_table(const PackTJob *jobs_host, int n, int dtype, void *table_dev, hipStream_t stream);
int launch_pack_weight_t_batch(const PackTJob *jobs, int n, int dtype, hipStream_t stream);

// ---- backward.hip ----
int launch_pack_weight_t(const float *w, int Co, int Ci, int ci_lo, int n_out, int dtype, float scale, void *wp,
                         int unit_off, int cout_pad, hipStream_t stream);
int launch_pool2x2(const void *in, int N, int H, int W, int U, int dtype, const void *mask, int mask_units,
                   int mask_unit_off, float mask_slope, float *out_f32, void *out_act, int out_units, int out_unit_off,
                   hipStream_t stream);
size_t wgrad_workspace_bytes(int cout, int cin, int N, int H, int W);
// All five weight / bias gradients of a dense block in three launches (see backward.hip).
size_t wgrad_block_workspace_bytes(int nf, int gc, int N, int H, int W);
// One reduction of per-workgroup slabs into a conv's weight / bias gradient (see wgrad_reduce_body in backward.hip).
struct WgradReduceJob {
    const float *slabs, *bias_slabs;
    float *dw, *db;
    int cout, cin, CO, G;
    float scale;
};
int launch_wgrad3x3_block(const void *act, const void *stack, int units, int nf, int gc, int N, int H, int W, float s5,
                          float *const dw[5], float *const db[5], float *slabs, hipStream_t stream,
                          WgradReduceJob *defer = nullptr, float s_growth = 1.f);
int launch_wgrad3x3_blocks(int n_blocks, const void *const *act, const void *const *stack, int units, int nf, int gc,
                           int N, int H, int W, const float *s5, float *const (*dw)[5], float *const (*db)[5],
                           float *slabs, hipStream_t stream, WgradReduceJob *defer = nullptr, float s_growth = 1.f);
int launch_wgrad_reduce_jobs(const WgradReduceJob *jobs, int n, int nf, int gc, hipStream_t stream);
int launch_wgrad3x3(const void *in, int in_units, int ups, const void *dy, int dy_units, int dy_unit_off, int N, int H,
                    int W, int cout, int cin, float scale, int accumulate, float *dw, float *db, float *slabs,
                    hipStream_t stream, int dy_stride2 = 0);

int launch_wgrad_reduce(const float *slabs, float *dw, int cout, int cin, int CO, int G, float scale, int accumulate,
                        const float *bias_slabs, float *db, hipStream_t stream);

// ---- wgrad_f32.hip: the weight gradient on blocked fp32 tensors (4-channel units), same slab layout ----
size_t wgrad_f32_workspace_bytes(int cout, int cin, int N, int H, int W);
int launch_wgrad3x3_f32(const float *in, int in_units, int ups, const float *dy, int dy_units, int dy_unit_off, int N,
                        int H, int W, int cout, int cin, float scale, int accumulate, float *dw, float *db, float *slabs,
                        hipStream_t stream);

// Device address, on the current device, of the library's static scratch array (SR355_STATIC_SCRATCH_BYTES).
int wgrad_static_scratch(float **out);

// ---- aux_layers.hip ----
size_t bn_workspace_floats(int N, int H, int W, int C);
int launch_bn_lrelu_fwd(const float *x, int N, int H, int W, int C, const float *gamma, const float *beta, float eps,
                        float slope, int training, float momentum, float *running_mean, float *running_var,
                        long long *num_batches, float *save_mean, float *save_rstd, void *out_act, int out_units,
                        int out_unit_off, float *workspace, hipStream_t stream, int groups = 1);
int launch_bn_lrelu_bwd(const float *x, const void *dy, int dy_units, int dy_unit_off, int N, int H, int W, int C,
                        const float *gamma, const float *beta, const float *save_mean, const float *save_rstd,
                        float slope, int training, void *dx, int dx_units, int dx_unit_off, float *dgamma, float *dbeta,
                        float *workspace, hipStream_t stream, int groups = 1);
int launch_maxpool2x2_fwd(const void *in, int N, int H, int W, int U, void *out, hipStream_t stream);
int launch_maxpool2x2_bwd(const void *in, const void *dy, int N, int H, int W, int U, void *dx, hipStream_t stream);
int launch_pixel_shuffle2(const void *src, int N, int H, int W, int c_out, int dtype, int inverse, void *dst,
                          hipStream_t stream);
int loss_parts(size_t n);
int launch_loss(const float *a, const float *b, float b_const, size_t n, int mode, float *loss, float *grad,
                float *workspace, hipStream_t stream);

inline int unit_channels(int dtype) { return dtype == SR355_DTYPE_BF16 ? 8 : dtype == SR355_DTYPE_F32 ? 4 : -1; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int cin_units_for(int cin, int dtype) { return round_up((cin + unit_channels(dtype) - 1) / unit_channels(dtype), 2); }
inline int cout_pad_for(int cout) { return round_up(cout, 32); }
// Units of real input a conv consumes under a split mode (input split doubles the channels).
inline int real_units_for(int cin, int dtype, int split) {
    return cin_units_for(split == SR355_SPLIT_INPUT ? 2 * cin : cin, dtype);
}
// Units of the packed weight image (both split modes append the residue image).
inline int packed_units_for(int cin, int dtype, int split) {
    return real_units_for(cin, dtype, split) * (split ? 2 : 1);
}

}  // namespace sr355
