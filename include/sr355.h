/*
 * sr355 — MI355X (gfx950) native kernels for the ESRGAN x4 generator hot path.
 *
 * C ABI of libsr355.so.  Plain pointers and sizes only; every pointer named *_dev is a
 * device pointer owned by the caller (PyTorch's allocator in the Python host); the library
 * never allocates, frees or retains device memory.  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  Every function returns SR355_OK or a negative error
 * code; sr355_last_error() returns the message of the calling thread's last failure.
 *
 * Reference interfaces replaced (all in /root/reference/esrgan/models/esrgan_model.py):
 *   nn.Conv2d(cin,cout,3,1,1) + torch.cat + LeakyReLU + scaled skip   :10-24
 *   RRDB skip                                                          :35-39
 *   GeneratorRRDB.forward                                              :55-62
 *   F.interpolate(nearest, x2) feeding a conv                          :59-60
 *
 * Activation layout ("blocked"): [N][H][U][W][16 bytes] where one 16-byte unit holds
 * 8 consecutive channels in bf16 mode or 4 in fp32 mode and U = channels / (8|4).
 * A dense block keeps x and its four growth outputs in ONE such buffer (U = 24 in bf16,
 * channels 0..191); each conv reads units [0, cin_units) and writes its result at a unit
 * offset, so no concatenation is ever materialised.
 * fp32 "trunk" tensors (residual carriers) use the same layout with 4-channel units.
 */
#ifndef SR355_H
#define SR355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR355_OK 0
#define SR355_ERR_INVALID (-1)   /* bad argument / shape / alignment / size   */
#define SR355_ERR_HIP (-2)       /* HIP runtime or launch failure             */
#define SR355_ERR_UNSUPPORTED (-3)

#define SR355_DTYPE_BF16 0       /* bf16 storage, fp32 accumulate (32x32x16 MFMA) */
#define SR355_DTYPE_F32 1        /* fp32 storage, exact fp32 MFMA (32x32x2)        */

/* Split modes (bf16 only).  A bf16 product keeps 8 significant bits per factor; where that is
 * not enough the rounding residue is carried as extra terms of the same MFMA accumulation:
 *   SPLIT_WEIGHT: packed weights hold hi = bf16(w) and lo = bf16(w - hi); the conv runs over the
 *                 input twice (hi pass, lo pass) -> weight rounding error ~2^-17 instead of 2^-9.
 *   SPLIT_INPUT : as SPLIT_WEIGHT, and the input buffer holds bf16(x) in channels [0,C) and the
 *                 residue bf16(x - bf16(x)) in channels [C,2C) (sr355_nchw_to_blocked split=1);
 *                 used for the first conv, whose 3 input channels leave the K dimension idle. */
#define SR355_SPLIT_NONE 0
#define SR355_SPLIT_WEIGHT 1
#define SR355_SPLIT_INPUT 2
#define SR355_PACK_F16 3    /* sr355_pack_weight only (bf16 dtype): one weight image stored as IEEE fp16, for a
                              conv run with f16_in = 1 */

#define SR355_ABI_VERSION 18

int sr355_abi_version(void);
const char *sr355_last_error(void);

/* Channels per 16-byte unit for a dtype (8 or 4); <0 on bad dtype. */
int sr355_unit_channels(int dtype);

/* ---- layout conversion (module boundary: NCHW fp32 tensors, esrgan_model.py:55,:62) ---- */

/* src [N][C][H][W] fp32 -> dst blocked [N][H][dst_units][W][unit]; channels >= C are zero.
 * split=1 (bf16): channels [C,2C) receive the bf16 rounding residue of channels [0,C). */
int sr355_nchw_to_blocked(const float *src_dev, int N, int C, int H, int W, int dtype,
                          void *dst_dev, int dst_units, int split, size_t dst_bytes, void *stream);
/* blocked units [unit_off, unit_off + ceil(C/unit)) -> dst [N][C][H][W] fp32. */
int sr355_blocked_to_nchw(const void *src_dev, int src_units, int unit_off, int dtype,
                          int N, int C, int H, int W, float *dst_dev, size_t dst_bytes,
                          void *stream);

/* ---- weights: OIHW fp32 (the state_dict layout, kept as the parameter) -> packed ---- */

/* Packed layout [units][9][cout_pad][unit]; cout_pad = round_up(cout, 32),
 * units = round_up(ceil(cin/unit), 2), doubled by a split mode (residue image appended;
 * SPLIT_INPUT counts 2*cin channels).  Returns bytes (0 on bad args). */
size_t sr355_packed_weight_bytes(int cout, int cin, int dtype, int split);
/* bias_packed_dev receives cout_pad floats (zero padded).  */
int sr355_pack_weight(const float *w_oihw_dev, const float *bias_dev, int cout, int cin,
                      int dtype, int split, void *w_packed_dev, size_t w_packed_bytes,
                      float *bias_packed_dev, void *stream);

/* ---- the convolution op (replaces nn.Conv2d 3x3/s1/p1 + its fused neighbours) ---- */
typedef struct sr355_conv_args {
    int dtype;                 /* SR355_DTYPE_*                                            */
    int N, H, W;               /* OUTPUT batch / height / width                            */
    int upsample;              /* 1: input is [N][H/2][..][W/2], nearest x2 fused on load   */
    const void *in_dev;        /* blocked activations                                      */
    size_t in_bytes;
    int in_units;              /* units per pixel of the input buffer                      */
    int cin_units;             /* units consumed: channels [0, cin_units*unit); even       */
    const void *w_packed_dev;  /* sr355_pack_weight output                                 */
    size_t w_packed_bytes;
    int weight_split;          /* SR355_SPLIT_*: the mode the weights were packed with     */
    const float *bias_packed_dev;
    int cout;                  /* real output channels                                     */
    float lrelu_slope;         /* 1.0f = no activation                                     */
    /* v = conv + bias; v = lrelu(v);
     * if mask: v *= (mask > 0 ? 1 : mask_slope)   (mask: blocked activations of this dtype, read
     *          at unit mask_unit_off: the LeakyReLU derivative in the backward pass);
     * if res1: v = res1_weight*res1 + res1_scale*v;  if res2: v = res2_weight*res2 + res2_scale*v
     * (res*: fp32 trunk tensors with cout channels)                                        */
    const void *mask_dev; size_t mask_bytes;
    int mask_units, mask_unit_off;
    float mask_slope;
    const float *res1_dev; float res1_weight, res1_scale;
    const float *res2_dev; float res2_weight, res2_scale;
    size_t res_bytes;          /* size of each residual / out_f32 tensor                   */
    /* outputs (any subset, at least one) */
    void *out_act_dev;         /* blocked, dtype storage, written at out_unit_off          */
    size_t out_act_bytes;
    int out_units, out_unit_off;
    float *out_f32_dev;        /* fp32 trunk [N][H][cout/4][W][4]                           */
    float *out_nchw_dev;       /* [N][cout][H][W] fp32                                     */
    size_t out_nchw_bytes;
    /* bf16 dtype only: the 16-byte units of the input and of the weights (packed with SR355_PACK_F16,
     * weight_split = SR355_SPLIT_NONE) hold IEEE fp16 and the fp16 MFMA multiplies them; out_act is
     * stored as fp16 (for a following f16_in conv).  No counterpart in the reference: a precision
     * device for the four HR-side convs of the generator's inference forward. */
    int f16_in, f16_out;
    /* bf16 dtype only: the first residual read from blocked bf16 activations (channels [0, cout) of a buffer with
     * res_act_units units per pixel and the output's N, H, W) instead of the fp32 tensor res1_dev, with the same
     * weights: v = res1_weight*res_act + res1_scale*v.  This is how a dense-block skip x + 0.2*conv5(...)
     * (esrgan_model.py:24) runs without an fp32 carrier.  Exclusive with res1_dev. */
    const void *res_act_dev; size_t res_act_bytes;
    int res_act_units;
    /* Stride (0 or 1: stride 1).  2 = nn.Conv2d(c_in, c_out, 3, 2, 1), the discriminator's down-sampling convs
     * (esrgan_model.py:67-84), bf16, unsplit weights, no upsample: the input is in_H x in_W pixels and the output
     * H = (in_H+1)/2, W = (in_W+1)/2; computed at the output positions only. */
    int stride, in_H, in_W;
    /* With upsample = 1: the input (stored at (H+1)/2 x (W+1)/2) is read as its zero-inserted x2 image (its value at
     * even y and x, zero elsewhere) instead of its nearest-neighbour one.  On transposed weights this is the data
     * gradient of a stride-2 conv whose input has H x W pixels (what autograd runs for esrgan_model.py:67-84). */
    int zero_insert;
} sr355_conv_args;

int sr355_conv3x3_fwd(const sr355_conv_args *args, void *stream);

/* ---- a whole dense block in one launch (ResidualDenseBlock.forward, esrgan_model.py:18-24) ----
 * For images at most 32 pixels wide (training patches), bf16, 64 block channels + 4 x 32 growth channels:
 * conv1..conv4 (each followed by LeakyReLU, appended at units 8, 12, 16, 20 of the block buffer) and conv5 with
 * the fused epilogue of sr355_conv3x3_fwd, bit-identical to the five sr355_conv3x3_fwd calls it replaces.  A
 * workgroup owns two image rows and recomputes the halo rows of the growth outputs in LDS.
 * The same entry point runs the five data-gradient convs of the block's backward pass on the transposed packs
 * (sr355_pack_weight_transposed) and the mirrored gradient stack [dOut | dY4 | dY3 | dY2 | dY1]: zero biases,
 * lrelu_slope 1, mask_dev = the block's saved forward buffer with mask_unit_off = {20, 16, 12, 8}. */
typedef struct sr355_dense_block_args {
    int N, H, W;                       /* W <= 32 */
    void *buf_dev; size_t buf_bytes;   /* [N][H][24][W][16 B] bf16; units [0, 8) hold the block input            */
    int store_growth;                  /* 1: steps 1..4 write their outputs to units [8, 24) of buf_dev (needed by
                                          a backward pass); 0: they stay in LDS                                  */
    const void *w_packed_dev[5]; size_t w_packed_bytes[5];   /* packed weights of conv1..conv5                   */
    const float *bias_packed_dev[5];                          /* 32, 32, 32, 32, 64 floats                        */
    float lrelu_slope;                 /* steps 1..4; 1.0f = none                                                */
    const void *mask_dev; size_t mask_bytes;   /* optional sign gate of steps 1..4, as in sr355_conv_args        */
    int mask_units, mask_unit_off[4];
    float mask_slope;
    /* step 5 (conv5, no activation): residuals and outputs as in sr355_conv_args, 64 channels */
    const float *res1_dev; float res1_weight, res1_scale;
    const float *res2_dev; float res2_weight, res2_scale;
    size_t res_bytes;
    const void *res_act_dev; size_t res_act_bytes; int res_act_units;
    void *out_act_dev; size_t out_act_bytes; int out_units, out_unit_off;
    float *out_f32_dev;
} sr355_dense_block_args;
int sr355_dense_block_fwd(const sr355_dense_block_args *args, void *stream);

/* ---- backward pass of the convolution (autograd of esrgan_model.py:10-14 under
 *      loss.backward(), /root/reference/esrgan/train.py:161, :220) ---- */

/* Transposed packing for the data gradient: the forward weight w[cout][cin][3][3] contributes its
 * `cout` output channels as INPUT units [unit_off, unit_off + ceil(cout/unit)) of a packed tensor
 * with `packed_units` units and round_up(n_out, 32) output channels, which produce the forward
 * input channels [ci_lo, ci_lo + n_out):  wt[c][o][tap] = scale * w[c][ci_lo + o][8 - tap].
 * sr355_conv3x3_fwd run on the result (zero bias) IS the data-gradient convolution. */
int sr355_pack_weight_transposed(const float *w_oihw_dev, int cout, int cin, int ci_lo, int n_out,
                                 int dtype, float scale, void *w_packed_dev, int packed_units,
                                 int unit_off, size_t w_packed_bytes, void *stream);
/* Several packings in one launch each (the nine convs of the discriminator, esrgan_model.py:74-84, repack after every
 * optimizer step; one launch per conv and form is eighteen launches of a few microseconds): `jobs[i]` are the arguments
 * of one sr355_pack_weight / sr355_pack_weight_transposed call, the results are those calls'. */
typedef struct sr355_pack_job {
    const float *w_dev, *bias_dev;   /* OIHW fp32 weight, bias (may be NULL: zeros) */
    void *wp_dev; size_t wp_bytes;   /* sr355_packed_weight_bytes(cout, cin, dtype, split) */
    float *bp_dev;                   /* round_up(cout, 32) floats */
    int cout, cin, split;
} sr355_pack_job;
int sr355_pack_weights(const sr355_pack_job *jobs, int n, int dtype, void *stream);
typedef struct sr355_pack_t_job {
    const float *w_dev;
    void *wp_dev; size_t wp_bytes;
    int cout, cin, ci_lo, n_out, packed_units, unit_off;
    float scale;
} sr355_pack_t_job;
int sr355_pack_weights_transposed(const sr355_pack_t_job *jobs, int n, int dtype, void *stream);

/* Adjoint of the nearest x2 upsample fused in front of a conv: out[y][x] = sum of the 2x2 block of
 * `in_dev` ([N][2H][units][2W] blocked), optionally gated by sign(mask) (LeakyReLU backward), written
 * as fp32 trunk and/or blocked activations at out_unit_off. */
int sr355_pool2x2_sum(const void *in_dev, size_t in_bytes, int N, int H, int W, int units, int dtype,
                      const void *mask_dev, size_t mask_bytes, int mask_units, int mask_unit_off,
                      float mask_slope, float *out_f32_dev, size_t out_f32_bytes, void *out_act_dev,
                      size_t out_act_bytes, int out_units, int out_unit_off, void *stream);

/* Weight and bias gradient (bf16 activations):
 *   dw[co][ci][ky][kx] (=|+=) scale * sum_{n,y,x} dy[co][n,y,x] * in[ci][n, y+ky-1, x+kx-1]
 *   db[co]             (=|+=) scale * sum_{n,y,x} dy[co][n,y,x]
 * `in` is the conv's blocked input (upsample=1: stored at half resolution), `dy` the blocked
 * gradient of its output at unit dy_unit_off.  Deterministic: per-workgroup partial slabs in
 * `workspace_dev` are summed in a fixed order. */
typedef struct sr355_wgrad_args {
    int N, H, W;               /* output-resolution batch / height / width */
    int upsample;
    const void *in_dev; size_t in_bytes; int in_units;
    int cin;                   /* real input channels (channels [0, cin) of `in`) */
    const void *dy_dev; size_t dy_bytes; int dy_units, dy_unit_off;
    int cout;                  /* 1 .. 4096 (above 64: 32-channel slices as the jobs of one launch) */
    float scale;
    int accumulate;            /* 0: overwrite dw/db, 1: add into them */
    float *dw_dev; size_t dw_bytes;
    float *db_dev;             /* may be NULL */
    void *workspace_dev; size_t workspace_bytes;
    int dy_stride2;            /* 1: weight gradient of a stride-2 conv whose input has H x W pixels: dy is stored at
                                  (H+1)/2 x (W+1)/2 (no upsample) */
} sr355_wgrad_args;

size_t sr355_wgrad_workspace_bytes(int cout, int cin, int N, int H, int W);
int sr355_conv3x3_wgrad(const sr355_wgrad_args *args, void *stream);

/* The five weight / bias gradients of one dense block (autograd of /root/reference/esrgan/models/esrgan_model.py:18-24
 * under loss.backward(), train.py:161, :220) as ONE batched launch plus one fixed-order reduction: `act` is the block's
 * forward buffer [N][H][units][W][16 B] (x in channels [0, nf), the four growth outputs behind it), `stack` its
 * gradient stack of the same geometry, [dOut (nf) | dY4 | dY3 | dY2 | dY1] (what the block's data-gradient pass
 * leaves behind).  dw_dev[k] / db_dev[k] receive conv(k+1)'s gradients [cout][cin][3][3] / [cout] (overwritten), the
 * growth convs' scaled by growth_scale, conv5's by conv5_scale (the dense skip's 0.2 times any outer factor).
 * bf16, growth_channels <= 32, num_features <= 64, both multiples of 8.  The same kernels and summation order as five
 * sr355_conv3x3_wgrad calls: the results are bit-identical. */
typedef struct sr355_dense_wgrad_args {
    int N, H, W;
    const void *act_dev; size_t act_bytes;
    const void *stack_dev; size_t stack_bytes;
    int units;                 /* units per pixel of both buffers (>= (nf + 4 gc) / 8) */
    int num_features, growth_channels;
    float growth_scale, conv5_scale;
    float *dw_dev[5];
    float *db_dev[5];
    void *workspace_dev; size_t workspace_bytes;
} sr355_dense_wgrad_args;
size_t sr355_dense_block_wgrad_workspace_bytes(int num_features, int growth_channels, int N, int H, int W);
int sr355_dense_block_wgrad(const sr355_dense_wgrad_args *args, void *stream);
/* The same for n_blocks (1 .. SR355_DENSE_WGRAD_MAX_BLOCKS) dense blocks of one shape in ONE launch + one reduction: the
 * blocks of an RRDB (/root/reference/esrgan/models/esrgan_model.py:30-33), or of two or three.  At training-patch sizes one
 * block's launch cannot keep the chip busy (a few hundred workgroups bound by the latency of their tile fetches); nine
 * blocks per launch take 10 us per block where one takes 19.  args[0 .. n_blocks): every block has its own act / stack / dw / db;
 * shape, units and growth_scale must agree; the scratch is args[0].workspace_dev, n_blocks times the one-block size
 * (the other entries' workspace fields are ignored).  Results are bit-identical to n_blocks single calls. */
#define SR355_DENSE_WGRAD_MAX_BLOCKS 9
int sr355_dense_blocks_wgrad(const sr355_dense_wgrad_args *args, int n_blocks, void *stream);

/* The same gradient on the exact-fp32 path (precision="fp32" modules under loss.backward(),
 * /root/reference/esrgan/train.py:161 with autocast off): `in` and `dy` are blocked fp32 tensors [N][H][units][W][4],
 * in_units / dy_units / dy_unit_off count 4-channel units, any cout up to 4096; upsample as above, dy_stride2 must
 * be 0.  v_mfma_f32_32x32x2f32 accumulation, the same fixed-order slab reduction.  Scratch:
 * sr355_wgrad_f32_workspace_bytes(). */
size_t sr355_wgrad_f32_workspace_bytes(int cout, int cin, int N, int H, int W);
int sr355_conv3x3_wgrad_f32(const sr355_wgrad_args *args, void *stream);

/* sr355_conv3x3_wgrad without a workspace: the per-workgroup partial slabs go to a static device array of the
 * library (SR355_STATIC_SCRATCH_BYTES per GPU, held until the process ends); args->workspace_dev / workspace_bytes
 * are ignored, and the call makes no hipMalloc / hipFree or any other allocating call.  No counterpart in the
 * reference: it is for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation).  Shapes whose sr355_wgrad_workspace_bytes() exceeds SR355_STATIC_SCRATCH_BYTES are
 * rejected with SR355_ERR_INVALID.  All calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (two streams, two threads) race; calls on one stream are ordered.  Results are bit-identical to
 * sr355_conv3x3_wgrad's.  The FIRST call on a device also takes the array's address (hipGetSymbolAddress) and sets
 * the kernel's shared-memory attribute: make one un-captured warm-up call per device before calling under stream
 * capture. */
#define SR355_STATIC_SCRATCH_BYTES (16u << 20)
int sr355_conv3x3_wgrad_static(const sr355_wgrad_args *args, void *stream);

/* ---- BatchNorm2d + LeakyReLU behind a conv (the discriminator's conv_block, esrgan_model.py:67-72) ----
 * x_f32 is the conv's fp32 output [N][H][C/4][W][4] (sr355_conv3x3_fwd's out_f32_dev; C a multiple of 8).
 * training = 1: batch statistics over N, H, W (two passes, fixed summation order), running_mean / running_var /
 * num_batches_tracked updated as nn.BatchNorm2d does (momentum, unbiased variance; any of them may be NULL);
 * training = 0: the running statistics.  save_mean / save_rstd [C] receive what the backward pass needs.  The output
 * is blocked bf16 at out_unit_off: y = lrelu(gamma * (x - mean) * rstd + beta). */
size_t sr355_bn_workspace_bytes(int N, int H, int W, int C);
int sr355_bn_lrelu_fwd(const float *x_f32_dev, size_t x_bytes, int N, int H, int W, int C, const float *gamma_dev,
                       const float *beta_dev, float eps, float lrelu_slope, int training, float momentum,
                       float *running_mean_dev, float *running_var_dev, long long *num_batches_tracked_dev,
                       float *save_mean_dev, float *save_rstd_dev, void *out_act_dev, size_t out_act_bytes,
                       int out_units, int out_unit_off, void *workspace_dev, size_t workspace_bytes, void *stream);
/* Its backward pass (autograd of the same lines): dy_act = gradient of the output (blocked bf16; or, with dy_units = 0
 * and dy_unit_off = 0, an fp32 tensor [N][H][C/4][W][4]: what a data-gradient conv writes through out_f32_dev, so that
 * the subtraction of the gradient's mean in here works on unrounded values); dx_act (blocked bf16, the layout the
 * conv's data / weight gradients read) = gradient of x; dgamma / dbeta [C] overwritten. */
int sr355_bn_lrelu_bwd(const float *x_f32_dev, size_t x_bytes, const void *dy_act_dev, size_t dy_bytes, int dy_units,
                       int dy_unit_off, int N, int H, int W, int C, const float *gamma_dev, const float *beta_dev,
                       const float *save_mean_dev, const float *save_rstd_dev, float lrelu_slope, int training,
                       void *dx_act_dev, size_t dx_bytes, int dx_units, int dx_unit_off, float *dgamma_dev,
                       float *dbeta_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
/* The same two passes for a batch that stands for `groups` separate calls of the layer on its `groups` consecutive
 * sub-batches of N / groups images (the discriminator update of /root/reference/esrgan/train.py:200-201 calls the
 * network on the real and on the generated batch: run as one batch, every normalised layer still has to take its
 * statistics per call): statistics, save_mean / save_rstd and dgamma / dbeta are [groups][C], the running buffers are
 * updated once per group in order, and the results are bit-identical to `groups` calls of the entry points above on
 * the slices -- in one set of launches.  workspace: groups * sr355_bn_workspace_bytes(N / groups, H, W, C). */
int sr355_bn_lrelu_fwd_groups(const float *x_f32_dev, size_t x_bytes, int N, int H, int W, int C, const float *gamma_dev,
                              const float *beta_dev, float eps, float lrelu_slope, int training, float momentum,
                              float *running_mean_dev, float *running_var_dev, long long *num_batches_tracked_dev,
                              float *save_mean_dev, float *save_rstd_dev, void *out_act_dev, size_t out_act_bytes,
                              int out_units, int out_unit_off, void *workspace_dev, size_t workspace_bytes, int groups,
                              void *stream);
int sr355_bn_lrelu_bwd_groups(const float *x_f32_dev, size_t x_bytes, const void *dy_act_dev, size_t dy_bytes,
                              int dy_units, int dy_unit_off, int N, int H, int W, int C, const float *gamma_dev,
                              const float *beta_dev, const float *save_mean_dev, const float *save_rstd_dev,
                              float lrelu_slope, int training, void *dx_act_dev, size_t dx_bytes, int dx_units,
                              int dx_unit_off, float *dgamma_dev, float *dbeta_dev, void *workspace_dev,
                              size_t workspace_bytes, int groups, void *stream);

/* nn.PixelShuffle(2) on the blocked layout (the sub-pixel stages of CycleGAN/CycleGAN_w_v1/CycleGAN.py:117-129,
 * condtional VAE/model.py:118-133): src [N][H][4*c_out/c][W][c] -> dst [N][2H][c_out/c][2W][c],
 * dst[n][2y+dy][ch][2x+dx] = src[n][y][4*ch + 2*dy + dx][x]; c = channels per 16-byte unit of `dtype`, c_out a
 * multiple of it.  inverse = 1 runs the other way (src is the 2H x 2W tensor): the gradient of the forward op. */
int sr355_pixel_shuffle2(const void *src_dev, size_t src_bytes, int N, int H, int W, int c_out, int dtype, int inverse,
                         void *dst_dev, size_t dst_bytes, void *stream);

/* ---- nn.MaxPool2d(2, 2) on blocked bf16 (vgg19.features' pooling layers, esrgan_model.py:93-98) ----
 * in [N][H][units][W] -> out [N][H/2][units][W/2] (an odd trailing row / column is dropped).  The backward pass
 * recomputes the argmax from the forward input: the whole gradient goes to the first maximum of a window in row-major
 * order (nn.MaxPool2d's rule); dx is written in full. */
int sr355_maxpool2x2_fwd(const void *in_act_dev, size_t in_bytes, int N, int H, int W, int units, void *out_act_dev,
                         size_t out_bytes, void *stream);
int sr355_maxpool2x2_bwd(const void *in_act_dev, size_t in_bytes, const void *dy_act_dev, size_t dy_bytes, int N, int H,
                         int W, int units, void *dx_act_dev, size_t dx_bytes, void *stream);

/* ---- nn.L1Loss / nn.MSELoss (mean reduction) of the training iteration (train.py:114-115, :160, :201-204, :214-219)
 * mode 0: mean |a - b|; mode 1: mean (a - b)^2 over n fp32 elements; b_dev == NULL: the constant b_const (the
 * ones / zeros targets of the adversarial terms).  grad_dev (optional) receives d(loss)/da in the same pass, so the
 * backward pass is one scale by the incoming gradient.  Deterministic (per-workgroup partials, fixed order). */
size_t sr355_loss_workspace_bytes(size_t n);
int sr355_loss_fwd(const float *a_dev, const float *b_dev, float b_const, size_t n, int mode, float *loss_dev,
                   float *grad_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- whole generator (GeneratorRRDB.forward, esrgan_model.py:55-62) ---- */
typedef struct sr355_generator_cfg {
    int in_channels, out_channels, num_features, num_blocks, growth_channels; /* :42 */
    int dtype;
    int trunk_fp32;            /* 1: every skip (dense block, RRDB, global) adds onto an fp32 carrier kept
                                  beside the bf16 activations.  SR355_TRUNK_RRDB (2), bf16 inference only:
                                  the dense-block skips read the bf16 activations themselves and only the
                                  RRDB and global skips keep fp32 carriers (41 % fewer epilogue bytes per
                                  RRDB for about 3 dB of PSNR)                                             */
    int hr_split;              /* bf16 only.  1: SPLIT_INPUT on conv_first and SPLIT_WEIGHT on the
                                  four HR-side convs (upconv1/2, HR_conv, conv_last).
                                  SR355_HR_F16 (2), inference only: SPLIT_INPUT on conv_first and
                                  the four HR-side convs in IEEE fp16 (weights and the activations
                                  they read: 11 significant bits instead of 8 at the same MFMA rate,
                                  no second weight image); the training entry points refuse it   */
} sr355_generator_cfg;
#define SR355_HR_F16 2
#define SR355_TRUNK_RRDB 2

/* Number of 3x3 convs (351 for the default config). */
int sr355_generator_num_convs(const sr355_generator_cfg *cfg);
/* Bytes of the packed parameter blob (all packed weights + padded biases). */
size_t sr355_generator_packed_bytes(const sr355_generator_cfg *cfg);
/* w_ptrs/b_ptrs: device pointers to the OIHW fp32 weights / biases in state_dict order. */
int sr355_generator_pack(const sr355_generator_cfg *cfg, const float *const *w_ptrs,
                         const float *const *b_ptrs, int n_convs, void *packed_dev,
                         size_t packed_bytes, void *stream);
size_t sr355_generator_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W);
/* in [N][in_channels][H][W] fp32 -> out [N][out_channels][4H][4W] fp32 (unclamped). */
int sr355_generator_forward(const sr355_generator_cfg *cfg, const void *packed_dev,
                            size_t packed_bytes, const float *in_nchw_dev, float *out_nchw_dev,
                            int N, int H, int W, void *workspace_dev, size_t workspace_bytes,
                            void *stream);

/* The same forward between image buffers, as the reference's inference script drives it
 * (/root/reference/esrgan/test.py:68-74): in uint8 [N][H][W][in_channels] (PIL / numpy HWC layout) read as
 * byte / 255 (torchvision ToTensor, esrgan/dataset.py:67-68); out uint8 [N][4H][4W][out_channels] =
 * trunc(255 * clamp(y, 0, 1)) (`.clamp(0, 1)` then ToPILImage's mul(255).byte()).  The conversions run inside
 * the first layout kernel and conv_last's epilogue; no fp32 image is materialised. */
int sr355_generator_forward_u8(const sr355_generator_cfg *cfg, const void *packed_dev,
                               size_t packed_bytes, const unsigned char *in_u8hwc_dev,
                               unsigned char *out_u8hwc_dev, int N, int H, int W, void *workspace_dev,
                               size_t workspace_bytes, void *stream);

/* Measurement hook for the persistent launch that runs the body of sr355_generator_forward / _forward_u8 (every dense
 * block and trunk_conv as one kernel, taken for batches / images of at least 128 tiles of 16 x 32 LR pixels; the
 * module of /root/reference/esrgan/models/esrgan_model.py:55-58 has no counterpart: there every conv is a launch).
 * sr355_chain_timing(1) makes the following forwards record a HIP event on the launch stream right before and right
 * after that kernel; sr355_chain_last_ms() waits for the later event of the most recent such forward on the current
 * device and returns the kernel's duration in milliseconds (a negative value if none was recorded). */
void sr355_chain_timing(int on);
float sr355_chain_last_ms(void);

/* Health of that persistent launch (again without a counterpart in the reference, whose convs are separate launches,
 * esrgan_model.py:55-58).  Its workgroups wait for each other and must all be resident; where they cannot be (a
 * second process on the GPU, masked CUs) a wait gives up after two seconds, the forward's output is INVALID and the
 * kernel raises a word in pinned host memory that stays raised until it has been reported (a failed launch is not
 * forgotten because later launches succeeded).  sr355_chain_status(wait) answers for the forwards issued so far on the
 * current device: SR355_OK, SR355_CHAIN_PENDING (wait == 0 only: the most recent launch has not finished and none has
 * failed yet) or, once per failure, SR355_ERR_HIP with the reason in sr355_last_error().  wait != 0 blocks until the
 * most recent launch has completed: call it before consuming the output of a one-shot forward.  The next forward call
 * makes the same check without blocking.  After a failure the device runs the body as one launch per conv (slower,
 * no residency requirement) until sr355_chain_reenable(). */
#define SR355_CHAIN_PENDING 1
int sr355_chain_status(int wait);
void sr355_chain_reenable(void);

/* One Adam update over flat fp32 device buffers of n elements, in one launch: the arithmetic of
 * torch.optim.Adam(lr, betas) as the reference configures it (/root/reference/esrgan/train.py:117-122; no amsgrad,
 * weight_decay added to the gradient when non-zero).  `step` is the 1-based count of this update (bias corrections
 * 1 - beta^step are taken on the host in double).  The generator backward already delivers every gradient as one
 * flat buffer in state_dict order; with the parameters kept in one buffer of the same order, the optimizer tail
 * of a training iteration is this single call instead of a pass over 702 tensors. */
int sr355_adam_step(float *param_dev, const float *grad_dev, float *exp_avg_dev, float *exp_avg_sq_dev, size_t n,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                    void *stream);

/* ---- training: forward that keeps activations, and the backward pass (bf16 kernels only) ----
 * Replaces what autograd records for GeneratorRRDB.forward and replays under loss.backward()
 * (/root/reference/esrgan/train.py:159-161, :212-220). */
size_t sr355_generator_saved_bytes(const sr355_generator_cfg *cfg, int N, int H, int W);
size_t sr355_generator_train_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W);
/* As sr355_generator_forward; `saved_dev` receives the activations the backward pass reads. */
int sr355_generator_forward_train(const sr355_generator_cfg *cfg, const void *packed_dev,
                                  size_t packed_bytes, const float *in_nchw_dev, float *out_nchw_dev,
                                  int N, int H, int W, void *saved_dev, size_t saved_bytes,
                                  void *workspace_dev, size_t workspace_bytes, void *stream);
/* Transposed (data-gradient) form of every conv weight, from the same OIHW fp32 tensors. */
size_t sr355_generator_packed_bwd_bytes(const sr355_generator_cfg *cfg);
int sr355_generator_pack_bwd(const sr355_generator_cfg *cfg, const float *const *w_ptrs, int n_convs,
                             void *packed_bwd_dev, size_t packed_bwd_bytes, void *stream);
size_t sr355_generator_bwd_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W);
/* Number of floats of the flat gradient buffer: per conv in state_dict order, weight then bias. */
size_t sr355_generator_grads_floats(const sr355_generator_cfg *cfg);
/* grad_out [N][out][4H][4W] fp32 -> grads_dev (every parameter gradient, overwritten) and, when
 * grad_in_nchw_dev is not NULL, the gradient of the input [N][in][H][W]. */
int sr355_generator_backward(const sr355_generator_cfg *cfg, const void *packed_bwd_dev,
                             size_t packed_bwd_bytes, const void *saved_dev, size_t saved_bytes,
                             const float *grad_out_nchw_dev, float *grads_dev, size_t grads_floats,
                             float *grad_in_nchw_dev, int N, int H, int W, void *workspace_dev,
                             size_t workspace_bytes, void *stream);

/* The backward pass in ordered phases, so that a data-parallel host can start the all-reduce of the
 * gradients a phase has finished while later phases still compute (no counterpart in the reference,
 * whose nn.DataParallel reduces after the whole backward: /root/reference/esrgan/train.py:108-111).
 * Phase 0 = conv_last .. trunk_conv, phase p in [1, num_blocks] = RRDB (num_blocks - p),
 * phase num_blocks + 1 = conv_first; a phase writes exactly the gradients of its own convs.
 * Phases must run in increasing order with the same buffers. */
int sr355_generator_backward_phases(const sr355_generator_cfg *cfg);
int sr355_generator_backward_range(const sr355_generator_cfg *cfg, const void *packed_bwd_dev,
                                   size_t packed_bwd_bytes, const void *saved_dev, size_t saved_bytes,
                                   const float *grad_out_nchw_dev, float *grads_dev, size_t grads_floats,
                                   float *grad_in_nchw_dev, int N, int H, int W, void *workspace_dev,
                                   size_t workspace_bytes, int phase_begin, int phase_end, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SR355_H */
