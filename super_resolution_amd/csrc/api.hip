// C ABI of libsr355.so (declared in include/sr355.h) and the generator schedule.
//
// The schedule in sr355_generator_forward follows GeneratorRRDB.forward
// (/root/reference/esrgan/models/esrgan_model.py:55-62): conv_first, num_blocks x RRDB
// (3 dense blocks of 5 convs, :18-24, :35-39), trunk_conv + global skip, two
// nearest-x2 + conv + LeakyReLU stages, HR_conv + LeakyReLU, conv_last.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "sr355_internal.h"

namespace sr355 {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return SR355_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return SR355_ERR_HIP;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- generator geometry -------------------------------------------------------------------
struct ConvSpec {
    int cin, cout, split;
    size_t w_off, b_off;  // offsets into the packed blob
    int f16 = 0;          // weights packed as fp16, input read as fp16 (HR side of an inference-only pack)
};

static bool cfg_ok(const sr355_generator_cfg *c) {
    if (!c) {
        set_error("generator cfg is null");
        return false;
    }
    if (unit_channels(c->dtype) < 0) {
        set_error("generator cfg: unknown dtype %d", c->dtype);
        return false;
    }
    if (c->in_channels < 1 || c->out_channels < 1 || c->num_blocks < 0 || c->num_features < 16 ||
        c->growth_channels < 16 || (c->num_features % 16) || (c->growth_channels % 16)) {
        set_error("generator cfg: need in/out channels >= 1, num_blocks >= 0 and num_features / "
                  "growth_channels positive multiples of 16 (got in=%d out=%d nf=%d nb=%d gc=%d)",
                  c->in_channels, c->out_channels, c->num_features, c->num_blocks, c->growth_channels);
        return false;
    }
    return true;
}

// Convs in state_dict order (constructor order of esrgan_model.py:42-53).
static std::vector<ConvSpec> conv_specs(const sr355_generator_cfg &c, size_t *total_bytes) {
    std::vector<ConvSpec> v;
    const int nf = c.num_features, gc = c.growth_channels;
    const bool split = c.hr_split && c.dtype == SR355_DTYPE_BF16;
    const bool hr16 = c.hr_split == SR355_HR_F16 && c.dtype == SR355_DTYPE_BF16;
    const int hs = (split && !hr16) ? SR355_SPLIT_WEIGHT : SR355_SPLIT_NONE;
    v.push_back({c.in_channels, nf, split ? SR355_SPLIT_INPUT : SR355_SPLIT_NONE, 0, 0});
    for (int b = 0; b < c.num_blocks; ++b)
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 5; ++k) v.push_back({nf + gc * k, k < 4 ? gc : nf, SR355_SPLIT_NONE, 0, 0});
    v.push_back({nf, nf, SR355_SPLIT_NONE, 0, 0});              // trunk_conv
    for (int i = 0; i < 3; ++i) v.push_back({nf, nf, hs, 0, 0});  // upconv1, upconv2, HR_conv
    v.push_back({nf, c.out_channels, hs, 0, 0});                 // conv_last
    if (hr16)
        for (size_t i = v.size() - 4; i < v.size(); ++i) v[i].f16 = 1;
    size_t off = 0;
    for (auto &s : v) {
        s.w_off = off;
        off += align256(sr355_packed_weight_bytes(s.cout, s.cin, c.dtype, s.split));
        s.b_off = off;
        off += align256((size_t)cout_pad_for(s.cout) * sizeof(float));
    }
    if (total_bytes) *total_bytes = off;
    return v;
}

struct Workspace {
    size_t x0, bufA, bufB, T, R, F, L1, H1, H2, H3, chain_flags, chain_table, total;
    size_t chain_flag_bytes;
    int in_units, nf_units, blk_units, f32_units;
};

static Workspace carve(const sr355_generator_cfg &c, int N, int H, int W) {
    Workspace w;
    const int cpu = unit_channels(c.dtype);
    const size_t P = (size_t)N * H * W;
    w.in_units = real_units_for(c.in_channels, c.dtype,
                                (c.hr_split && c.dtype == SR355_DTYPE_BF16) ? SR355_SPLIT_INPUT : SR355_SPLIT_NONE);
    w.nf_units = c.num_features / cpu;
    w.blk_units = (c.num_features + 4 * c.growth_channels) / cpu;
    w.f32_units = c.num_features / 4;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align256(bytes);
        return o;
    };
    w.x0 = take(P * w.in_units * 16);
    w.bufA = take(P * w.blk_units * 16);
#ifdef SR355_EXP_ALIAS   /* timing experiments only: wrong results */
    w.bufB = w.bufA;
#else
    w.bufB = take(P * w.blk_units * 16);
#endif
    w.T = take(P * c.num_features * 4);
    w.R = take(P * c.num_features * 4);
    w.F = take(P * c.num_features * 4);
    w.L1 = take(P * w.nf_units * 16);
    w.H1 = take(4 * P * w.nf_units * 16);
    w.H2 = take(16 * P * w.nf_units * 16);
    w.H3 = take(16 * P * w.nf_units * 16);
    // the chain kernel's per-tile pass counters (+ its error word; a block of its own, a multiple of 16 bytes) and its
    // table of passes: six per dense block, two for trunk_conv
    w.chain_flag_bytes = ((size_t)N * ((H + CHAIN_TILE_ROWS - 1) / CHAIN_TILE_ROWS) * ((W + 31) / 32) + 1 + 3) / 4 * 16;
    w.chain_flags = take(w.chain_flag_bytes);
    w.chain_table = take((size_t)(18 * c.num_blocks + 2) * sizeof(ChainPass));
    w.total = off;
    return w;
}

static ConvArgs conv_args_zero() {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.res1_w = a.res2_w = 1.f;
    a.slope = 1.f;
    a.mask_slope = 1.f;
    return a;
}

}  // namespace sr355

using namespace sr355;

extern "C" {

int sr355_abi_version(void) { return SR355_ABI_VERSION; }
const char *sr355_last_error(void) { return g_err; }
int sr355_unit_channels(int dtype) { return unit_channels(dtype); }

int sr355_nchw_to_blocked(const float *src, int N, int C, int H, int W, int dtype, void *dst, int dst_units,
                          int split, size_t dst_bytes, void *stream) {
    const int cpu = unit_channels(dtype);
    if (cpu < 0 || !src || !dst || N < 1 || C < 1 || H < 1 || W < 1 || dst_units * cpu < (split ? 2 * C : C) ||
        (split && dtype != SR355_DTYPE_BF16)) {
        set_error("nchw_to_blocked: bad arguments (dtype=%d N=%d C=%d H=%d W=%d units=%d)", dtype, N, C, H, W,
                  dst_units);
        return SR355_ERR_INVALID;
    }
    if (!aligned16(dst) || dst_bytes < (size_t)N * H * dst_units * W * 16) {
        set_error("nchw_to_blocked: destination too small or not 16-byte aligned");
        return SR355_ERR_INVALID;
    }
    return launch_nchw_to_blocked(src, N, C, H, W, dtype, dst, dst_units, split ? 1 : 0, (hipStream_t)stream);
}

int sr355_blocked_to_nchw(const void *src, int src_units, int unit_off, int dtype, int N, int C, int H, int W,
                          float *dst, size_t dst_bytes, void *stream) {
    const int cpu = unit_channels(dtype);
    if (cpu < 0 || !src || !dst || N < 1 || C < 1 || H < 1 || W < 1 || unit_off < 0 ||
        (unit_off + (C + cpu - 1) / cpu) > src_units) {
        set_error("blocked_to_nchw: bad arguments (dtype=%d N=%d C=%d H=%d W=%d units=%d off=%d)", dtype, N, C,
                  H, W, src_units, unit_off);
        return SR355_ERR_INVALID;
    }
    if (dst_bytes < (size_t)N * C * H * W * sizeof(float)) {
        set_error("blocked_to_nchw: destination too small");
        return SR355_ERR_INVALID;
    }
    return launch_blocked_to_nchw(src, src_units, unit_off, dtype, N, C, H, W, dst, (hipStream_t)stream);
}

static bool split_ok(int dtype, int split) {
    return split == SR355_SPLIT_NONE ||
           (dtype == SR355_DTYPE_BF16 && (split == SR355_SPLIT_WEIGHT || split == SR355_SPLIT_INPUT));
}
// sr355_pack_weight also takes SR355_PACK_F16 (one image, fp16 values)
static bool pack_mode_ok(int dtype, int split) {
    return split_ok(dtype, split) || (dtype == SR355_DTYPE_BF16 && split == SR355_PACK_F16);
}

size_t sr355_packed_weight_bytes(int cout, int cin, int dtype, int split) {
    if (unit_channels(dtype) < 0 || cout < 1 || cin < 1 || !pack_mode_ok(dtype, split)) return 0;
    if (split == SR355_PACK_F16) split = SR355_SPLIT_NONE;
    return (size_t)packed_units_for(cin, dtype, split) * 9 * cout_pad_for(cout) * 16;
}

int sr355_pack_weight(const float *w, const float *bias, int cout, int cin, int dtype, int split, void *wp,
                      size_t wp_bytes, float *bp, void *stream) {
    const size_t need = sr355_packed_weight_bytes(cout, cin, dtype, split);
    if (!need || !w || !wp || !aligned16(wp) || wp_bytes < need) {
        set_error("pack_weight: bad arguments (cout=%d cin=%d dtype=%d bytes=%zu need=%zu)", cout, cin, dtype,
                  wp_bytes, need);
        return SR355_ERR_INVALID;
    }
    return launch_pack_weight(w, bias, cout, cin, dtype, split, wp, bp, (hipStream_t)stream);
}

int sr355_pack_weights(const sr355_pack_job *jobs, int n, int dtype, void *stream) {
    if (!jobs || n < 1 || n > 4096) {
        set_error("pack_weights: jobs is null or n=%d outside [1, 4096]", n);
        return SR355_ERR_INVALID;
    }
    std::vector<PackJob> v((size_t)n);
    for (int i = 0; i < n; ++i) {
        const sr355_pack_job &j = jobs[i];
        const size_t need = sr355_packed_weight_bytes(j.cout, j.cin, dtype, j.split);
        if (!need || !j.w_dev || !j.wp_dev || !aligned16(j.wp_dev) || j.wp_bytes < need) {
            set_error("pack_weights: job %d: bad arguments (cout=%d cin=%d dtype=%d bytes=%zu need=%zu)", i, j.cout, j.cin,
                      dtype, j.wp_bytes, need);
            return SR355_ERR_INVALID;
        }
        const bool f16 = dtype == SR355_DTYPE_BF16 && j.split == SR355_PACK_F16;
        v[i] = PackJob{j.w_dev, j.bias_dev, j.wp_dev, j.bp_dev, j.cout, j.cin, f16 ? 0 : j.split, f16 ? 1 : 0};
    }
    return launch_pack_weight_batch(v.data(), n, dtype, (hipStream_t)stream);
}

int sr355_conv3x3_fwd(const sr355_conv_args *p, void *stream) {
    if (!p) {
        set_error("conv3x3_fwd: args is null");
        return SR355_ERR_INVALID;
    }
    const int cpu = unit_channels(p->dtype);
    if (cpu < 0) {
        set_error("conv3x3_fwd: unknown dtype %d", p->dtype);
        return SR355_ERR_INVALID;
    }
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->cout < 1 || (p->upsample != 0 && p->upsample != 1)) {
        set_error("conv3x3_fwd: bad shape N=%d H=%d W=%d cout=%d upsample=%d", p->N, p->H, p->W, p->cout,
                  p->upsample);
        return SR355_ERR_INVALID;
    }
    if (p->upsample && !p->zero_insert && ((p->H | p->W) & 1)) {
        set_error("conv3x3_fwd: fused x2 upsample needs even output H, W (got %d x %d)", p->H, p->W);
        return SR355_ERR_INVALID;
    }
    const int stride = p->stride ? p->stride : 1;
    if ((stride != 1 && stride != 2) || (p->zero_insert && (!p->upsample || stride != 1)) ||
        (stride == 2 && (p->upsample || p->dtype != SR355_DTYPE_BF16 || p->weight_split != SR355_SPLIT_NONE || p->f16_in ||
                         p->in_H < 1 || p->in_W < 1 || p->H != (p->in_H + 1) / 2 || p->W != (p->in_W + 1) / 2))) {
        set_error("conv3x3_fwd: stride must be 1 or 2; stride 2 takes a plain bf16 in_H x in_W input with H = (in_H+1)/2, "
                  "W = (in_W+1)/2 and no upsample; zero_insert goes with upsample = 1 and stride 1 (got stride=%d "
                  "upsample=%d zero_insert=%d in %dx%d out %dx%d)", stride, p->upsample, p->zero_insert, p->in_H, p->in_W,
                  p->H, p->W);
        return SR355_ERR_INVALID;
    }
    const int Hin = stride == 2 ? p->in_H : (p->H + (p->zero_insert ? 1 : 0)) >> p->upsample;
    const int Win = stride == 2 ? p->in_W : (p->W + (p->zero_insert ? 1 : 0)) >> p->upsample;
    if (p->cin_units < 2 || (p->cin_units & 1) || p->cin_units > p->in_units) {
        set_error("conv3x3_fwd: cin_units=%d must be even and within in_units=%d", p->cin_units, p->in_units);
        return SR355_ERR_INVALID;
    }
    const size_t img_bytes = (size_t)Hin * p->in_units * Win * 16;
    if (img_bytes >= ((size_t)1 << 31)) {
        set_error("conv3x3_fwd: one input image is %zu bytes; the kernel indexes an image with 31 bits",
                  img_bytes);
        return SR355_ERR_UNSUPPORTED;
    }
    {   // the epilogue addresses one image of every output / residual tensor with a 31-bit offset as well
        const size_t px = (size_t)p->H * p->W;
        const size_t out_img = px * (size_t)(p->out_act_dev ? p->out_units : 0) * 16;
        const size_t f32_img = px * (size_t)p->cout * 4;
        const size_t ra_img = px * (size_t)(p->res_act_dev ? p->res_act_units : 0) * 16;
        const size_t mk_img = px * (size_t)(p->mask_dev ? p->mask_units : 0) * 16;
        if (out_img >= ((size_t)1 << 31) || f32_img >= ((size_t)1 << 31) || ra_img >= ((size_t)1 << 31) ||
            mk_img >= ((size_t)1 << 31)) {
            set_error("conv3x3_fwd: one output / residual / mask image exceeds 2 GiB; the kernel indexes an image with 31 bits");
            return SR355_ERR_UNSUPPORTED;
        }
    }
    if (!p->in_dev || !aligned16(p->in_dev) || p->in_bytes < img_bytes * p->N) {
        set_error("conv3x3_fwd: input buffer missing, misaligned or too small (%zu < %zu)", p->in_bytes,
                  img_bytes * p->N);
        return SR355_ERR_INVALID;
    }
    const int cout_pad = cout_pad_for(p->cout);
    if (!p->w_packed_dev || !aligned16(p->w_packed_dev) || !p->bias_packed_dev || !aligned16(p->bias_packed_dev) ||
        !split_ok(p->dtype, p->weight_split) ||
        p->w_packed_bytes < (size_t)p->cin_units * (p->weight_split ? 2 : 1) * 9 * cout_pad * 16) {
        set_error("conv3x3_fwd: packed weights/bias missing, misaligned, too small or bad split mode");
        return SR355_ERR_INVALID;
    }
    if (p->mask_dev) {
        const int units = (p->cout + cpu - 1) / cpu;
        if (p->mask_unit_off < 0 || p->mask_unit_off + units > p->mask_units || !aligned16(p->mask_dev) ||
            p->mask_bytes < (size_t)p->N * p->H * p->mask_units * p->W * 16 || (p->cout & 3)) {
            set_error("conv3x3_fwd: mask geometry invalid (off=%d units=%d of %d)", p->mask_unit_off, units,
                      p->mask_units);
            return SR355_ERR_INVALID;
        }
    }
    if (!p->out_act_dev && !p->out_f32_dev && !p->out_nchw_dev) {
        set_error("conv3x3_fwd: no output requested");
        return SR355_ERR_INVALID;
    }
    const bool trunk = p->res1_dev || p->res2_dev || p->out_f32_dev;
    if ((trunk || p->out_act_dev) && (p->cout & 3)) {
        set_error("conv3x3_fwd: cout=%d must be a multiple of 4 for blocked outputs / residuals", p->cout);
        return SR355_ERR_INVALID;
    }
    if (trunk) {
        const size_t need = (size_t)p->N * p->H * p->W * p->cout * 4;
        if (p->res_bytes < need || !aligned16(p->res1_dev) || !aligned16(p->res2_dev) ||
            !aligned16(p->out_f32_dev)) {
            set_error("conv3x3_fwd: residual / fp32 output tensors misaligned or too small (%zu < %zu)",
                      p->res_bytes, need);
            return SR355_ERR_INVALID;
        }
    }
    if (p->out_act_dev) {
        const int units = (p->cout + cpu - 1) / cpu;
        const size_t need = (size_t)p->N * p->H * p->out_units * p->W * 16;
        if (p->out_unit_off < 0 || p->out_unit_off + units > p->out_units || !aligned16(p->out_act_dev) ||
            p->out_act_bytes < need) {
            set_error("conv3x3_fwd: activation output geometry invalid (off=%d units=%d of %d, %zu < %zu bytes)",
                      p->out_unit_off, units, p->out_units, p->out_act_bytes, need);
            return SR355_ERR_INVALID;
        }
    }
    if (p->out_nchw_dev && p->out_nchw_bytes < (size_t)p->N * p->cout * p->H * p->W * 4) {
        set_error("conv3x3_fwd: NCHW output too small");
        return SR355_ERR_INVALID;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const unsigned char *)p->in_dev;
    a.w = (const unsigned char *)p->w_packed_dev;
    a.bias = p->bias_packed_dev;
    a.out_act = (unsigned char *)p->out_act_dev;
    a.out_f32 = p->out_f32_dev;
    a.out_nchw = p->out_nchw_dev;
    if ((p->f16_in || p->f16_out) && (p->dtype != SR355_DTYPE_BF16 || (p->f16_in && p->weight_split != SR355_SPLIT_NONE))) {
        set_error("conv3x3_fwd: f16_in / f16_out exist on the bf16 path only, f16_in with unsplit weights");
        return SR355_ERR_INVALID;
    }
    a.f16_in = p->f16_in ? 1 : 0;
    a.f16_out = p->f16_out ? 1 : 0;
    if (p->res_act_dev) {
        const size_t need = (size_t)p->N * p->H * p->W * p->res_act_units * 16;
        if (p->dtype != SR355_DTYPE_BF16 || p->res1_dev || !aligned16(p->res_act_dev) ||
            p->res_act_units * 8 < p->cout || p->res_act_bytes < need) {
            set_error("conv3x3_fwd: res_act needs the bf16 path, no res1, >= cout channels and %zu bytes (got %zu)", need,
                      p->res_act_bytes);
            return SR355_ERR_INVALID;
        }
        a.res_act = (const unsigned char *)p->res_act_dev;
        a.res_act_units = p->res_act_units;
    }
    a.res1 = p->res1_dev;
    a.res2 = p->res2_dev;
    a.res1_scale = p->res1_scale;
    a.res2_scale = p->res2_scale;
    a.res1_w = p->res1_weight;
    a.res2_w = p->res2_weight;
    a.mask_act = (const unsigned char *)p->mask_dev;
    a.mask_units = p->mask_units;
    a.mask_unit_off = p->mask_unit_off;
    a.mask_slope = p->mask_slope;
    a.slope = p->lrelu_slope;
    a.N = p->N;
    a.H = p->H;
    a.W = p->W;
    a.in_units = p->in_units;
    a.ups = p->upsample;
    a.zins = p->zero_insert ? 1 : 0;
    a.stride = stride;
    a.in_H = Hin, a.in_W = Win;
    a.out_units = p->out_units;
    a.out_unit_off = p->out_unit_off;
    a.cout_real = p->cout;
    a.cout_pad = cout_pad;
    a.f32_units = p->cout / 4;
    return launch_conv3x3(a, p->dtype, p->cin_units, p->weight_split != SR355_SPLIT_NONE, (hipStream_t)stream);
}

int sr355_dense_block_fwd(const sr355_dense_block_args *p, void *stream) {
    if (!p) {
        set_error("dense_block_fwd: args is null");
        return SR355_ERR_INVALID;
    }
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->W > 32) {
        set_error("dense_block_fwd: bad shape N=%d H=%d W=%d (images at most 32 pixels wide)", p->N, p->H, p->W);
        return p->W > 32 ? SR355_ERR_UNSUPPORTED : SR355_ERR_INVALID;
    }
    const size_t px = (size_t)p->N * p->H * p->W;
    if (!p->buf_dev || !aligned16(p->buf_dev) || p->buf_bytes < px * 24 * 16) {
        set_error("dense_block_fwd: block buffer missing, misaligned or smaller than %zu bytes", px * 24 * 16);
        return SR355_ERR_INVALID;
    }
    for (int k = 0; k < 5; ++k) {
        const size_t need = (size_t)(8 + 4 * k) * 9 * (k < 4 ? 32 : 64) * 16;
        if (!p->w_packed_dev[k] || !aligned16(p->w_packed_dev[k]) || p->w_packed_bytes[k] < need ||
            !p->bias_packed_dev[k] || !aligned16(p->bias_packed_dev[k])) {
            set_error("dense_block_fwd: packed weights / bias of conv%d missing, misaligned or smaller than %zu bytes",
                      k + 1, need);
            return SR355_ERR_INVALID;
        }
    }
    if (p->mask_dev) {
        bool ok = aligned16(p->mask_dev) && p->mask_bytes >= px * p->mask_units * 16;
        for (int k = 0; k < 4; ++k) ok = ok && p->mask_unit_off[k] >= 0 && p->mask_unit_off[k] + 4 <= p->mask_units;
        if (!ok) {
            set_error("dense_block_fwd: mask geometry invalid");
            return SR355_ERR_INVALID;
        }
    }
    if (!p->out_act_dev && !p->out_f32_dev) {
        set_error("dense_block_fwd: no output requested");
        return SR355_ERR_INVALID;
    }
    if ((p->res1_dev || p->res2_dev || p->out_f32_dev) &&
        (p->res_bytes < px * 64 * 4 || !aligned16(p->res1_dev) || !aligned16(p->res2_dev) || !aligned16(p->out_f32_dev))) {
        set_error("dense_block_fwd: residual / fp32 output tensors misaligned or smaller than %zu bytes", px * 64 * 4);
        return SR355_ERR_INVALID;
    }
    if (p->res_act_dev && (p->res1_dev || !aligned16(p->res_act_dev) || p->res_act_units < 8 ||
                           p->res_act_bytes < px * p->res_act_units * 16)) {
        set_error("dense_block_fwd: res_act needs no res1, >= 8 units and %zu bytes", px * p->res_act_units * 16);
        return SR355_ERR_INVALID;
    }
    if (p->out_act_dev && (p->out_unit_off < 0 || p->out_unit_off + 8 > p->out_units || !aligned16(p->out_act_dev) ||
                           p->out_act_bytes < px * p->out_units * 16)) {
        set_error("dense_block_fwd: activation output geometry invalid");
        return SR355_ERR_INVALID;
    }
    DenseArgs d;
    memset(&d, 0, sizeof(d));
    d.buf = (const unsigned char *)p->buf_dev;
    d.growth_out = p->store_growth ? (unsigned char *)p->buf_dev : nullptr;
    for (int k = 0; k < 5; ++k) {
        d.w[k] = (const unsigned char *)p->w_packed_dev[k];
        d.bias[k] = p->bias_packed_dev[k];
    }
    d.mask = (const unsigned char *)p->mask_dev;
    d.mask_units = p->mask_units;
    for (int k = 0; k < 4; ++k) d.mask_off[k] = p->mask_unit_off[k];
    d.mask_slope = p->mask_slope;
    d.slope = p->lrelu_slope;
    d.N = p->N, d.H = p->H, d.W = p->W, d.units = 24;
    ConvArgs &a = d.tail;
    a = conv_args_zero();
    a.res1 = p->res1_dev, a.res1_w = p->res1_weight, a.res1_scale = p->res1_scale;
    a.res2 = p->res2_dev, a.res2_w = p->res2_weight, a.res2_scale = p->res2_scale;
    a.res_act = (const unsigned char *)p->res_act_dev;
    a.res_act_units = p->res_act_units;
    a.out_act = (unsigned char *)p->out_act_dev;
    a.out_units = p->out_units, a.out_unit_off = p->out_unit_off;
    a.out_f32 = p->out_f32_dev;
    a.cout_real = a.cout_pad = 64;
    a.f32_units = 16;
    return launch_dense_block(d, (hipStream_t)stream);
}

// ---- backward primitives ------------------------------------------------------------------
int sr355_pack_weight_transposed(const float *w, int cout, int cin, int ci_lo, int n_out, int dtype, float scale,
                                 void *wp, int packed_units, int unit_off, size_t wp_bytes, void *stream) {
    const int cpu = unit_channels(dtype);
    if (cpu < 0 || !w || !wp || !aligned16(wp) || cout < 1 || cin < 1 || ci_lo < 0 || n_out < 1 ||
        ci_lo + n_out > cin || unit_off < 0 || unit_off + (cout + cpu - 1) / cpu > packed_units ||
        wp_bytes < (size_t)packed_units * 9 * cout_pad_for(n_out) * 16) {
        set_error("pack_weight_transposed: bad arguments (cout=%d cin=%d ci_lo=%d n_out=%d units=%d off=%d)", cout,
                  cin, ci_lo, n_out, packed_units, unit_off);
        return SR355_ERR_INVALID;
    }
    return launch_pack_weight_t(w, cout, cin, ci_lo, n_out, dtype, scale, wp, unit_off, cout_pad_for(n_out),
                                (hipStream_t)stream);
}

int sr355_pack_weights_transposed(const sr355_pack_t_job *jobs, int n, int dtype, void *stream) {
    const int cpu = unit_channels(dtype);
    if (!jobs || n < 1 || n > 4096 || cpu < 0) {
        set_error("pack_weights_transposed: jobs is null, n=%d outside [1, 4096] or unknown dtype %d", n, dtype);
        return SR355_ERR_INVALID;
    }
    std::vector<PackTJob> v((size_t)n);
    for (int i = 0; i < n; ++i) {
        const sr355_pack_t_job &j = jobs[i];
        if (!j.w_dev || !j.wp_dev || !aligned16(j.wp_dev) || j.cout < 1 || j.cin < 1 || j.ci_lo < 0 || j.n_out < 1 ||
            j.ci_lo + j.n_out > j.cin || j.unit_off < 0 || j.unit_off + (j.cout + cpu - 1) / cpu > j.packed_units ||
            j.wp_bytes < (size_t)j.packed_units * 9 * cout_pad_for(j.n_out) * 16) {
            set_error("pack_weights_transposed: job %d: bad arguments (cout=%d cin=%d ci_lo=%d n_out=%d units=%d off=%d)", i,
                      j.cout, j.cin, j.ci_lo, j.n_out, j.packed_units, j.unit_off);
            return SR355_ERR_INVALID;
        }
        v[i] = PackTJob{j.w_dev, j.wp_dev, j.cout, j.cin, j.ci_lo, j.n_out, j.unit_off, cout_pad_for(j.n_out), j.scale};
    }
    return launch_pack_weight_t_batch(v.data(), n, dtype, (hipStream_t)stream);
}

int sr355_pool2x2_sum(const void *in, size_t in_bytes, int N, int H, int W, int units, int dtype, const void *mask,
                      size_t mask_bytes, int mask_units, int mask_unit_off, float mask_slope, float *out_f32,
                      size_t out_f32_bytes, void *out_act, size_t out_act_bytes, int out_units, int out_unit_off,
                      void *stream) {
    const int cpu = unit_channels(dtype);
    if (cpu < 0 || !in || !aligned16(in) || N < 1 || H < 1 || W < 1 || units < 1 ||
        in_bytes < (size_t)N * 4 * H * W * units * 16 || (!out_f32 && !out_act)) {
        set_error("pool2x2_sum: bad arguments (N=%d H=%d W=%d units=%d)", N, H, W, units);
        return SR355_ERR_INVALID;
    }
    if (mask && (mask_unit_off < 0 || mask_unit_off + units > mask_units || !aligned16(mask) ||
                 mask_bytes < (size_t)N * H * W * mask_units * 16)) {
        set_error("pool2x2_sum: mask geometry invalid");
        return SR355_ERR_INVALID;
    }
    if (out_f32 && (!aligned16(out_f32) || out_f32_bytes < (size_t)N * H * W * units * cpu * 4)) {
        set_error("pool2x2_sum: fp32 output too small");
        return SR355_ERR_INVALID;
    }
    if (out_act && (out_unit_off < 0 || out_unit_off + units > out_units || !aligned16(out_act) ||
                    out_act_bytes < (size_t)N * H * W * out_units * 16)) {
        set_error("pool2x2_sum: activation output geometry invalid");
        return SR355_ERR_INVALID;
    }
    return launch_pool2x2(in, N, H, W, units, dtype, mask, mask_units, mask_unit_off, mask_slope, out_f32, out_act,
                          out_units, out_unit_off, (hipStream_t)stream);
}

size_t sr355_wgrad_workspace_bytes(int cout, int cin, int N, int H, int W) {
    if (cout < 1 || cin < 1 || N < 1 || H < 1 || W < 1 || cout > 4096) return 0;
    return wgrad_workspace_bytes(cout, cin, N, H, W);
}

// Argument checks shared by the two weight-gradient entry points; `ws_bytes` is the scratch the caller has.
static int wgrad_checks(const char *who, const sr355_wgrad_args *p, const void *ws, size_t ws_bytes) {
    if (!p) {
        set_error("%s: args is null", who);
        return SR355_ERR_INVALID;
    }
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->cout < 1 || p->cout > 4096 ||
        p->cin < 1 ||
        (p->upsample != 0 && p->upsample != 1) || (p->upsample && ((p->H | p->W) & 1))) {
        set_error("%s: bad shape N=%d H=%d W=%d cout=%d cin=%d upsample=%d", who, p->N, p->H, p->W, p->cout,
                  p->cin, p->upsample);
        return SR355_ERR_INVALID;
    }
    const int Hin = p->H >> p->upsample, Win = p->W >> p->upsample;
    const int s2 = p->dy_stride2 ? 1 : 0;
    if (s2 && p->upsample) {
        set_error("%s: dy_stride2 and upsample exclude each other", who);
        return SR355_ERR_INVALID;
    }
    const size_t in_img = (size_t)Hin * p->in_units * Win * 16;
    const size_t dy_img = (size_t)((p->H + s2) >> s2) * p->dy_units * ((p->W + s2) >> s2) * 16;
    if (in_img >= ((size_t)1 << 31) || dy_img >= ((size_t)1 << 31)) {
        set_error("%s: one image exceeds the 31-bit offsets of the kernel", who);
        return SR355_ERR_UNSUPPORTED;
    }
    if (!p->in_dev || !aligned16(p->in_dev) || p->in_bytes < in_img * p->N || (p->cin + 7) / 8 > p->in_units ||
        !p->dy_dev || !aligned16(p->dy_dev) || p->dy_bytes < dy_img * p->N || p->dy_unit_off < 0 ||
        p->dy_unit_off + (p->cout + 7) / 8 > p->dy_units) {
        set_error("%s: input / gradient buffers missing, misaligned, too small or out of range", who);
        return SR355_ERR_INVALID;
    }
    const size_t need = wgrad_workspace_bytes(p->cout, p->cin, p->N, p->H, p->W);
    if (!p->dw_dev || p->dw_bytes < (size_t)p->cout * p->cin * 9 * 4 || !ws || !aligned16(ws) || ws_bytes < need) {
        set_error("%s: dw or scratch missing / too small (this shape needs %zu scratch bytes, %zu available)", who,
                  need, ws_bytes);
        return SR355_ERR_INVALID;
    }
    return SR355_OK;
}

int sr355_conv3x3_wgrad(const sr355_wgrad_args *p, void *stream) {
    const int rc = wgrad_checks("conv3x3_wgrad", p, p ? p->workspace_dev : nullptr, p ? p->workspace_bytes : 0);
    if (rc) return rc;
    return launch_wgrad3x3(p->in_dev, p->in_units, p->upsample, p->dy_dev, p->dy_units, p->dy_unit_off, p->N, p->H,
                           p->W, p->cout, p->cin, p->scale, p->accumulate, p->dw_dev, p->db_dev,
                           (float *)p->workspace_dev, (hipStream_t)stream, p->dy_stride2);
}

int sr355_conv3x3_wgrad_static(const sr355_wgrad_args *p, void *stream) {
    if (!p) {
        set_error("conv3x3_wgrad_static: args is null");
        return SR355_ERR_INVALID;
    }
    float *scratch = nullptr;
    int rc = wgrad_static_scratch(&scratch);   // this device's copy of the library's scratch array
    if (rc) return rc;
    // the caller's workspace fields are ignored: the partial slabs go to the static array
    if ((rc = wgrad_checks("conv3x3_wgrad_static", p, scratch, SR355_STATIC_SCRATCH_BYTES))) return rc;
    return launch_wgrad3x3(p->in_dev, p->in_units, p->upsample, p->dy_dev, p->dy_units, p->dy_unit_off, p->N, p->H,
                           p->W, p->cout, p->cin, p->scale, p->accumulate, p->dw_dev, p->db_dev, scratch,
                           (hipStream_t)stream, p->dy_stride2);
}

size_t sr355_dense_block_wgrad_workspace_bytes(int num_features, int growth_channels, int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1 || cout_pad_for(growth_channels) != 32 || num_features < 8 || cout_pad_for(num_features) > 64 ||
        ((num_features | growth_channels) & 7))
        return 0;
    return wgrad_block_workspace_bytes(num_features, growth_channels, N, H, W);
}

int sr355_dense_blocks_wgrad(const sr355_dense_wgrad_args *args, int n_blocks, void *stream) {
    if (!args || n_blocks < 1 || n_blocks > SR355_DENSE_WGRAD_MAX_BLOCKS) {
        set_error("dense_blocks_wgrad: args is null or n_blocks=%d outside [1, %d]", n_blocks, SR355_DENSE_WGRAD_MAX_BLOCKS);
        return SR355_ERR_INVALID;
    }
    const sr355_dense_wgrad_args *p = args;
    const int nf = p->num_features, gc = p->growth_channels;
    const size_t need = sr355_dense_block_wgrad_workspace_bytes(nf, gc, p->N, p->H, p->W);
    if (!need) {
        set_error("dense_block_wgrad: N=%d H=%d W=%d nf=%d gc=%d not supported (growth <= 32, features <= 64, multiples of 8)",
                  p->N, p->H, p->W, nf, gc);
        return SR355_ERR_UNSUPPORTED;
    }
    const size_t buf = (size_t)p->N * p->H * p->W * p->units * 16;
    if (p->units * 8 < nf + 4 * gc || !p->workspace_dev || !aligned16(p->workspace_dev) ||
        p->workspace_bytes < need * (size_t)n_blocks) {
        set_error("dense_block_wgrad: %d units for %d channels, or scratch missing / smaller than %d x %zu bytes", p->units,
                  nf + 4 * gc, n_blocks, need);
        return SR355_ERR_INVALID;
    }
    const void *act[SR355_DENSE_WGRAD_MAX_BLOCKS], *stack[SR355_DENSE_WGRAD_MAX_BLOCKS];
    float s5[SR355_DENSE_WGRAD_MAX_BLOCKS];
    float *dw[SR355_DENSE_WGRAD_MAX_BLOCKS][5], *db[SR355_DENSE_WGRAD_MAX_BLOCKS][5];
    for (int b = 0; b < n_blocks; ++b) {
        const sr355_dense_wgrad_args *q = args + b;
        if (q->N != p->N || q->H != p->H || q->W != p->W || q->units != p->units || q->num_features != nf ||
            q->growth_channels != gc || q->growth_scale != p->growth_scale) {
            set_error("dense_blocks_wgrad: block %d differs from block 0 in shape, units or growth scale", b);
            return SR355_ERR_INVALID;
        }
        if (!q->act_dev || !q->stack_dev || !aligned16(q->act_dev) || !aligned16(q->stack_dev) || q->act_bytes < buf ||
            q->stack_bytes < buf) {
            set_error("dense_block_wgrad: buffers of block %d missing / smaller than %zu bytes", b, buf);
            return SR355_ERR_INVALID;
        }
        for (int k = 0; k < 5; ++k) {
            if (!q->dw_dev[k] || !q->db_dev[k]) {
                set_error("dense_block_wgrad: dw / db of conv%d missing (block %d)", k + 1, b);
                return SR355_ERR_INVALID;
            }
            dw[b][k] = q->dw_dev[k], db[b][k] = q->db_dev[k];
        }
        act[b] = q->act_dev, stack[b] = q->stack_dev, s5[b] = q->conv5_scale;
    }
    return launch_wgrad3x3_blocks(n_blocks, act, stack, p->units, nf, gc, p->N, p->H, p->W, s5, dw, db,
                                  (float *)p->workspace_dev, (hipStream_t)stream, nullptr, p->growth_scale);
}

int sr355_dense_block_wgrad(const sr355_dense_wgrad_args *p, void *stream) {
    if (!p) {
        set_error("dense_block_wgrad: args is null");
        return SR355_ERR_INVALID;
    }
    return sr355_dense_blocks_wgrad(p, 1, stream);
}

size_t sr355_wgrad_f32_workspace_bytes(int cout, int cin, int N, int H, int W) {
    if (cout < 1 || cin < 1 || N < 1 || H < 1 || W < 1 || cout > 4096) return 0;
    return wgrad_f32_workspace_bytes(cout, cin, N, H, W);
}

int sr355_conv3x3_wgrad_f32(const sr355_wgrad_args *p, void *stream) {
    const char *who = "conv3x3_wgrad_f32";
    if (!p) {
        set_error("%s: args is null", who);
        return SR355_ERR_INVALID;
    }
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->cout < 1 || p->cout > 4096 || p->cin < 1 || p->dy_stride2 ||
        (p->upsample != 0 && p->upsample != 1) || (p->upsample && ((p->H | p->W) & 1))) {
        set_error("%s: bad shape N=%d H=%d W=%d cout=%d cin=%d upsample=%d dy_stride2=%d", who, p->N, p->H, p->W, p->cout,
                  p->cin, p->upsample, p->dy_stride2);
        return SR355_ERR_INVALID;
    }
    const int Hin = p->H >> p->upsample, Win = p->W >> p->upsample;
    const size_t in_img = (size_t)Hin * p->in_units * Win * 16, dy_img = (size_t)p->H * p->dy_units * p->W * 16;
    if (!p->in_dev || !aligned16(p->in_dev) || p->in_bytes < in_img * p->N || (p->cin + 3) / 4 > p->in_units ||
        !p->dy_dev || !aligned16(p->dy_dev) || p->dy_bytes < dy_img * p->N || p->dy_unit_off < 0 ||
        p->dy_unit_off + (p->cout + 3) / 4 > p->dy_units) {
        set_error("%s: input / gradient buffers missing, misaligned, too small or out of range", who);
        return SR355_ERR_INVALID;
    }
    const size_t need = wgrad_f32_workspace_bytes(p->cout, p->cin, p->N, p->H, p->W);
    if (!p->dw_dev || p->dw_bytes < (size_t)p->cout * p->cin * 9 * 4 || !p->workspace_dev || !aligned16(p->workspace_dev) ||
        p->workspace_bytes < need) {
        set_error("%s: dw or scratch missing / too small (this shape needs %zu scratch bytes, %zu available)", who, need,
                  p->workspace_bytes);
        return SR355_ERR_INVALID;
    }
    return launch_wgrad3x3_f32((const float *)p->in_dev, p->in_units, p->upsample, (const float *)p->dy_dev, p->dy_units,
                               p->dy_unit_off, p->N, p->H, p->W, p->cout, p->cin, p->scale, p->accumulate, p->dw_dev,
                               p->db_dev, (float *)p->workspace_dev, (hipStream_t)stream);
}

// ---- layers around the convs of the discriminator / VGG19, loss reductions ---------------------
size_t sr355_bn_workspace_bytes(int N, int H, int W, int C) {
    if (N < 1 || H < 1 || W < 1 || C < 8 || (C & 7)) return 0;
    return bn_workspace_floats(N, H, W, C) * sizeof(float);
}

static int bn_checks(const char *who, const float *x, size_t x_bytes, int N, int H, int W, int C, const float *gamma,
                     const float *beta, const float *mean, const float *rstd, const void *act, size_t act_bytes,
                     int units, int unit_off, const void *ws, size_t ws_bytes, int groups = 1) {
    if (N < 1 || H < 1 || W < 1 || C < 8 || (C & 7)) {
        set_error("%s: bad shape N=%d H=%d W=%d C=%d (channels in whole 8-channel units)", who, N, H, W, C);
        return SR355_ERR_INVALID;
    }
    if (groups < 1 || N % groups) {
        set_error("%s: %d groups do not divide the batch of %d", who, groups, N);
        return SR355_ERR_INVALID;
    }
    const size_t px = (size_t)N * H * W;
    if (!x || !aligned16(x) || x_bytes < px * C * 4 || !gamma || !beta || !mean || !rstd) {
        set_error("%s: fp32 input, gamma / beta or statistics missing, misaligned or too small", who);
        return SR355_ERR_INVALID;
    }
    if (!act || !aligned16(act) || unit_off < 0 || unit_off + C / 8 > units || act_bytes < px * units * 16) {
        set_error("%s: blocked tensor geometry invalid (off=%d units=%d, %zu bytes)", who, unit_off, units, act_bytes);
        return SR355_ERR_INVALID;
    }
    const size_t ws_need = (size_t)groups * sr355_bn_workspace_bytes(N / groups, H, W, C);
    if (!ws || !aligned16(ws) || ws_bytes < ws_need) {
        set_error("%s: workspace missing or smaller than %zu bytes", who, ws_need);
        return SR355_ERR_INVALID;
    }
    return SR355_OK;
}

int sr355_bn_lrelu_fwd(const float *x_f32, size_t x_bytes, int N, int H, int W, int C, const float *gamma,
                       const float *beta, float eps, float lrelu_slope, int training, float momentum,
                       float *running_mean, float *running_var, long long *num_batches_tracked, float *save_mean,
                       float *save_rstd, void *out_act, size_t out_act_bytes, int out_units, int out_unit_off,
                       void *workspace, size_t workspace_bytes, void *stream) {
    return sr355_bn_lrelu_fwd_groups(x_f32, x_bytes, N, H, W, C, gamma, beta, eps, lrelu_slope, training, momentum,
                                     running_mean, running_var, num_batches_tracked, save_mean, save_rstd, out_act,
                                     out_act_bytes, out_units, out_unit_off, workspace, workspace_bytes, 1, stream);
}

int sr355_bn_lrelu_fwd_groups(const float *x_f32, size_t x_bytes, int N, int H, int W, int C, const float *gamma,
                              const float *beta, float eps, float lrelu_slope, int training, float momentum,
                              float *running_mean, float *running_var, long long *num_batches_tracked, float *save_mean,
                              float *save_rstd, void *out_act, size_t out_act_bytes, int out_units, int out_unit_off,
                              void *workspace, size_t workspace_bytes, int groups, void *stream) {
    const int rc = bn_checks("bn_lrelu_fwd", x_f32, x_bytes, N, H, W, C, gamma, beta, save_mean, save_rstd, out_act,
                             out_act_bytes, out_units, out_unit_off, workspace, workspace_bytes, groups);
    if (rc) return rc;
    if (!training && (!running_mean || !running_var)) {
        set_error("bn_lrelu_fwd: eval mode needs the running statistics");
        return SR355_ERR_INVALID;
    }
    return launch_bn_lrelu_fwd(x_f32, N, H, W, C, gamma, beta, eps, lrelu_slope, training, momentum, running_mean,
                               running_var, num_batches_tracked, save_mean, save_rstd, out_act, out_units, out_unit_off,
                               (float *)workspace, (hipStream_t)stream, groups);
}

int sr355_bn_lrelu_bwd(const float *x_f32, size_t x_bytes, const void *dy_act, size_t dy_bytes, int dy_units,
                       int dy_unit_off, int N, int H, int W, int C, const float *gamma, const float *beta,
                       const float *save_mean, const float *save_rstd, float lrelu_slope, int training, void *dx_act,
                       size_t dx_bytes, int dx_units, int dx_unit_off, float *dgamma, float *dbeta, void *workspace,
                       size_t workspace_bytes, void *stream) {
    return sr355_bn_lrelu_bwd_groups(x_f32, x_bytes, dy_act, dy_bytes, dy_units, dy_unit_off, N, H, W, C, gamma, beta,
                                     save_mean, save_rstd, lrelu_slope, training, dx_act, dx_bytes, dx_units, dx_unit_off,
                                     dgamma, dbeta, workspace, workspace_bytes, 1, stream);
}

int sr355_bn_lrelu_bwd_groups(const float *x_f32, size_t x_bytes, const void *dy_act, size_t dy_bytes, int dy_units,
                              int dy_unit_off, int N, int H, int W, int C, const float *gamma, const float *beta,
                              const float *save_mean, const float *save_rstd, float lrelu_slope, int training,
                              void *dx_act, size_t dx_bytes, int dx_units, int dx_unit_off, float *dgamma, float *dbeta,
                              void *workspace, size_t workspace_bytes, int groups, void *stream) {
    // dy_units == 0: the gradient is an fp32 tensor in x's layout; the geometry check below then runs on dx's
    const bool dy_f32 = dy_units == 0;
    int rc = bn_checks("bn_lrelu_bwd", x_f32, x_bytes, N, H, W, C, gamma, beta, save_mean, save_rstd,
                       dy_f32 ? dx_act : dy_act, dy_f32 ? dx_bytes : dy_bytes, dy_f32 ? dx_units : dy_units,
                       dy_f32 ? dx_unit_off : dy_unit_off, workspace, workspace_bytes, groups);
    if (rc) return rc;
    const size_t px = (size_t)N * H * W;
    if (dy_f32 && (!dy_act || !aligned16(dy_act) || dy_unit_off != 0 || dy_bytes < px * C * 4)) {
        set_error("bn_lrelu_bwd: fp32 gradient (dy_units = 0) missing, misaligned or smaller than %zu bytes", px * C * 4);
        return SR355_ERR_INVALID;
    }
    if (!dx_act || !aligned16(dx_act) || dx_unit_off < 0 || dx_unit_off + C / 8 > dx_units ||
        dx_bytes < px * dx_units * 16 || !dgamma || !dbeta) {
        set_error("bn_lrelu_bwd: dx geometry invalid or dgamma / dbeta missing");
        return SR355_ERR_INVALID;
    }
    return launch_bn_lrelu_bwd(x_f32, dy_act, dy_units, dy_unit_off, N, H, W, C, gamma, beta, save_mean, save_rstd,
                               lrelu_slope, training, dx_act, dx_units, dx_unit_off, dgamma, dbeta, (float *)workspace,
                               (hipStream_t)stream, groups);
}

int sr355_maxpool2x2_fwd(const void *in_act, size_t in_bytes, int N, int H, int W, int units, void *out_act,
                         size_t out_bytes, void *stream) {
    if (N < 1 || H < 1 || W < 1 || units < 1 || !in_act || !aligned16(in_act) || !out_act || !aligned16(out_act) ||
        in_bytes < (size_t)N * H * W * units * 16 || out_bytes < (size_t)N * (H / 2) * (W / 2) * units * 16) {
        set_error("maxpool2x2_fwd: bad arguments (N=%d H=%d W=%d units=%d)", N, H, W, units);
        return SR355_ERR_INVALID;
    }
    return launch_maxpool2x2_fwd(in_act, N, H, W, units, out_act, (hipStream_t)stream);
}

int sr355_pixel_shuffle2(const void *src, size_t src_bytes, int N, int H, int W, int c_out, int dtype, int inverse, void *dst,
                         size_t dst_bytes, void *stream) {
    const int cpu = unit_channels(dtype);
    if (cpu < 0 || N < 1 || H < 1 || W < 1 || c_out < cpu || c_out % cpu || (long long)N * H * W * c_out > 0x3fffffffLL) {
        set_error("pixel_shuffle2: bad arguments (N=%d H=%d W=%d c_out=%d dtype=%d)", N, H, W, c_out, dtype);
        return SR355_ERR_INVALID;
    }
    const size_t bytes = (size_t)N * H * W * 4 * c_out * (16 / cpu);     // both tensors hold the same elements
    if (!src || !dst || !aligned16(src) || !aligned16(dst) || src_bytes < bytes || dst_bytes < bytes) {
        set_error("pixel_shuffle2: buffers missing, misaligned or smaller than %zu bytes", bytes);
        return SR355_ERR_INVALID;
    }
    return launch_pixel_shuffle2(src, N, H, W, c_out, dtype, inverse ? 1 : 0, dst, (hipStream_t)stream);
}

int sr355_maxpool2x2_bwd(const void *in_act, size_t in_bytes, const void *dy_act, size_t dy_bytes, int N, int H, int W,
                         int units, void *dx_act, size_t dx_bytes, void *stream) {
    if (N < 1 || H < 1 || W < 1 || units < 1 || !in_act || !aligned16(in_act) || !dy_act || !aligned16(dy_act) ||
        !dx_act || !aligned16(dx_act) || in_bytes < (size_t)N * H * W * units * 16 || dx_bytes < (size_t)N * H * W * units * 16 ||
        dy_bytes < (size_t)N * (H / 2) * (W / 2) * units * 16) {
        set_error("maxpool2x2_bwd: bad arguments (N=%d H=%d W=%d units=%d)", N, H, W, units);
        return SR355_ERR_INVALID;
    }
    return launch_maxpool2x2_bwd(in_act, dy_act, N, H, W, units, dx_act, (hipStream_t)stream);
}

size_t sr355_loss_workspace_bytes(size_t n) { return n ? (size_t)loss_parts(n) * sizeof(float) : 0; }

int sr355_loss_fwd(const float *a, const float *b, float b_const, size_t n, int mode, float *loss_out, float *grad_out,
                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!a || !aligned16(a) || !aligned16(b) || !aligned16(grad_out) || n < 1 || (mode != 0 && mode != 1) || !loss_out ||
        !workspace || workspace_bytes < sr355_loss_workspace_bytes(n)) {
        set_error("loss_fwd: bad arguments (n=%zu mode=%d; tensors 16-byte aligned, workspace >= %zu bytes)", n, mode,
                  sr355_loss_workspace_bytes(n));
        return SR355_ERR_INVALID;
    }
    return launch_loss(a, b, b_const, n, mode, loss_out, grad_out, (float *)workspace, (hipStream_t)stream);
}

// ---- generator ----------------------------------------------------------------------------
int sr355_generator_num_convs(const sr355_generator_cfg *cfg) {
    if (!cfg_ok(cfg)) return SR355_ERR_INVALID;
    return 1 + 15 * cfg->num_blocks + 5;
}

// The packed blob ends with room for the table of pack jobs (one per conv), which sr355_generator_pack copies there so
// that one launch packs every conv; the forward passes read [0, total) only.
static size_t pack_table_bytes(size_t n_convs) { return align256(n_convs * sizeof(PackJob)); }

size_t sr355_generator_packed_bytes(const sr355_generator_cfg *cfg) {
    if (!cfg_ok(cfg)) return 0;
    size_t total = 0;
    const size_t n = conv_specs(*cfg, &total).size();
    return total + pack_table_bytes(n);
}

int sr355_generator_pack(const sr355_generator_cfg *cfg, const float *const *w_ptrs, const float *const *b_ptrs,
                         int n_convs, void *packed, size_t packed_bytes, void *stream) {
    if (!cfg_ok(cfg)) return SR355_ERR_INVALID;
    size_t total = 0;
    std::vector<ConvSpec> specs = conv_specs(*cfg, &total);
    if (!w_ptrs || !b_ptrs || n_convs != (int)specs.size() || !packed || !aligned16(packed) ||
        packed_bytes < total + pack_table_bytes(specs.size())) {
        set_error("generator_pack: need %zu conv tensors and %zu bytes (got %d, %zu)", specs.size(),
                  total + pack_table_bytes(specs.size()), n_convs, packed_bytes);
        return SR355_ERR_INVALID;
    }
    unsigned char *base = (unsigned char *)packed;
    std::vector<PackJob> jobs(specs.size());
    for (size_t i = 0; i < specs.size(); ++i) {
        if (!w_ptrs[i] || !b_ptrs[i]) {
            set_error("generator_pack: tensor %zu is null", i);
            return SR355_ERR_INVALID;
        }
        jobs[i] = PackJob{w_ptrs[i], b_ptrs[i], base + specs[i].w_off, (float *)(base + specs[i].b_off), specs[i].cout,
                          specs[i].cin, specs[i].split, specs[i].f16};
    }
    return launch_pack_weight_table(jobs.data(), (int)jobs.size(), cfg->dtype, base + total, (hipStream_t)stream);
}

size_t sr355_generator_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W) {
    if (!cfg_ok(cfg) || N < 1 || H < 1 || W < 1) return 0;
    return carve(*cfg, N, H, W).total;
}

// Buffers of one forward pass.  Inference ping-pongs two dense-block buffers; training keeps one per
// dense block (plus the last one's x slice) because the backward pass needs every activation.
struct FwdBufs {
    unsigned char *x0;
    unsigned char *blk_base;
    size_t blk_stride;
    int blk_count;       // 2 = ping-pong, otherwise 3*num_blocks + 1 distinct buffers
    float *T, *R, *F;
    unsigned char *L1, *H1, *H2, *H3;
    unsigned *chain_flags = nullptr;      // chain kernel state (inference workspace only)
    size_t chain_flag_bytes = 0;
    ChainPass *chain_table = nullptr;
    unsigned char *blk(int i) const { return blk_base + (size_t)(blk_count == 2 ? (i & 1) : i) * blk_stride; }
};

// Second stream of a device for the two-stream forward (created on first use, never destroyed) and the
// fork/join events around it.
constexpr int MAX_PARTS = 4;
struct SideStream {
    hipStream_t stream[MAX_PARTS - 1] = {};
    hipEvent_t fork = nullptr, join[MAX_PARTS - 1] = {};
    hipEvent_t main_mark = nullptr, side_mark[2] = {};   // backward: hand-offs between the two streams
    std::mutex in_use;   // held while one forward / backward issues its launches: the events are not reentrant
};

static SideStream *side_stream() {
    static SideStream per_dev[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    SideStream &ss = per_dev[dev];
    if (!ss.fork) {
        for (int i = 0; i < MAX_PARTS - 1; ++i)
            if (hipStreamCreateWithFlags(&ss.stream[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ss.join[i], hipEventDisableTiming) != hipSuccess)
                return nullptr;
        if (hipEventCreateWithFlags(&ss.main_mark, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.side_mark[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.side_mark[1], hipEventDisableTiming) != hipSuccess)
            return nullptr;
        if (hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &ss;
}

// Images are independent and every buffer is image-major, so a batch can run as several part-batches
// on as many streams: one part's kernels fill the launch gaps, prologues and store bursts of the
// others'.  Worth it only while each part still fills the chip; returns the number of parts.
static int batch_parts(int N, int H, int W) {
    static const int knob = [] { const char *e = getenv("SR355_STREAMS"); return e ? atoi(e) : 2; }();
    int parts = knob < 1 ? 1 : (knob > MAX_PARTS ? MAX_PARTS : knob);
    if (parts > N) parts = N;
    const long long tiles_img = (long long)((W + 31) / 32) * ((H + 7) / 8);
    while (parts > 1 && (N / parts) * tiles_img < 256) --parts;
    return parts;
}

// A dense block as one launch (dense_block.hip) instead of five: for images at most 32 pixels wide whose two-row
// bands do not overfill the chip many times over (the fused kernel recomputes halo rows: 2.15 x the MFMA work).
// SR355_DENSE_FUSED=0 keeps the five-launch path; SR355_DENSE_FUSED_MAX overrides the band limit.
static bool use_dense_fused(const sr355_generator_cfg *cfg, int N, int H, int W) {
    static const int knob = [] { const char *e = getenv("SR355_DENSE_FUSED"); return e ? atoi(e) : 1; }();
    static const long long max_bands = [] { const char *e = getenv("SR355_DENSE_FUSED_MAX"); return e ? atoll(e) : 1024LL; }();
    if (!knob || !dense_block_supported(cfg->num_features, cfg->growth_channels, cfg->dtype, W)) return false;
    return (long long)N * ((H + 1) / 2) <= max_bands;
}

// The word the chain kernel raises when a wait gave up: pinned host memory mapped into the device's address space, one
// word per device.  The kernel only ever stores 1 to it (a system-scope store, straight to the host word), the host
// clears it when it reports the failure: a launch that gave up stays visible however many launches follow it before
// anyone asks.
static unsigned *chain_err_host(int dev, unsigned **device_view = nullptr) {
    static std::once_flag once[64];
    static unsigned *word[64] = {};
    static unsigned *view[64] = {};
    if (dev < 0 || dev >= 64) return nullptr;
    std::call_once(once[dev], [dev] {
        void *p = nullptr, *d = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return;
        *(volatile unsigned *)p = 0;
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) {
            (void)hipHostFree(p);
            return;
        }
        word[dev] = (unsigned *)p;
        view[dev] = (unsigned *)d;
    });
    if (device_view) *device_view = view[dev];
    return word[dev];
}

// Two chain kernels in flight on one GPU could each hold part of the workgroup slots and wait for peers that never
// become resident.  Within a process they are therefore serialised: every chain launch waits for the event recorded
// behind the previous one (a no-op on the same stream).  Another PROCESS running the same kernel on the same GPU is
// not covered: its waits would give up after two seconds and the next call reports it.
struct ChainOrder {
    std::mutex m;
    hipEvent_t done = nullptr;
    bool recorded = false;
    hipEvent_t err_copied = nullptr;     // recorded behind the chain launch: once it is done the error word is final for that launch
    bool err_recorded = false;
    std::atomic<bool> disabled{false};   // a launch gave up once: this device runs one launch per conv from then on
    // pinned staging area of the pass table: the copy to the device is a true asynchronous copy, and the bytes it
    // reads stay valid until `table_copied` says it is done (a copy from a local vector would have to be staged by the
    // runtime before the call returns)
    void *table_host = nullptr;
    size_t table_cap = 0;
    hipEvent_t table_copied = nullptr;
    bool table_recorded = false;
};
static ChainOrder *chain_order(int dev) {
    static std::once_flag once[64];
    static ChainOrder ord[64];
    if (dev < 0 || dev >= 64) return nullptr;
    std::call_once(once[dev], [dev] {
        if (hipEventCreateWithFlags(&ord[dev].done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ord[dev].err_copied, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ord[dev].table_copied, hipEventDisableTiming) != hipSuccess)
            ord[dev].done = nullptr;
    });
    return ord[dev].done ? &ord[dev] : nullptr;
}

// The body (every dense block and trunk_conv) as one persistent launch (conv3x3_chain.hip) where 16-row tiles give every
// CU work: large batches / images.  SR355_CHAIN=0 keeps one launch per conv.
static bool use_chain(const sr355_generator_cfg *cfg, int N, int H, int W) {
    const char *e = getenv("SR355_CHAIN");      // read on every call: tests compare the two paths in one process
    const int knob = e ? atoi(e) : 1;
    if (!knob || cfg->dtype != SR355_DTYPE_BF16 || cfg->num_features != 64 || cfg->growth_channels != 32) return false;
    int slots = 0, dev = 0;
    if (chain_resident_workgroups(&slots) || slots < 2) return false;
    if (hipGetDevice(&dev) == hipSuccess) {
        ChainOrder *order = chain_order(dev);
        if (order && order->disabled.load(std::memory_order_relaxed)) return false;
    }
    return (long long)N * ((H + CHAIN_TILE_ROWS - 1) / CHAIN_TILE_ROWS) * ((W + 31) / 32) >= slots / CHAIN_WG_PER_CU;
}

// Whether the most recent chain launch on `dev` gave up a wait (its output is then invalid).  wait: block until that
// launch and the copy of its error word have completed; otherwise answer from what has completed so far.  A launch that
// gave up is reported once (SR355_ERR_HIP, with the reason) and switches the device to one launch per conv: the
// workgroups of the persistent launch evidently cannot all be resident here (another process on the GPU, masked CUs).
static int chain_status_of(int dev, bool wait) {
    ChainOrder *order = chain_order(dev);
    unsigned *err_host = chain_err_host(dev);
    if (!order || !err_host) return SR355_OK;
    std::lock_guard<std::mutex> lock(order->m);
    bool pending = false;
    if (order->err_recorded) {
        if (wait) {
            if (check_hip(hipEventSynchronize(order->err_copied), "hipEventSynchronize")) return SR355_ERR_HIP;
        } else {
            pending = hipEventQuery(order->err_copied) != hipSuccess;
        }
    }
    if (!*(volatile unsigned *)err_host) return pending ? SR355_CHAIN_PENDING : SR355_OK;
    *(volatile unsigned *)err_host = 0;
    order->disabled.store(true, std::memory_order_relaxed);
    set_error("generator forward: the body's persistent launch gave up waiting for a workgroup that never became "
              "resident (is another process using this GPU?); that forward's output is invalid; this device now runs "
              "one launch per conv");
    return SR355_ERR_HIP;
}

// SR355_CHAIN_GRID=<n> (read on every call, tests only): launch the persistent kernel with exactly n workgroups even
// if they cannot all be resident, which is how the give-up path of its waits is exercised.
static int chain_grid_knob() {
    const char *e = getenv("SR355_CHAIN_GRID");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? -n : 0;
}

// Optional events around the chain launch (sr355_chain_timing / sr355_chain_last_ms).
static std::atomic<int> g_chain_timing{0};
struct ChainEvents {
    hipEvent_t before = nullptr, after = nullptr;
    bool recorded = false;
};
static ChainEvents *chain_events(int dev) {
    static std::once_flag once[64];
    static ChainEvents ev[64];
    if (dev < 0 || dev >= 64) return nullptr;
    std::call_once(once[dev], [dev] {
        if (hipEventCreate(&ev[dev].before) != hipSuccess || hipEventCreate(&ev[dev].after) != hipSuccess)
            ev[dev].before = ev[dev].after = nullptr;
    });
    return ev[dev].before ? &ev[dev] : nullptr;
}

// Input: fp32 NCHW, or (in_u8) a uint8 HWC image batch read as byte / 255.  Output: fp32 NCHW unclamped, or (out_u8)
// the uint8 HWC image batch the reference's inference saves (clamp, mul 255, truncate).
static int run_forward(const sr355_generator_cfg *cfg, const std::vector<ConvSpec> &specs, const unsigned char *pk,
                       const float *in_nchw, float *out_nchw, int N, int H, int W, const Workspace &ws,
                       const FwdBufs &fb, hipStream_t stream, const unsigned char *in_u8 = nullptr,
                       unsigned char *out_u8 = nullptr) {
    const int dtype = cfg->dtype, cpu = unit_channels(dtype);
    const int nf = cfg->num_features, gc = cfg->growth_channels;
    float *T = fb.T, *R = fb.R, *F = fb.F;
    const int in_split = specs[0].split == SR355_SPLIT_INPUT;
    const bool rrdb_only = cfg->trunk_fp32 == SR355_TRUNK_RRDB && dtype == SR355_DTYPE_BF16;
    int rc = in_u8 ? launch_u8hwc_to_blocked(in_u8, N, cfg->in_channels, H, W, dtype, fb.x0, ws.in_units, in_split,
                                             stream)
                   : launch_nchw_to_blocked(in_nchw, N, cfg->in_channels, H, W, dtype, fb.x0, ws.in_units, in_split,
                                            stream);
    if (rc) return rc;

    const bool fused = use_dense_fused(cfg, N, H, W);
    const bool chain = !fused && fb.chain_table && use_chain(cfg, N, H, W);
    struct ChainConv {
        ConvArgs a;
        int cin_units;
        bool all_stages;      // every stage reads the previous conv's output (a block's first conv, trunk_conv)
    };
    std::vector<ChainConv> chain_convs;
    bool chain_collect = false, chain_all = false;
    // parts of the batch and the stream each runs on (the chain kernel keeps the whole batch in one launch)
    int n_parts = dtype == SR355_DTYPE_BF16 && !chain ? batch_parts(N, H, W) : 1;
    int part_n0[MAX_PARTS] = {0}, part_n[MAX_PARTS] = {N};
    hipStream_t part_stream[MAX_PARTS] = {stream};
    SideStream *ss = n_parts > 1 ? side_stream() : nullptr;
    if (!ss) n_parts = 1;
    std::unique_lock<std::mutex> side_lock;
    if (ss) {
        side_lock = std::unique_lock<std::mutex>(ss->in_use);
        if (check_hip(hipEventRecord(ss->fork, stream), "hipEventRecord")) return SR355_ERR_HIP;
        for (int p = 0; p < n_parts; ++p) {
            part_n0[p] = (int)((long long)N * p / n_parts);
            part_n[p] = (int)((long long)N * (p + 1) / n_parts) - part_n0[p];
            if (p > 0) {
                part_stream[p] = ss->stream[p - 1];
                if (check_hip(hipStreamWaitEvent(part_stream[p], ss->fork, 0), "hipStreamWaitEvent"))
                    return SR355_ERR_HIP;
            }
        }
    }

    size_t ci = 0;  // running conv index
    auto conv = [&](const unsigned char *in, int in_units, int cin_units, int h, int w, int ups,
                    unsigned char *out_act, int out_units, int out_off, float slope, const float *res1, float s1,
                    const float *res2, float s2, float *out_f32, float *out_nc,
                    unsigned char *out_img = nullptr, const unsigned char *res_act = nullptr) -> int {
        const ConvSpec &sp = specs[ci++];
        const size_t in_img = (size_t)(h >> ups) * (w >> ups) * in_units * 16;
        const size_t act_img = (size_t)h * w * out_units * 16;
        const size_t f32_img = (size_t)h * w * sp.cout;   // floats
        for (int p = 0; p < n_parts; ++p) {
            const size_t n0 = (size_t)part_n0[p];
            ConvArgs a = conv_args_zero();
            a.in = in + n0 * in_img;
            a.w = pk + sp.w_off;
            a.bias = (const float *)(pk + sp.b_off);
            a.out_act = out_act ? out_act + n0 * act_img : nullptr;
            a.out_f32 = out_f32 ? out_f32 + n0 * f32_img : nullptr;
            a.out_nchw = out_nc ? out_nc + n0 * f32_img : nullptr;
            a.out_u8 = out_img ? out_img + n0 * f32_img : nullptr;   // same element count, one byte each
            a.res_act = res_act ? res_act + n0 * in_img : nullptr;   // a buffer of the input's geometry
            a.res_act_units = in_units;
            a.res1 = res1 ? res1 + n0 * f32_img : nullptr;
            a.res2 = res2 ? res2 + n0 * f32_img : nullptr;
            a.res1_scale = s1;
            a.res2_scale = s2;
            a.slope = slope;
            a.N = part_n[p];
            a.H = h;
            a.W = w;
            a.in_units = in_units;
            a.ups = ups;
            a.out_units = out_units;
            a.out_unit_off = out_off;
            a.cout_real = sp.cout;
            a.cout_pad = cout_pad_for(sp.cout);
            a.f32_units = sp.cout / 4;
            a.f16_in = sp.f16;
            // a blocked output that the next conv reads as fp16: every conv in front of an fp16 one
            a.f16_out = (out_act && ci < specs.size() && specs[ci].f16) ? 1 : 0;
            if (chain_collect) {
                if (!chain_supported(a, cin_units) || sp.split != SR355_SPLIT_NONE) {
                    set_error("generator forward: a conv of the body does not fit the chain kernel");
                    return SR355_ERR_UNSUPPORTED;
                }
                chain_convs.push_back({a, cin_units, chain_all});
                continue;
            }
            const int rc_ = launch_conv3x3(a, dtype, cin_units, sp.split != SR355_SPLIT_NONE, part_stream[p]);
            if (rc_) return rc_;
        }
        return SR355_OK;
    };

    // one dense block (:18-24) as a single launch; conv5's epilogue arguments as in `conv`
    auto dense = [&](unsigned char *cur, unsigned char *nxt, const float *res1, float s1, const float *res2, float s2,
                     float *out_f32, bool res_from_act) -> int {
        const size_t blk_img = (size_t)H * W * ws.blk_units * 16;
        const size_t f32_img = (size_t)H * W * nf;   // floats
        const ConvSpec *sp = &specs[ci];
        ci += 5;
        for (int p = 0; p < n_parts; ++p) {
            const size_t n0 = (size_t)part_n0[p];
            DenseArgs d;
            memset(&d, 0, sizeof(d));
            d.buf = cur + n0 * blk_img;
            d.growth_out = fb.blk_count == 2 ? nullptr : cur + n0 * blk_img;   // kept only where a backward reads them
            for (int k = 0; k < 5; ++k) {
                d.w[k] = pk + sp[k].w_off;
                d.bias[k] = (const float *)(pk + sp[k].b_off);
            }
            d.slope = 0.2f;
            d.mask_slope = 1.f;
            d.N = part_n[p], d.H = H, d.W = W, d.units = ws.blk_units;
            ConvArgs &a = d.tail;
            a = conv_args_zero();
            a.out_act = nxt + n0 * blk_img;
            a.out_units = ws.blk_units;
            a.out_f32 = out_f32 ? out_f32 + n0 * f32_img : nullptr;
            a.res_act = res_from_act ? cur + n0 * blk_img : nullptr;
            a.res_act_units = ws.blk_units;
            a.res1 = res1 ? res1 + n0 * f32_img : nullptr;
            a.res2 = res2 ? res2 + n0 * f32_img : nullptr;
            a.res1_scale = s1;
            a.res2_scale = s2;
            a.cout_real = a.cout_pad = nf;
            a.f32_units = nf / 4;
            const int rc_ = launch_dense_block(d, part_stream[p]);
            if (rc_) return rc_;
        }
        return SR355_OK;
    };

    int bi = 0;  // dense-block buffer index
    // conv_first (:56): x -> block buffer units [0, nf) and the fp32 copy F kept for the global skip.
    if ((rc = conv(fb.x0, ws.in_units, ws.in_units, H, W, 0, fb.blk(0), ws.blk_units, 0, 1.f, nullptr, 0.f, nullptr,
                   0.f, F, nullptr)))
        return rc;
    chain_collect = chain;
    for (int b = 0; b < cfg->num_blocks; ++b) {
        const float *rrdb_in = (b == 0) ? F : R;  // fp32 input of this RRDB (:35)
        for (int r = 0; r < 3; ++r, ++bi) {
            unsigned char *cur = fb.blk(bi), *nxt = fb.blk(bi + 1);
            if (fused) {
                if (rrdb_only)
                    rc = dense(cur, nxt, nullptr, 0.2f, r < 2 ? nullptr : rrdb_in, r < 2 ? 0.f : 0.2f, r < 2 ? nullptr : R,
                               true);
                else
                    rc = dense(cur, nxt, (r == 0) ? rrdb_in : T, 0.2f, r < 2 ? nullptr : rrdb_in, r < 2 ? 0.f : 0.2f,
                               r < 2 ? T : R, false);
                if (rc) return rc;
                continue;
            }
            for (int k = 0; k < 4; ++k) {  // growth convs (:19-22): read [0, nf+k*gc), append gc channels
                chain_all = k == 0;
                if ((rc = conv(cur, ws.blk_units, (nf + k * gc) / cpu, H, W, 0, cur, ws.blk_units,
                               (nf + k * gc) / cpu, 0.2f, nullptr, 0.f, nullptr, 0.f, nullptr, nullptr)))
                    return rc;
            }
            // conv5 + dense skip (:23-24); the third block also folds the RRDB skip (:39).
            chain_all = false;
            if (rrdb_only) {
                // the dense skip reads the bf16 activations (units [0, nf) of this block's buffer); only the RRDB skip
                // keeps an fp32 carrier
                if (r < 2)
                    rc = conv(cur, ws.blk_units, ws.blk_units, H, W, 0, nxt, ws.blk_units, 0, 1.f, nullptr, 0.2f, nullptr,
                              0.f, nullptr, nullptr, nullptr, cur);
                else
                    rc = conv(cur, ws.blk_units, ws.blk_units, H, W, 0, nxt, ws.blk_units, 0, 1.f, nullptr, 0.2f, rrdb_in,
                              0.2f, R, nullptr, nullptr, cur);
                if (rc) return rc;
                continue;
            }
            const float *res1 = (r == 0) ? rrdb_in : T;
            if (r < 2)
                rc = conv(cur, ws.blk_units, ws.blk_units, H, W, 0, nxt, ws.blk_units, 0, 1.f, res1, 0.2f, nullptr,
                          0.f, T, nullptr);
            else
                rc = conv(cur, ws.blk_units, ws.blk_units, H, W, 0, nxt, ws.blk_units, 0, 1.f, res1, 0.2f, rrdb_in,
                          0.2f, R, nullptr);
            if (rc) return rc;
        }
    }
    // trunk_conv + global skip (:57-58)
    chain_all = true;
    if ((rc = conv(fb.blk(bi), ws.blk_units, ws.nf_units, H, W, 0, fb.L1, ws.nf_units, 0, 1.f, F, 1.f, nullptr, 0.f,
                   nullptr, nullptr)))
        return rc;
    if (chain) {
        // The collected convs as passes of 32 output channels.  A pass's `need` is its own index: everything before it.
        chain_collect = false;
        std::vector<ChainPass> passes;
        int prev_blocks = 0, prev_cout = 0;       // output-channel blocks (= passes) and channels of the conv before
        // Stage order (ChainPass::perm).  A conv that appends to what the pass before it read walks the old stages in the
        // reverse of that pass's order (the lines the L2 touched last come first) and the new channels last, where the
        // wait for the neighbours sits; a second block of output channels walks the first block's order backwards.
        const char *perm_env = getenv("SR355_CHAIN_PERM");      // 0: every pass in channel order (A/B measurements)
        const bool permute = !perm_env || atoi(perm_env) != 0;
        std::vector<int> prev_order;
        for (const ChainConv &cc : chain_convs) {
            const int S = cc.cin_units / 2;
            const int first_pass = (int)passes.size();
            for (int cb = 0; cb < cc.a.cout_pad / 32; ++cb) {
                ChainPass ps;
                memset(&ps, 0, sizeof(ps));
                ps.a = cc.a;
                ps.a.n_stages = ps.a.n_real_stages = S;
                {
                    std::vector<int> order;
                    if (permute && S <= 16 && !cc.all_stages && (int)prev_order.size() <= S) {
                        order.assign(prev_order.rbegin(), prev_order.rend());
                        std::vector<bool> seen(S, false);
                        bool ok = true;
                        for (int v : order) {
                            if (v >= S || seen[v]) ok = false;
                            else seen[v] = true;
                        }
                        if (!ok) order.clear(), seen.assign(S, false);
                        for (int v = 0; v < S; ++v)
                            if (!seen[v]) order.push_back(v);
                    } else {
                        for (int v = 0; v < S; ++v) order.push_back(v);
                    }
                    if (S > 16) {
                        set_error("generator forward: chain pass with %d stages (at most 16)", S);
                        return SR355_ERR_UNSUPPORTED;
                    }
                    ps.perm = 0;
                    for (int i = 0; i < 16; ++i) ps.perm |= (unsigned long long)(i < S ? order[i] : i) << (4 * i);
                    prev_order = order;
                }
                ps.a.w = cc.a.w + (size_t)cb * 32 * 16;
                ps.a.bias = cc.a.bias + cb * 32;
                ps.cb_co = cb * 32;
                ps.need = first_pass;
                // the second block of output channels reads what the first one read: nothing new to wait for
                ps.dep_stage = cb ? S : (cc.all_stages ? 0 : S - 2);
                if (cb == 0 && cc.all_stages && prev_blocks == 2 && cc.cin_units * 8 == prev_cout && S == 4) {
                    // The whole input was written by the two passes of the conv before: stages 0, 1 read the first
                    // pass's 32 channels, stages 2, 3 the second's.  The first stage can then be requested while this
                    // workgroup still computes that second pass, and the pipeline does not restart.
                    ps.need = first_pass - 1;
                    ps.dep2_stage = 2;
                    ps.need2 = first_pass;
                }
                if (first_pass == 0) ps.dep_stage = S, ps.need2 = 0;      // the chain's input comes from the launch before it
                if (ps.dep_stage == 1 || S < 2) {
                    set_error("generator forward: chain pass with %d stages cannot be ordered", S);
                    return SR355_ERR_UNSUPPORTED;
                }
                chain_fill_cursor(ps);
                passes.push_back(ps);
            }
            prev_blocks = cc.a.cout_pad / 32;
            prev_cout = cc.a.cout_pad;
        }
        if (passes.size() * sizeof(ChainPass) > (size_t)(18 * cfg->num_blocks + 2) * sizeof(ChainPass)) {
            set_error("generator forward: chain table overflow");
            return SR355_ERR_INVALID;
        }
        int dev = 0;
        if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
        unsigned *err_dev = nullptr;
        if (!chain_err_host(dev, &err_dev) || !err_dev) {
            set_error("generator forward: cannot map the host word the persistent launch reports a failed wait in");
            return SR355_ERR_HIP;
        }
        if (chain_status_of(dev, false) == SR355_ERR_HIP) return SR355_ERR_HIP;   // (the device now runs one launch per conv)
        ChainOrder *order = chain_order(dev);
        if (!order) {
            set_error("generator forward: cannot create the event that orders chain launches");
            return SR355_ERR_HIP;
        }
        {
            std::lock_guard<std::mutex> lock(order->m);
            const size_t bytes = passes.size() * sizeof(ChainPass);
            if (order->table_recorded && check_hip(hipEventSynchronize(order->table_copied), "hipEventSynchronize"))
                return SR355_ERR_HIP;      // (the previous forward's copy out of the staging area: long done)
            if (order->table_cap < bytes) {
                if (order->table_host) (void)hipHostFree(order->table_host);
                order->table_host = nullptr, order->table_cap = 0;
                if (check_hip(hipHostMalloc(&order->table_host, bytes, hipHostMallocDefault), "hipHostMalloc"))
                    return SR355_ERR_HIP;
                order->table_cap = bytes;
            }
            memcpy(order->table_host, passes.data(), bytes);
            if (check_hip(hipMemcpyAsync(fb.chain_table, order->table_host, bytes, hipMemcpyHostToDevice, stream),
                          "hipMemcpyAsync") ||
                check_hip(hipEventRecord(order->table_copied, stream), "hipEventRecord") ||
                check_hip(hipMemsetAsync(fb.chain_flags, 0, fb.chain_flag_bytes, stream), "hipMemsetAsync"))
                return SR355_ERR_HIP;
            order->table_recorded = true;
        }
        ChainArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.passes = fb.chain_table;
        ca.n_passes = (int)passes.size();
        ca.flags = fb.chain_flags;
        ca.err = err_dev;
        ca.N = N, ca.H = H, ca.W = W, ca.in_units = ws.blk_units;
#ifdef SR355_TRACE
        // timing builds: where wave 0 of every workgroup spent the launch (medians over the workgroups, microseconds)
        static unsigned long long *trace_dev = nullptr;
        if (getenv("SR355_CHAIN_TRACE")) {
            if (!trace_dev && hipMalloc((void **)&trace_dev, 4096 * 96) != hipSuccess) trace_dev = nullptr;
            ca.trace = trace_dev;
        }
#endif
        ChainEvents *cev = g_chain_timing.load() ? chain_events(dev) : nullptr;
        {
            std::lock_guard<std::mutex> lock(order->m);
            if (order->recorded && check_hip(hipStreamWaitEvent(stream, order->done, 0), "hipStreamWaitEvent"))
                return SR355_ERR_HIP;
            if (cev && check_hip(hipEventRecord(cev->before, stream), "hipEventRecord")) return SR355_ERR_HIP;
            if ((rc = launch_conv3x3_chain(ca, chain_grid_knob(), stream))) return rc;
            if (cev) {
                if (check_hip(hipEventRecord(cev->after, stream), "hipEventRecord")) return SR355_ERR_HIP;
                cev->recorded = true;
            }
            if (check_hip(hipEventRecord(order->done, stream), "hipEventRecord")) return SR355_ERR_HIP;
            order->recorded = true;
        }
#ifdef SR355_TRACE
        if (ca.trace) {
            static int calls = 0;
            if (++calls == 5) {
                std::vector<unsigned long long> t(4096 * 12);
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(t.data(), trace_dev, 4096 * 96, hipMemcpyDeviceToHost);
                const char *names[12] = {"poll + acquire issue", "stage wait (DMA landed)", "stage barrier", "epilogue issue",
                                         "restart (drain, publish, poll, acquire, first stage issue)", "acquire completion",
                                         "first-stage wait of a pass (stores drain)", "total", "cursor enters the next pass",
                                         "descriptor fetch at the start of a pass", "", ""};
                const int nwg = 512;
                for (int k = 0; k < 10; ++k) {
                    std::vector<double> v;
                    for (int b = 0; b < nwg; ++b) v.push_back(t[(size_t)b * 12 + k] * 0.01);
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "chain trace: %-60s p10 %9.1f median %9.1f p90 %9.1f us\n", names[k], v[nwg / 10],
                            v[nwg / 2], v[nwg * 9 / 10]);
                }
                if (const char *path = getenv("SR355_CHAIN_TRACE_CSV")) {
                    if (FILE *f = fopen(path, "w")) {
                        fprintf(f, "wg,xcc,se,cu,shader_mhz,total_us,poll_us,epilogue_us\n");
                        for (int b = 0; b < nwg; ++b) {
                            const unsigned long long *w = &t[(size_t)b * 12];
                            const unsigned hw = (unsigned)w[11], xcc = (unsigned)(w[11] >> 32) & 15;
                            fprintf(f, "%d,%u,%u,%u,%.1f,%.1f,%.1f,%.1f\n", b, xcc, (hw >> 13) & 7, (hw >> 8) & 15,
                                    (double)w[10] / (w[7] * 0.01), w[7] * 0.01, w[0] * 0.01, w[3] * 0.01);
                        }
                        fclose(f);
                    }
                }
            }
        }
#endif
        {
            std::lock_guard<std::mutex> lock(order->m);
            if (check_hip(hipEventRecord(order->err_copied, stream), "hipEventRecord")) return SR355_ERR_HIP;
            order->err_recorded = true;
        }
    }
    // nearest x2 + upconv1 + lrelu (:59), nearest x2 + upconv2 + lrelu (:60)
    if ((rc = conv(fb.L1, ws.nf_units, ws.nf_units, 2 * H, 2 * W, 1, fb.H1, ws.nf_units, 0, 0.2f, nullptr, 0.f,
                   nullptr, 0.f, nullptr, nullptr)))
        return rc;
    if ((rc = conv(fb.H1, ws.nf_units, ws.nf_units, 4 * H, 4 * W, 1, fb.H2, ws.nf_units, 0, 0.2f, nullptr, 0.f,
                   nullptr, 0.f, nullptr, nullptr)))
        return rc;
    // HR_conv + lrelu, conv_last (:61)
    if ((rc = conv(fb.H2, ws.nf_units, ws.nf_units, 4 * H, 4 * W, 0, fb.H3, ws.nf_units, 0, 0.2f, nullptr, 0.f,
                   nullptr, 0.f, nullptr, nullptr)))
        return rc;
    rc = conv(fb.H3, ws.nf_units, ws.nf_units, 4 * H, 4 * W, 0, nullptr, 0, 0, 1.f, nullptr, 0.f, nullptr, 0.f,
              nullptr, out_u8 ? nullptr : out_nchw, out_u8);
    if (rc) return rc;
    for (int p = 1; p < n_parts; ++p) {   // join: the caller's stream continues after every part
        if (check_hip(hipEventRecord(ss->join[p - 1], part_stream[p]), "hipEventRecord") ||
            check_hip(hipStreamWaitEvent(stream, ss->join[p - 1], 0), "hipStreamWaitEvent"))
            return SR355_ERR_HIP;
    }
    return SR355_OK;
}

static int forward_checks(const char *who, const sr355_generator_cfg *cfg, const void *packed, size_t packed_bytes,
                          size_t packed_need, const float *in_nchw, float *out_nchw, int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1 || !packed || !aligned16(packed) || packed_bytes < packed_need || !in_nchw ||
        !out_nchw) {
        set_error("%s: bad arguments (N=%d H=%d W=%d packed %zu of %zu bytes)", who, N, H, W, packed_bytes,
                  packed_need);
        return SR355_ERR_INVALID;
    }
    if ((size_t)16 * H * W * (cfg->num_features / unit_channels(cfg->dtype)) * 16 >= ((size_t)1 << 31)) {
        set_error("%s: %dx%d input makes one HR feature map exceed 2 GiB; tile the image", who, H, W);
        return SR355_ERR_UNSUPPORTED;
    }
    return SR355_OK;
}

int sr355_generator_forward(const sr355_generator_cfg *cfg, const void *packed, size_t packed_bytes,
                            const float *in_nchw, float *out_nchw, int N, int H, int W, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    if (!cfg_ok(cfg)) return SR355_ERR_INVALID;
    size_t total = 0;
    std::vector<ConvSpec> specs = conv_specs(*cfg, &total);
    int rc = forward_checks("generator_forward", cfg, packed, packed_bytes, total + pack_table_bytes(specs.size()), in_nchw, out_nchw, N, H, W);
    if (rc) return rc;
    const Workspace ws = carve(*cfg, N, H, W);
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < ws.total) {
        set_error("generator_forward: workspace must be 256-byte aligned and >= %zu bytes (got %zu)", ws.total,
                  workspace_bytes);
        return SR355_ERR_INVALID;
    }
    unsigned char *wsb = (unsigned char *)workspace;
    FwdBufs fb;
    fb.x0 = wsb + ws.x0;
    fb.blk_base = wsb + ws.bufA;
    fb.blk_stride = ws.bufB - ws.bufA;
    fb.blk_count = 2;
    fb.T = (float *)(wsb + ws.T), fb.R = (float *)(wsb + ws.R), fb.F = (float *)(wsb + ws.F);
    fb.L1 = wsb + ws.L1, fb.H1 = wsb + ws.H1, fb.H2 = wsb + ws.H2, fb.H3 = wsb + ws.H3;
    fb.chain_flags = (unsigned *)(wsb + ws.chain_flags), fb.chain_flag_bytes = ws.chain_flag_bytes;
    fb.chain_table = (ChainPass *)(wsb + ws.chain_table);
    return run_forward(cfg, specs, (const unsigned char *)packed, in_nchw, out_nchw, N, H, W, ws, fb,
                       (hipStream_t)stream_);
}

int sr355_generator_forward_u8(const sr355_generator_cfg *cfg, const void *packed, size_t packed_bytes,
                               const unsigned char *in_u8hwc, unsigned char *out_u8hwc, int N, int H, int W,
                               void *workspace, size_t workspace_bytes, void *stream_) {
    if (!cfg_ok(cfg)) return SR355_ERR_INVALID;
    size_t total = 0;
    std::vector<ConvSpec> specs = conv_specs(*cfg, &total);
    int rc = forward_checks("generator_forward_u8", cfg, packed, packed_bytes, total + pack_table_bytes(specs.size()),
                            reinterpret_cast<const float *>(in_u8hwc), reinterpret_cast<float *>(out_u8hwc), N, H, W);
    if (rc) return rc;
    const Workspace ws = carve(*cfg, N, H, W);
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < ws.total) {
        set_error("generator_forward_u8: workspace must be 256-byte aligned and >= %zu bytes (got %zu)", ws.total,
                  workspace_bytes);
        return SR355_ERR_INVALID;
    }
    unsigned char *wsb = (unsigned char *)workspace;
    FwdBufs fb;
    fb.x0 = wsb + ws.x0;
    fb.blk_base = wsb + ws.bufA;
    fb.blk_stride = ws.bufB - ws.bufA;
    fb.blk_count = 2;
    fb.T = (float *)(wsb + ws.T), fb.R = (float *)(wsb + ws.R), fb.F = (float *)(wsb + ws.F);
    fb.L1 = wsb + ws.L1, fb.H1 = wsb + ws.H1, fb.H2 = wsb + ws.H2, fb.H3 = wsb + ws.H3;
    fb.chain_flags = (unsigned *)(wsb + ws.chain_flags), fb.chain_flag_bytes = ws.chain_flag_bytes;
    fb.chain_table = (ChainPass *)(wsb + ws.chain_table);
    return run_forward(cfg, specs, (const unsigned char *)packed, nullptr, nullptr, N, H, W, ws, fb,
                       (hipStream_t)stream_, in_u8hwc, out_u8hwc);
}

void sr355_chain_timing(int on) { g_chain_timing.store(on ? 1 : 0); }

float sr355_chain_last_ms(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1.f;
    ChainEvents *cev = chain_events(dev);
    if (!cev || !cev->recorded) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(cev->after) != hipSuccess || hipEventElapsedTime(&ms, cev->before, cev->after) != hipSuccess)
        return -1.f;
    return ms;
}

int sr355_chain_status(int wait) {
    int dev = 0;
    if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return SR355_ERR_HIP;
    return chain_status_of(dev, wait != 0);
}

void sr355_chain_reenable(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (ChainOrder *order = chain_order(dev)) order->disabled.store(false, std::memory_order_relaxed);
}

int sr355_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int step, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1 || !(beta1 >= 0. && beta1 < 1.) ||
        !(beta2 >= 0. && beta2 < 1.) || !aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) ||
        !aligned16(exp_avg_sq)) {
        set_error("adam_step: bad arguments (n=%zu step=%d betas %g %g; buffers must be 16-byte aligned)", n, step,
                  (double)beta1, (double)beta2);
        return SR355_ERR_INVALID;
    }
    if (n == 0) return SR355_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    return launch_adam_flat(param, grad, exp_avg, exp_avg_sq, n, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                            (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(lr / bc1), (float)sqrt(bc2),
                            (hipStream_t)stream);
}

// ---- training: forward that keeps activations, backward -------------------------------------
// Saved activations (bf16 path only): blocked input, one dense-block buffer per dense block plus the
// body output, the LR feature map fed to upconv1 and the three HR activations.
struct Saved {
    size_t x0, blk, L1, H1, H2, H3, total;
    size_t blk_stride;
    int n_blk;
};

static Saved carve_saved(const sr355_generator_cfg &c, const Workspace &ws, int N, int H, int W) {
    Saved s;
    const size_t P = (size_t)N * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align256(bytes);
        return o;
    };
    s.n_blk = 3 * c.num_blocks + 1;
    s.blk_stride = align256(P * ws.blk_units * 16);
    s.x0 = take(P * ws.in_units * 16);
    s.blk = take(s.blk_stride * s.n_blk);
    s.L1 = take(P * ws.nf_units * 16);
    s.H1 = take(4 * P * ws.nf_units * 16);
    s.H2 = take(16 * P * ws.nf_units * 16);
    s.H3 = take(16 * P * ws.nf_units * 16);
    s.total = off;
    return s;
}

static bool train_ok(const sr355_generator_cfg *cfg) {
    if (!cfg_ok(cfg)) return false;
    if (cfg->dtype != SR355_DTYPE_BF16) {
        set_error("training path: bf16 kernels only (the exact-fp32 path trains conv by conv)");
        return false;
    }
    if (cfg->hr_split == SR355_HR_F16 || cfg->trunk_fp32 == SR355_TRUNK_RRDB) {
        set_error("training path: the fp16 HR side (hr_split = 2) and RRDB-only fp32 carriers (trunk_fp32 = 2) are "
                  "inference modes");
        return false;
    }
    return true;
}

size_t sr355_generator_saved_bytes(const sr355_generator_cfg *cfg, int N, int H, int W) {
    if (!train_ok(cfg) || N < 1 || H < 1 || W < 1) return 0;
    return carve_saved(*cfg, carve(*cfg, N, H, W), N, H, W).total;
}

size_t sr355_generator_train_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W) {
    if (!train_ok(cfg) || N < 1 || H < 1 || W < 1) return 0;
    return 3 * align256((size_t)N * H * W * cfg->num_features * 4);   // fp32 trunks T, R, F
}

int sr355_generator_forward_train(const sr355_generator_cfg *cfg, const void *packed, size_t packed_bytes,
                                  const float *in_nchw, float *out_nchw, int N, int H, int W, void *saved,
                                  size_t saved_bytes, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!train_ok(cfg)) return SR355_ERR_INVALID;
    size_t total = 0;
    std::vector<ConvSpec> specs = conv_specs(*cfg, &total);
    int rc = forward_checks("generator_forward_train", cfg, packed, packed_bytes, total + pack_table_bytes(specs.size()), in_nchw, out_nchw, N, H, W);
    if (rc) return rc;
    const Workspace ws = carve(*cfg, N, H, W);
    const Saved sv = carve_saved(*cfg, ws, N, H, W);
    const size_t trunk = align256((size_t)N * H * W * cfg->num_features * 4);
    if (!saved || (reinterpret_cast<uintptr_t>(saved) & 255) || saved_bytes < sv.total || !workspace ||
        (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < 3 * trunk) {
        set_error("generator_forward_train: need 256-byte aligned saved >= %zu and workspace >= %zu bytes", sv.total,
                  3 * trunk);
        return SR355_ERR_INVALID;
    }
    unsigned char *sb = (unsigned char *)saved, *wsb = (unsigned char *)workspace;
    FwdBufs fb;
    fb.x0 = sb + sv.x0;
    fb.blk_base = sb + sv.blk;
    fb.blk_stride = sv.blk_stride;
    fb.blk_count = sv.n_blk;
    fb.T = (float *)wsb, fb.R = (float *)(wsb + trunk), fb.F = (float *)(wsb + 2 * trunk);
    fb.L1 = sb + sv.L1, fb.H1 = sb + sv.H1, fb.H2 = sb + sv.H2, fb.H3 = sb + sv.H3;
    return run_forward(cfg, specs, (const unsigned char *)packed, in_nchw, out_nchw, N, H, W, ws, fb,
                       (hipStream_t)stream_);
}

// Transposed (data-gradient) weights.  Per dense block five tensors, step j = 4..1 producing the gradient
// of growth output x_j and step 0 the gradient of the block input x; then trunk_conv, upconv1, upconv2,
// HR_conv, conv_last and conv_first.  Offsets into the blob; every tensor is followed by nothing (the
// data-gradient convs use one shared all-zero bias at the start of the blob).
struct BwdPack {
    std::vector<size_t> rdb;   // [3*nb][5]: index (i*5 + j)
    size_t trunk, up1, up2, hr, last, first, zero_bias, table, total;
};

static BwdPack carve_bwd(const sr355_generator_cfg &c) {
    BwdPack p;
    const int nf = c.num_features, gc = c.growth_channels;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align256(bytes);
        return o;
    };
    {   // one all-zero bias for every data-gradient conv: as long as the widest of them
        int widest = nf > gc ? nf : gc;
        if (c.in_channels > widest) widest = c.in_channels;
        p.zero_bias = take((size_t)(cout_pad_for(widest) < 64 ? 64 : cout_pad_for(widest)) * sizeof(float));
    }
    p.rdb.resize((size_t)3 * c.num_blocks * 5);
    for (int i = 0; i < 3 * c.num_blocks; ++i)
        for (int j = 4; j >= 0; --j) {
            const int cin = nf + (4 - j) * gc, cout = j ? gc : nf;
            p.rdb[(size_t)i * 5 + j] = take((size_t)(cin / 8) * 9 * cout_pad_for(cout) * 16);
        }
    const size_t sq = (size_t)(nf / 8) * 9 * cout_pad_for(nf) * 16;
    p.trunk = take(sq), p.up1 = take(sq), p.up2 = take(sq), p.hr = take(sq);
    p.last = take((size_t)cin_units_for(c.out_channels, c.dtype) * 9 * cout_pad_for(nf) * 16);
    p.first = take((size_t)(nf / 8) * 9 * cout_pad_for(c.in_channels) * 16);
    p.table = take(((size_t)45 * c.num_blocks + 6) * sizeof(PackTJob));   // the pack jobs themselves (one launch packs all)
    p.total = off;
    return p;
}

size_t sr355_generator_packed_bwd_bytes(const sr355_generator_cfg *cfg) {
    if (!train_ok(cfg)) return 0;
    return carve_bwd(*cfg).total;
}

int sr355_generator_pack_bwd(const sr355_generator_cfg *cfg, const float *const *w_ptrs, int n_convs, void *packed,
                             size_t packed_bytes, void *stream_) {
    if (!train_ok(cfg)) return SR355_ERR_INVALID;
    hipStream_t stream = (hipStream_t)stream_;
    const BwdPack bp = carve_bwd(*cfg);
    const int nb = cfg->num_blocks, nf = cfg->num_features, gc = cfg->growth_channels, dt = cfg->dtype;
    if (!w_ptrs || n_convs != 1 + 15 * nb + 5 || !packed || !aligned16(packed) || packed_bytes < bp.total) {
        set_error("generator_pack_bwd: need %d weight tensors and %zu bytes (got %d, %zu)", 1 + 15 * nb + 5, bp.total,
                  n_convs, packed_bytes);
        return SR355_ERR_INVALID;
    }
    for (int i = 0; i < n_convs; ++i)
        if (!w_ptrs[i]) {
            set_error("generator_pack_bwd: tensor %d is null", i);
            return SR355_ERR_INVALID;
        }
    unsigned char *base = (unsigned char *)packed;
    // zero the whole blob once: padded input units (conv_last) and the shared bias must read as zero
    if (check_hip(hipMemsetAsync(base, 0, bp.total, stream), "hipMemsetAsync")) return SR355_ERR_HIP;
    std::vector<PackTJob> jobs;
    jobs.reserve((size_t)45 * nb + 6);
    auto launch_pack_weight_t = [&jobs](const float *w, int Co, int Ci, int ci_lo, int n_out, int, float scale, void *wp,
                                        int unit_off, int cout_pad, hipStream_t) -> int {
        jobs.push_back(PackTJob{w, wp, Co, Ci, ci_lo, n_out, unit_off, cout_pad, scale});
        return SR355_OK;
    };
    int rc;
    for (int i = 0; i < 3 * nb; ++i) {
        const float *const *w = w_ptrs + 1 + (size_t)i * 5;   // conv1..conv5 of dense block i
        const float s5 = (i % 3 == 2) ? 0.04f : 0.2f;          // 0.2 (dense skip) x 0.2 (RRDB skip on the third block)
        for (int j = 4; j >= 0; --j) {
            const int ci_lo = j ? nf + (j - 1) * gc : 0, n_out = j ? gc : nf;
            const int units = (nf + (4 - j) * gc) / 8, cp = cout_pad_for(n_out);
            unsigned char *dst = base + bp.rdb[(size_t)i * 5 + j];
            // gradient stack order: [dOut (nf) | dY4 | dY3 | dY2 | dY1]
            if ((rc = launch_pack_weight_t(w[4], nf, nf + 4 * gc, ci_lo, n_out, dt, s5, dst, 0, cp, stream))) return rc;
            for (int k = 4; k > j; --k) {   // conv_k (1-based) reads x_j when k > j
                if ((rc = launch_pack_weight_t(w[k - 1], gc, nf + (k - 1) * gc, ci_lo, n_out, dt, 1.f, dst,
                                               (nf + (4 - k) * gc) / 8, cp, stream)))
                    return rc;
            }
            (void)units;
        }
    }
    const float *const *tail = w_ptrs + 1 + (size_t)15 * nb;   // trunk_conv, upconv1, upconv2, HR_conv, conv_last
    const size_t sq_off[4] = {bp.trunk, bp.up1, bp.up2, bp.hr};
    for (int t = 0; t < 4; ++t)
        if ((rc = launch_pack_weight_t(tail[t], nf, nf, 0, nf, dt, 1.f, base + sq_off[t], 0, cout_pad_for(nf), stream)))
            return rc;
    if ((rc = launch_pack_weight_t(tail[4], cfg->out_channels, nf, 0, nf, dt, 1.f, base + bp.last, 0, cout_pad_for(nf),
                                   stream)))
        return rc;
    if ((rc = launch_pack_weight_t(w_ptrs[0], nf, cfg->in_channels, 0, cfg->in_channels, dt, 1.f, base + bp.first, 0,
                                   cout_pad_for(cfg->in_channels), stream)))
        return rc;
    if (jobs.size() > (size_t)45 * nb + 6) {
        set_error("generator_pack_bwd: job table overflow");
        return SR355_ERR_INVALID;
    }
    return launch_pack_weight_t_table(jobs.data(), (int)jobs.size(), dt, base + bp.table, stream);
}

// Backward workspace.
struct BwdWs {
    size_t dout, G3, G2, G1u, G1, G0u, gL1, gL1b, gT, gR, GB0, GB1, GB2, GBx[7], dfea, slabs, total;
};

static BwdWs carve_bwd_ws(const sr355_generator_cfg &c, const Workspace &ws, int N, int H, int W) {
    BwdWs b;
    const size_t P = (size_t)N * H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align256(bytes);
        return o;
    };
    const int ou = cin_units_for(c.out_channels, c.dtype);
    b.dout = take(16 * P * ou * 16);
    b.G3 = take(16 * P * ws.nf_units * 16);
    b.G2 = take(16 * P * ws.nf_units * 16);
    b.G1u = take(16 * P * ws.nf_units * 16);
    b.G1 = take(4 * P * ws.nf_units * 16);
    b.G0u = take(4 * P * ws.nf_units * 16);
    b.gL1 = take(P * c.num_features * 4);
    b.gL1b = take(P * ws.nf_units * 16);
    b.gT = take(P * c.num_features * 4);
    b.gR = take(P * c.num_features * 4);
    b.GB0 = take(P * ws.blk_units * 16);
    b.GB1 = take(P * ws.blk_units * 16);
    b.GB2 = take(P * ws.blk_units * 16);   // third gradient stack of the one-launch dense-block backward
    // seven more where the dense blocks run as one launch each (narrow images): the stacks of three RRDBs' nine blocks stay
    // whole until their weight gradients have been issued as one launch
    const bool narrow = use_dense_fused(&c, N, H, W);
    for (int k = 0; k < 7; ++k) b.GBx[k] = take(narrow ? P * ws.blk_units * 16 : 0);
    b.dfea = take(P * ws.nf_units * 16);
    size_t slab = 0;
    const int nf = c.num_features, gc = c.growth_channels;
    auto upd = [&](int cout, int cin, int n, int h, int w) {
        size_t s = wgrad_workspace_bytes(cout, cin, n, h, w);
        if (s > slab) slab = s;
    };
    for (int k = 0; k < 4; ++k) upd(gc, nf + k * gc, N, H, W);
    upd(nf, nf + 4 * gc, N, H, W);
    upd(nf, c.in_channels, N, H, W);
    upd(nf, nf, N, 2 * H, 2 * W);
    upd(nf, nf, N, 4 * H, 4 * W);
    upd(c.out_channels, nf, N, 4 * H, 4 * W);
    // a dense block's five at once, and three blocks' worth: an RRDB's reductions run as one launch
    const size_t block_slabs = wgrad_block_workspace_bytes(nf, gc, N, H, W);
    if (3 * block_slabs > slab) slab = 3 * block_slabs;
    if (narrow && 9 * block_slabs > slab) slab = 9 * block_slabs;
    b.slabs = take(slab);
    b.total = off;
    return b;
}

size_t sr355_generator_bwd_workspace_bytes(const sr355_generator_cfg *cfg, int N, int H, int W) {
    if (!train_ok(cfg) || N < 1 || H < 1 || W < 1 || cfg->out_channels > 64) return 0;
    return carve_bwd_ws(*cfg, carve(*cfg, N, H, W), N, H, W).total;
}

size_t sr355_generator_grads_floats(const sr355_generator_cfg *cfg) {
    if (!cfg_ok(cfg)) return 0;
    size_t n = 0;
    for (const ConvSpec &s : conv_specs(*cfg, nullptr)) n += (size_t)s.cout * s.cin * 9 + s.cout;
    return n;
}

// trunk sum -> blocked bf16 (the gradient reaching conv_first is the body path plus the global skip)
__global__ void add_trunks_to_act_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                         __bf16 *__restrict__ out, size_t n_units8, int W, int f32_units) {
    // one thread per (row, unit8, x): reads the two fp32 quads of that unit
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units8; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t t = i / W;
        const int u8 = (int)(t % (f32_units / 2));
        const size_t row = t / (f32_units / 2);
        __bf16 v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t idx = (row * f32_units + 2 * u8 + q) * W + x;
            const float4 va = a[idx], vb = b[idx];
            v[4 * q + 0] = (__bf16)(va.x + vb.x);
            v[4 * q + 1] = (__bf16)(va.y + vb.y);
            v[4 * q + 2] = (__bf16)(va.z + vb.z);
            v[4 * q + 3] = (__bf16)(va.w + vb.w);
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(out)[i] = pk;
    }
}

int sr355_generator_backward_phases(const sr355_generator_cfg *cfg) {
    if (!train_ok(cfg)) return SR355_ERR_INVALID;
    return cfg->num_blocks + 2;
}

// Phase p of the backward pass: 0 = conv_last .. trunk_conv, 1..num_blocks = RRDB (num_blocks - p),
// num_blocks + 1 = conv_first.  Phases must run in order; state between them lives in `workspace`.
int sr355_generator_backward_range(const sr355_generator_cfg *cfg, const void *packed_bwd, size_t packed_bwd_bytes,
                                   const void *saved, size_t saved_bytes, const float *grad_out_nchw, float *grads,
                                   size_t grads_floats, float *grad_in_nchw, int N, int H, int W, void *workspace,
                                   size_t workspace_bytes, int phase_begin, int phase_end, void *stream_) {
    if (!train_ok(cfg)) return SR355_ERR_INVALID;
    if (phase_begin < 0 || phase_end > cfg->num_blocks + 2 || phase_begin >= phase_end) {
        set_error("generator_backward: phase range [%d, %d) outside [0, %d)", phase_begin, phase_end,
                  cfg->num_blocks + 2);
        return SR355_ERR_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const std::vector<ConvSpec> specs = conv_specs(*cfg, nullptr);
    const BwdPack bp = carve_bwd(*cfg);
    const Workspace ws = carve(*cfg, N, H, W);
    const Saved sv = carve_saved(*cfg, ws, N, H, W);
    if (N < 1 || H < 1 || W < 1 || cfg->out_channels > 64) {
        set_error("generator_backward: bad shape");
        return SR355_ERR_INVALID;
    }
    const BwdWs bw = carve_bwd_ws(*cfg, ws, N, H, W);
    const size_t need_grads = sr355_generator_grads_floats(cfg);
    if (!packed_bwd || !aligned16(packed_bwd) || packed_bwd_bytes < bp.total || !saved ||
        (reinterpret_cast<uintptr_t>(saved) & 255) || saved_bytes < sv.total || !grad_out_nchw || !grads ||
        grads_floats < need_grads || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) ||
        workspace_bytes < bw.total) {
        set_error("generator_backward: buffers missing / misaligned / too small (packed %zu/%zu, saved %zu/%zu, grads "
                  "%zu/%zu floats, workspace %zu/%zu)",
                  packed_bwd_bytes, bp.total, saved_bytes, sv.total, grads_floats, need_grads, workspace_bytes,
                  bw.total);
        return SR355_ERR_INVALID;
    }
    const int nb = cfg->num_blocks, nf = cfg->num_features, gc = cfg->growth_channels, dt = cfg->dtype;
    const int nfu = ws.nf_units, bu = ws.blk_units;
    const unsigned char *pk = (const unsigned char *)packed_bwd;
    const unsigned char *sb = (const unsigned char *)saved;
    unsigned char *wb = (unsigned char *)workspace;
    const float *zero_bias = (const float *)(pk + bp.zero_bias);
    float *slabs = (float *)(wb + bw.slabs);
    float *gL1 = (float *)(wb + bw.gL1), *gT = (float *)(wb + bw.gT), *gR = (float *)(wb + bw.gR);

    // parameter-gradient offsets (state_dict order: weight, bias per conv)
    std::vector<size_t> goff(specs.size());
    {
        size_t o = 0;
        for (size_t i = 0; i < specs.size(); ++i) {
            goff[i] = o;
            o += (size_t)specs[i].cout * specs[i].cin * 9 + specs[i].cout;
        }
    }
    // The weight gradients depend on the data-gradient chain but nothing depends on them: they run on a side
    // stream beside the chain (a data gradient at training sizes fills a fraction of the chip).  Every wgrad
    // waits for what the main stream has issued so far; the main stream waits for a dense block's wgrads
    // before it overwrites the gradient stack they read (one block later), and for all of them at the end.
    // Not with the one-launch dense block: its workgroup takes a CU's whole LDS, so a weight-gradient workgroup
    // cannot share a CU with it, the two kernels only displace each other, and the event traffic doubles the host
    // time of the call (measured at 16 x 32 x 32: 7.2 ms per pixel iteration with the side stream, 6.8 without;
    // with one launch per conv, 64 x 64 and up, the side stream gains 2-10 %).  SR355_WGRAD_STREAM=0 / 1 forces it.
    const bool fused = use_dense_fused(cfg, N, H, W);
    const bool block_wgrad_ok = cout_pad_for(gc) == 32 && cout_pad_for(nf) <= 64;   // what launch_wgrad3x3_blocks takes
    static const int side_knob = [] { const char *e = getenv("SR355_WGRAD_STREAM"); return e ? atoi(e) : -1; }();
    SideStream *ss = (side_knob < 0 ? !fused : side_knob != 0) ? side_stream() : nullptr;
    std::unique_lock<std::mutex> side_lock;
    if (ss) side_lock = std::unique_lock<std::mutex>(ss->in_use);
    hipStream_t wstream = ss ? ss->stream[0] : stream;
    // `after_main`: the side stream first waits for everything issued on the main stream so far
    auto wgrad = [&](size_t conv_idx, const void *in, int in_units, int ups, const void *dy, int dy_units, int dy_off,
                     int h, int w, float scale, bool after_main = true) -> int {
        const ConvSpec &sp = specs[conv_idx];
        float *dw = grads + goff[conv_idx];
        if (ss && after_main &&
            (check_hip(hipEventRecord(ss->main_mark, stream), "hipEventRecord") ||
             check_hip(hipStreamWaitEvent(wstream, ss->main_mark, 0), "hipStreamWaitEvent")))
            return SR355_ERR_HIP;
        return launch_wgrad3x3(in, in_units, ups, dy, dy_units, dy_off, N, h, w, sp.cout, sp.cin, scale, 0, dw,
                               dw + (size_t)sp.cout * sp.cin * 9, slabs, wstream);
    };
    // side -> main hand-offs: mark(k) after the wgrads issued so far, await(k) before the main stream goes on
    auto side_mark = [&](int k) -> int {
        return ss ? check_hip(hipEventRecord(ss->side_mark[k], wstream), "hipEventRecord") : 0;
    };
    auto side_await = [&](int k) -> int {
        return ss ? check_hip(hipStreamWaitEvent(stream, ss->side_mark[k], 0), "hipStreamWaitEvent") : 0;
    };
    // (every call ends with the main stream waiting for the side stream, so the slab workspace is free here)
    // data-gradient conv: forward kernel on transposed weights, zero bias
    auto dgrad = [&](const void *in, int in_units, int cin_units, size_t w_off, int cout, int h, int w,
                     const void *mask, int mask_units, int mask_off, const float *res1, float r1w, const float *res2,
                     float r2w, float *out_f32, void *out_act, int out_units, int out_off, float *out_nc) -> int {
        ConvArgs a = conv_args_zero();
        a.in = (const unsigned char *)in;
        a.w = pk + w_off;
        a.bias = zero_bias;
        a.out_act = (unsigned char *)out_act;
        a.out_f32 = out_f32;
        a.out_nchw = out_nc;
        a.res1 = res1, a.res1_w = r1w, a.res1_scale = 1.f;
        a.res2 = res2, a.res2_w = r2w, a.res2_scale = 1.f;
        a.mask_act = (const unsigned char *)mask;
        a.mask_units = mask_units, a.mask_unit_off = mask_off, a.mask_slope = 0.2f;
        a.N = N, a.H = h, a.W = w;
        a.in_units = in_units;
        a.out_units = out_units, a.out_unit_off = out_off;
        a.cout_real = cout, a.cout_pad = cout_pad_for(cout);
        a.f32_units = cout / 4;
        return launch_conv3x3(a, dt, cin_units, 0, stream);
    };
    const size_t i_trunk = 1 + (size_t)15 * nb, i_up1 = i_trunk + 1, i_up2 = i_trunk + 2, i_hr = i_trunk + 3,
                 i_last = i_trunk + 4;
    const int ou = cin_units_for(cfg->out_channels, dt);
    int rc = SR355_OK;
    // Gradient stacks, rotated once per dense block.  Five launches per block: two stacks (the final step of a block
    // writes the next stack's units [0, nf) only after the weight gradients that read that stack are done).  One
    // launch per block: three, because the single launch also fills its own stack's growth units, which must not
    // be the stack the previous block's weight gradients are still reading.
    // One-launch dense blocks with the weight gradients on the caller's stream: the weight gradients of an RRDB's three
    // blocks go out as ONE launch behind the third block's data gradients (their workgroups keep every CU busy where one
    // block's cannot); the three stacks then have to stay whole until that launch, hence a fourth buffer.
    // SR355_WGRAD_RRDB = RRDBs per weight-gradient launch (0: one launch per dense block, 1, 2, 3; default 3: measured
    // 5.85 / 5.12 / 4.80 / 4.71 ms per pixel iteration at 16 patches of 32 x 32 with sixteen tiles a workgroup)
    static const int rrdb_knob = [] { const char *e = getenv("SR355_WGRAD_RRDB"); const int v = e ? atoi(e) : 3; return v < 0 ? 0 : v > 3 ? 3 : v; }();
    const bool rrdb_wgrad = fused && !ss && rrdb_knob > 0;
    const int rr_max = 3 * rrdb_knob;              // dense blocks per weight-gradient launch
    const int n_gb = rrdb_wgrad ? rr_max + 1 : fused ? 3 : 2;
    unsigned char *GB[10] = {wb + bw.GB0, wb + bw.GB1, wb + bw.GB2, wb + bw.GBx[0], wb + bw.GBx[1], wb + bw.GBx[2], wb + bw.GBx[3],
                             wb + bw.GBx[4], wb + bw.GBx[5], wb + bw.GBx[6]};
    if (phase_begin == 0) {
    // grad_out -> blocked bf16
    if ((rc = launch_nchw_to_blocked(grad_out_nchw, N, cfg->out_channels, 4 * H, 4 * W, dt, wb + bw.dout, ou, 0, stream)))
        return rc;
    // conv_last (:61)
    if ((rc = wgrad(i_last, sb + sv.H3, nfu, 0, wb + bw.dout, ou, 0, 4 * H, 4 * W, 1.f))) return rc;
    if ((rc = dgrad(wb + bw.dout, ou, ou, bp.last, nf, 4 * H, 4 * W, sb + sv.H3, nfu, 0, nullptr, 1.f, nullptr, 1.f,
                    nullptr, wb + bw.G3, nfu, 0, nullptr)))
        return rc;
    // HR_conv
    if ((rc = wgrad(i_hr, sb + sv.H2, nfu, 0, wb + bw.G3, nfu, 0, 4 * H, 4 * W, 1.f))) return rc;
    if ((rc = dgrad(wb + bw.G3, nfu, nfu, bp.hr, nf, 4 * H, 4 * W, sb + sv.H2, nfu, 0, nullptr, 1.f, nullptr, 1.f,
                    nullptr, wb + bw.G2, nfu, 0, nullptr)))
        return rc;
    // upconv2 (:60): input was nearest-x2(H1)
    if ((rc = wgrad(i_up2, sb + sv.H1, nfu, 1, wb + bw.G2, nfu, 0, 4 * H, 4 * W, 1.f))) return rc;
    if ((rc = dgrad(wb + bw.G2, nfu, nfu, bp.up2, nf, 4 * H, 4 * W, nullptr, 0, 0, nullptr, 1.f, nullptr, 1.f, nullptr,
                    wb + bw.G1u, nfu, 0, nullptr)))
        return rc;
    if ((rc = launch_pool2x2(wb + bw.G1u, N, 2 * H, 2 * W, nfu, dt, sb + sv.H1, nfu, 0, 0.2f, nullptr, wb + bw.G1, nfu, 0,
                             stream)))
        return rc;
    // upconv1 (:59): input was nearest-x2(L1); L1 = fea + trunk has no activation
    if ((rc = wgrad(i_up1, sb + sv.L1, nfu, 1, wb + bw.G1, nfu, 0, 2 * H, 2 * W, 1.f))) return rc;
    if ((rc = dgrad(wb + bw.G1, nfu, nfu, bp.up1, nf, 2 * H, 2 * W, nullptr, 0, 0, nullptr, 1.f, nullptr, 1.f, nullptr,
                    wb + bw.G0u, nfu, 0, nullptr)))
        return rc;
    if ((rc = launch_pool2x2(wb + bw.G0u, N, H, W, nfu, dt, nullptr, 0, 0, 1.f, gL1, wb + bw.gL1b, nfu, 0, stream)))
        return rc;
    // trunk_conv (:57): its input is the body output = x slice of the last dense-block buffer
    const unsigned char *body_out = sb + sv.blk + (size_t)(3 * nb) * sv.blk_stride;
    if ((rc = wgrad(i_trunk, body_out, bu, 0, wb + bw.gL1b, nfu, 0, H, W, 1.f))) return rc;
    if ((rc = dgrad(wb + bw.gL1b, nfu, nfu, bp.trunk, nf, H, W, nullptr, 0, 0, nullptr, 1.f, nullptr, 1.f, gR, GB[0],
                    bu, 0, nullptr)))
        return rc;
    }   // phase 0
    // RRDBs in reverse (:35-39, :18-24); the gradient-stack buffer alternates once per dense block
    int blocks_done = 0;   // dense blocks of this call whose wgrads have been issued
    WgradReduceJob pending[45];
    int n_pending = 0;
    const void *rr_act[9] = {}, *rr_stack[9] = {};   // the blocks whose weight gradients wait for their launch, in the order of
    float rr_s5[9] = {};                              // their data gradients
    float *rr_dw[9][5] = {}, *rr_db[9][5] = {};
    int rr_n = 0;
    auto rr_flush = [&]() -> int {
        if (rr_n == 0) return SR355_OK;
        int rc_ = launch_wgrad3x3_blocks(rr_n, rr_act, rr_stack, bu, nf, gc, N, H, W, rr_s5, rr_dw, rr_db, slabs, stream, pending);
        if (!rc_) rc_ = launch_wgrad_reduce_jobs(pending, 5 * rr_n, nf, gc, stream);
        rr_n = 0;
        return rc_;
    };
    const size_t block_slab_floats = wgrad_block_workspace_bytes(nf, gc, N, H, W) / sizeof(float);
    for (int b = nb - 1; b >= 0; --b) {
        const int phase = nb - b;
        if (phase < phase_begin || phase >= phase_end) continue;
        int cur = (3 * (nb - 1 - b)) % n_gb;
        for (int r = 2; r >= 0; --r) {
            const int i = 3 * b + r;
            const unsigned char *A = sb + sv.blk + (size_t)i * sv.blk_stride;   // forward activations of this block
            const size_t c1 = 1 + (size_t)i * 5;                                 // conv1 of this block in `specs`
            const float s5 = (r == 2) ? 0.04f : 0.2f;
            if (fused) {
                // the block's five data gradients in one launch, then its five weight gradients on the side stream
                const int nxt = (cur + 1) % n_gb;
                const int par = blocks_done & 1;
                // GB[nxt] was the stack of the block two before this one: its weight gradients must be done
                if (blocks_done > 1 && side_await(par)) return SR355_ERR_HIP;
                DenseArgs d;
                memset(&d, 0, sizeof(d));
                d.buf = GB[cur];
                d.growth_out = GB[cur];
                for (int k = 0; k < 4; ++k) {   // step k + 1 produces dY(4 - k)
                    d.w[k] = pk + bp.rdb[(size_t)i * 5 + (4 - k)];
                    d.bias[k] = zero_bias;
                    d.mask_off[k] = (nf + (3 - k) * gc) / 8;
                }
                d.w[4] = pk + bp.rdb[(size_t)i * 5];
                d.bias[4] = zero_bias;
                d.mask = A;
                d.mask_units = bu;
                d.mask_slope = 0.2f;
                d.slope = 1.f;
                d.N = N, d.H = H, d.W = W, d.units = bu;
                ConvArgs &a = d.tail;
                a = conv_args_zero();
                a.res1_scale = a.res2_scale = 1.f;
                if (r == 2) {          // dX3 = 0.2 gR + conv
                    a.res1 = gR, a.res1_w = 0.2f, a.out_f32 = gT;
                } else if (r == 1) {   // dX2 = dX3 + conv
                    a.res1 = gT, a.res1_w = 1.f, a.out_f32 = gT;
                } else {               // gradient of the RRDB input = gR + dX2 + conv
                    a.res1 = gT, a.res1_w = 1.f, a.res2 = gR, a.res2_w = 1.f, a.out_f32 = gR;
                }
                a.out_act = GB[nxt];
                a.out_units = bu;
                a.cout_real = a.cout_pad = nf;
                a.f32_units = nf / 4;
                if ((rc = launch_dense_block(d, stream))) return rc;
                float *dw[5], *db[5];
                for (int k = 0; k < 5; ++k) {
                    dw[k] = grads + goff[c1 + k];
                    db[k] = dw[k] + (size_t)specs[c1 + k].cout * specs[c1 + k].cin * 9;
                }
                if (rrdb_wgrad) {
                    const int q = rr_n++;
                    rr_act[q] = A, rr_stack[q] = GB[cur], rr_s5[q] = s5;
                    for (int k = 0; k < 5; ++k) rr_dw[q][k] = dw[k], rr_db[q][k] = db[k];
                    if (rr_n == rr_max && (rc = rr_flush())) return rc;
                    ++blocks_done;
                    cur = nxt;
                    continue;
                }
                if (ss && (check_hip(hipEventRecord(ss->main_mark, stream), "hipEventRecord") ||
                           check_hip(hipStreamWaitEvent(wstream, ss->main_mark, 0), "hipStreamWaitEvent")))
                    return SR355_ERR_HIP;
                // (the slab reductions of an RRDB's three blocks go out as one launch behind the third)
                if ((rc = launch_wgrad3x3_block(A, GB[cur], bu, nf, gc, N, H, W, s5, dw, db,
                                                slabs + (size_t)(n_pending / 5) * block_slab_floats, wstream,
                                                pending + n_pending)))
                    return rc;
                n_pending += 5;
                if (r == 0) {
                    if ((rc = launch_wgrad_reduce_jobs(pending, n_pending, nf, gc, wstream))) return rc;
                    n_pending = 0;
                }
                if (side_mark(par)) return SR355_ERR_HIP;
                ++blocks_done;
                cur = nxt;
                continue;
            }
            // the four growth convs' data gradients fill the stack; then its five weight gradients go to the side
            // stream behind one hand-off (conv5: dY5 = s5 * the gradient held in GB[cur] units [0, nf))
            for (int j = 4; j >= 1; --j) {
                const int in_units = (nf + (4 - j) * gc) / 8, out_off = in_units;
                if ((rc = dgrad(GB[cur], bu, in_units, bp.rdb[(size_t)i * 5 + j], gc, H, W, A, bu,
                                (nf + (j - 1) * gc) / 8, nullptr, 1.f, nullptr, 1.f, nullptr, GB[cur], bu, out_off,
                                nullptr)))
                    return rc;
            }
            {
                float *dw[5], *db[5];
                for (int k = 0; k < 5; ++k) {
                    dw[k] = grads + goff[c1 + k];
                    db[k] = dw[k] + (size_t)specs[c1 + k].cout * specs[c1 + k].cin * 9;
                }
                if (ss && (check_hip(hipEventRecord(ss->main_mark, stream), "hipEventRecord") ||
                           check_hip(hipStreamWaitEvent(wstream, ss->main_mark, 0), "hipStreamWaitEvent")))
                    return SR355_ERR_HIP;
                if (block_wgrad_ok) {
                    if ((rc = launch_wgrad3x3_block(A, GB[cur], bu, nf, gc, N, H, W, s5, dw, db, slabs, wstream))) return rc;
                } else {
                    // widths the batched launch is not built for (growth above 32, features above 64): conv by conv
                    for (int k = 0; k < 5; ++k) {
                        const int cin = nf + k * gc, cout = k < 4 ? gc : nf;
                        const int dy_unit = k < 4 ? (nf + (3 - k) * gc) / 8 : 0;   // stack order: dOut, dY4, dY3, dY2, dY1
                        if ((rc = launch_wgrad3x3(A, bu, 0, GB[cur], bu, dy_unit, N, H, W, cout, cin, k < 4 ? 1.f : s5, 0,
                                                  dw[k], db[k], slabs, wstream)))
                            return rc;
                    }
                }
            }
            const int nxt = (cur + 1) % n_gb;
            // the next write goes to GB[nxt] units [0, nf): the conv5 weight gradient of the dense block before
            // this one reads them, and this block's own stack is rewritten one block later
            const int par = blocks_done & 1;
            if (blocks_done > 0 && side_await(par ^ 1)) return SR355_ERR_HIP;
            if (side_mark(par)) return SR355_ERR_HIP;
            ++blocks_done;
            if (r == 2)        // dX3 = 0.2 gR + conv
                rc = dgrad(GB[cur], bu, bu, bp.rdb[(size_t)i * 5], nf, H, W, nullptr, 0, 0, gR, 0.2f, nullptr, 1.f, gT,
                           GB[nxt], bu, 0, nullptr);
            else if (r == 1)   // dX2 = dX3 + conv
                rc = dgrad(GB[cur], bu, bu, bp.rdb[(size_t)i * 5], nf, H, W, nullptr, 0, 0, gT, 1.f, nullptr, 1.f, gT,
                           GB[nxt], bu, 0, nullptr);
            else               // gradient of the RRDB input = gR + dX2 + conv
                rc = dgrad(GB[cur], bu, bu, bp.rdb[(size_t)i * 5], nf, H, W, nullptr, 0, 0, gT, 1.f, gR, 1.f, gR,
                           GB[nxt], bu, 0, nullptr);
            if (rc) return rc;
            cur = nxt;
        }
    }
    if ((rc = rr_flush())) return rc;
    if (phase_end < nb + 2) {   // join: the gradients of these phases are complete in stream order
        if (side_mark(0) || side_await(0)) return SR355_ERR_HIP;
        return SR355_OK;
    }
    // conv_first (:56): gradient of fea = body path (gR) + global skip (gL1)
    {
        const size_t n_units8 = (size_t)N * H * W * nfu;
        size_t blocks = (n_units8 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(add_trunks_to_act_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float4 *)gR,
                           (const float4 *)gL1, (__bf16 *)(wb + bw.dfea), n_units8, W, nf / 4);
        if (check_hip(hipGetLastError(), "add_trunks launch")) return SR355_ERR_HIP;
    }
    if ((rc = wgrad(0, sb + sv.x0, ws.in_units, 0, wb + bw.dfea, nfu, 0, H, W, 1.f))) return rc;
    if (grad_in_nchw)
        rc = dgrad(wb + bw.dfea, nfu, nfu, bp.first, cfg->in_channels, H, W, nullptr, 0, 0, nullptr, 1.f, nullptr, 1.f,
                   nullptr, nullptr, 0, 0, grad_in_nchw);
    if (rc) return rc;
    if (side_mark(0) || side_await(0)) return SR355_ERR_HIP;   // join
    return SR355_OK;
}

int sr355_generator_backward(const sr355_generator_cfg *cfg, const void *packed_bwd, size_t packed_bwd_bytes,
                             const void *saved, size_t saved_bytes, const float *grad_out_nchw, float *grads,
                             size_t grads_floats, float *grad_in_nchw, int N, int H, int W, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    if (!train_ok(cfg)) return SR355_ERR_INVALID;
    return sr355_generator_backward_range(cfg, packed_bwd, packed_bwd_bytes, saved, saved_bytes, grad_out_nchw, grads,
                                          grads_floats, grad_in_nchw, N, H, W, workspace, workspace_bytes, 0,
                                          cfg->num_blocks + 2, stream_);
}

}  // extern "C"
