// The layers around the 3x3 convs of the discriminator and of the VGG19 feature extractor, on the blocked layout,
// and the loss reductions of the training iteration:
//   * BatchNorm2d (batch statistics in training mode, running statistics in eval mode) + LeakyReLU, forward and
//     backward: nn.BatchNorm2d(out_f) + nn.LeakyReLU(0.2) of /root/reference/esrgan/models/esrgan_model.py:68-72;
//   * 2x2 / stride 2 max pooling, forward and backward, with nn.MaxPool2d's tie rule (the gradient goes to the first
//     maximum of a window in row-major order): vgg19.features' "M" layers, esrgan_model.py:93-98;
//   * mean |a - b| and mean (a - b)^2 with the gradient factor written in the same pass: nn.L1Loss / nn.MSELoss of
//     /root/reference/esrgan/train.py:114-115 as used at :160, :201-204, :214-219.
// All of it is memory-bound elementwise / reduction work: 16-byte accesses, fixed-order two-level sums (per-workgroup
// partials, then one pass over the partials), no float atomics, so results do not depend on scheduling.
#include <string.h>

#include "sr355_internal.h"

namespace sr355 {

namespace {
constexpr int AUX_THREADS = 256;
constexpr int BN_MAX_PARTS = 64;

__device__ __forceinline__ float block_sum(float v, float *red) {
    // 256 threads = 4 waves: lanes by shuffles, waves through LDS, in a fixed order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Four sums at once (the four channels of a quad): two barriers instead of eight.
__device__ __forceinline__ void block_sum4(float (&v)[4], float *red /* [16] */) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_down(v[j], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[4 * wave + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = red[j] + red[4 + j] + red[8 + j] + red[12 + j];
}

// Sum over the 64 lanes of a wave in a fixed order, result in every lane.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return __shfl(v, 0, 64);
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm statistics.  x is an fp32 trunk tensor [rows = N*H][C/4][W][4].  Workgroup (part p, quad cq) owns the
// rows p, p + parts, ... of the four channels of quad cq, all 256 threads striding over their (row, pixel) items.
// It writes the mean of ITS items and the sum of squared deviations from that mean (the second loop re-reads what the
// first one just pulled into L2).  E[x^2] - E[x]^2 in fp32 would lose the variance of a channel whose mean dominates
// it; per-part means combined afterwards (Chan et al.) do not.
__device__ __forceinline__ int bn_part_rows(int rows, int parts, int p) { return (rows - p + parts - 1) / parts; }

__global__ void bn_partial_stats_kernel(const float4 *__restrict__ x, int rows, int Q, int W, int parts,
                                        float *__restrict__ pmean /* [groups][parts][4Q] */, float *__restrict__ pm2) {
    __shared__ float red[16];
    const int cq = blockIdx.y, p = blockIdx.x;
    // group blockIdx.z: `rows` consecutive rows of x with statistics of their own (the two halves of a batch that stand
    // for two separate calls of the layer)
    x += (size_t)blockIdx.z * rows * Q * W;
    pmean += (size_t)blockIdx.z * parts * 4 * Q, pm2 += (size_t)blockIdx.z * parts * 4 * Q;
    const int items = bn_part_rows(rows, parts, p) * W;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = threadIdx.x; i < items; i += AUX_THREADS) {
        const int r = i / W, px = i - r * W;
        const float4 v = x[((size_t)(p + r * parts) * Q + cq) * W + px];
        s[0] += v.x, s[1] += v.y, s[2] += v.z, s[3] += v.w;
    }
    block_sum4(s, red);
    const float inv = 1.0f / (float)items;
    const float mean[4] = {s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv};
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = threadIdx.x; i < items; i += AUX_THREADS) {
        const int r = i / W, px = i - r * W;
        const float4 v = x[((size_t)(p + r * parts) * Q + cq) * W + px];
        const float a = v.x - mean[0], b = v.y - mean[1], c = v.z - mean[2], d = v.w - mean[3];
        q[0] += a * a, q[1] += b * b, q[2] += c * c, q[3] += d * d;
    }
    block_sum4(q, red);
    if (threadIdx.x < 4) {
        pmean[(size_t)p * 4 * Q + 4 * cq + threadIdx.x] = mean[threadIdx.x];
        pm2[(size_t)p * 4 * Q + 4 * cq + threadIdx.x] = q[threadIdx.x];
    }
}

// One wave per channel, one lane per part (parts <= 64): batch mean / 1/sqrt(var + eps) for the apply pass, and the
// running buffers as nn.BatchNorm2d keeps them (momentum update with the unbiased variance, batch counter).
__global__ void bn_finalize_kernel(const float *__restrict__ pmean, const float *__restrict__ pm2, int parts, int C, int rows,
                                   int W, float m, float eps, float momentum, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, long long *__restrict__ num_batches,
                                   float *__restrict__ save_mean, float *__restrict__ save_rstd, int groups) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c == 0 && lane == 0 && num_batches) *num_batches += groups;
    if (c >= C) return;
    // the groups in order: each is one call of the layer as far as the running buffers are concerned
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        const float *pm = pmean + (size_t)g * parts * C, *pq = pm2 + (size_t)g * parts * C;
        float n_k = 0.f, mean_k = 0.f, m2_k = 0.f;
        if (lane < parts) {
            n_k = (float)(bn_part_rows(rows, parts, lane) * W);
            mean_k = pm[(size_t)lane * C + c];
            m2_k = pq[(size_t)lane * C + c];
        }
        const float mean = wave_sum(n_k * mean_k) / m;
        const float dlt = mean_k - mean;
        const float q = wave_sum(m2_k + n_k * dlt * dlt);
        const float var = q / m;
        if (lane == 0) {
            save_mean[(size_t)g * C + c] = mean;
            save_rstd[(size_t)g * C + c] = 1.0f / sqrtf(var + eps);
        }
        rm = (1.f - momentum) * rm + momentum * mean;
        rv = (1.f - momentum) * rv + momentum * (m > 1.f ? q / (m - 1.f) : var);
    }
    if (lane) return;
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

// eval mode: the statistics are the running buffers
__global__ void bn_eval_stats_kernel(const float *__restrict__ running_mean, const float *__restrict__ running_var, int C,
                                     float eps, float *__restrict__ save_mean, float *__restrict__ save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = running_mean[c];
    save_rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = lrelu(gamma * (x - mean) * rstd + beta), fp32 trunk in, blocked bf16 out.  One thread per (row, unit, pixel):
// the two fp32 quads of an 8-channel unit in, 16 bytes out.
__global__ void bn_lrelu_apply_kernel(const float4 *__restrict__ x, size_t n_items, int U, int W,
                                      const float *__restrict__ mean, const float *__restrict__ rstd,
                                      const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
                                      __bf16 *__restrict__ out, int out_units, int out_unit_off, int group_rows) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_items; i += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % W);
        const size_t t = i / W;
        const int u = (int)(t % U);
        const size_t row = t / U;
        const int gs = (int)(row / (size_t)group_rows) * 8 * U;      // this row's group: its statistics start here
        __bf16 v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 xv = x[(row * (2 * U) + 2 * u + q) * W + px];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 8 * u + 4 * q + j;
                const float z = gamma[c] * ((xs[j] - mean[gs + c]) * rstd[gs + c]) + beta[c];
                v[4 * q + j] = (__bf16)(z > 0.f ? z : z * slope);
            }
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(out)[(row * out_units + out_unit_off + u) * W + px] = pk;
    }
}

// The four gradient values of channel quad cq of one pixel: from a blocked bf16 tensor, or (DYF32) from an fp32 tensor
// in the layout of x, [rows][Q][W][4] — what a data-gradient conv writes when its result feeds this backward pass
// (rounding the gradient to bf16 first costs BatchNorm's backward most of its accuracy: it subtracts the gradient's
// mean, and an adversarial loss's gradient is nearly constant over the map).
template <bool DYF32>
__device__ __forceinline__ void bn_load_dy4(const void *dy, size_t row, int dy_units, int dy_unit_off, int Q, int cq, int W,
                                            int px, float (&g)[4]) {
    if constexpr (DYF32) {
        const float4 v = reinterpret_cast<const float4 *>(dy)[(row * Q + cq) * W + px];
        g[0] = v.x, g[1] = v.y, g[2] = v.z, g[3] = v.w;
    } else {
        const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __bf16 *>(dy) +
                                                           ((row * dy_units + dy_unit_off + (cq >> 1)) * W + px) * 8 +
                                                           (cq & 1) * 4);
        __bf16 gv[4];
        __builtin_memcpy(gv, &raw, 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = (float)gv[j];
    }
}

// Backward, pass 1: g = dy * lrelu'(z); per-workgroup sums of g and g * xhat for the four channels of a quad.
template <bool DYF32>
__global__ void bn_bwd_partial_kernel(const float4 *__restrict__ x, const void *__restrict__ dy, int dy_units,
                                      int dy_unit_off, int rows, int Q, int W, int parts, const float *__restrict__ mean,
                                      const float *__restrict__ rstd, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float slope, float *__restrict__ pg,
                                      float *__restrict__ pgx) {
    __shared__ float red[16];
    const int cq = blockIdx.y, p = blockIdx.x, grp = blockIdx.z;     // group: `rows` consecutive rows, own statistics
    mean += (size_t)grp * 4 * Q, rstd += (size_t)grp * 4 * Q;
    pg += (size_t)grp * parts * 4 * Q, pgx += (size_t)grp * parts * 4 * Q;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mu[j] = mean[4 * cq + j], rs[j] = rstd[4 * cq + j], ga[j] = gamma[4 * cq + j], be[j] = beta[4 * cq + j];
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    const int items = bn_part_rows(rows, parts, p) * W;
#pragma unroll 2
    for (int i = threadIdx.x; i < items; i += AUX_THREADS) {
        const int r = i / W, px = i - r * W;
        const size_t row = (size_t)grp * rows + (size_t)(p + r * parts);
        const float4 xv = x[(row * Q + cq) * W + px];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float gv[4];
        bn_load_dy4<DYF32>(dy, row, dy_units, dy_unit_off, Q, cq, W, px, gv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - mu[j]) * rs[j];
            const float z = ga[j] * xh + be[j];
            const float g = gv[j] * (z > 0.f ? 1.f : slope);
            sg[j] += g, sx[j] += g * xh;
        }
    }
    block_sum4(sg, red);
    block_sum4(sx, red);
    if (threadIdx.x < 4) {
        pg[(size_t)p * 4 * Q + 4 * cq + threadIdx.x] = sg[threadIdx.x];
        pgx[(size_t)p * 4 * Q + 4 * cq + threadIdx.x] = sx[threadIdx.x];
    }
}

// One wave per channel, one lane per part.
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ pg, const float *__restrict__ pgx, int parts, int C,
                                       float *__restrict__ dbeta, float *__restrict__ dgamma) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const size_t g = blockIdx.y;                                    // group
    pg += g * parts * C, pgx += g * parts * C;
    const float s = wave_sum(lane < parts ? pg[(size_t)lane * C + c] : 0.f);
    const float q = wave_sum(lane < parts ? pgx[(size_t)lane * C + c] : 0.f);
    if (lane == 0) dbeta[g * C + c] = s, dgamma[g * C + c] = q;
}

// Backward, pass 2: dx = gamma * rstd * (g - dbeta / m - xhat * dgamma / m) in training mode (the batch statistics
// depend on x), gamma * rstd * g in eval mode; blocked bf16 out, the layout the conv's gradients read.
template <bool DYF32>
__global__ void bn_bwd_apply_kernel(const float4 *__restrict__ x, const void *__restrict__ dy, int dy_units,
                                    int dy_unit_off, size_t n_items, int U, int W, const float *__restrict__ mean,
                                    const float *__restrict__ rstd, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, const float *__restrict__ dbeta,
                                    const float *__restrict__ dgamma, float inv_m, float slope, int training,
                                    __bf16 *__restrict__ dx, int dx_units, int dx_unit_off, int group_rows) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_items; i += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % W);
        const size_t t = i / W;
        const int u = (int)(t % U);
        const size_t row = t / U;
        const int gs = (int)(row / (size_t)group_rows) * 8 * U;      // this row's group: statistics and sums start here
        float gv[8];
        __bf16 v[8];
        {
            float g0[4], g1[4];
            bn_load_dy4<DYF32>(dy, row, dy_units, dy_unit_off, 2 * U, 2 * u, W, px, g0);
            bn_load_dy4<DYF32>(dy, row, dy_units, dy_unit_off, 2 * U, 2 * u + 1, W, px, g1);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = g0[j], gv[4 + j] = g1[j];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 xv = x[(row * (2 * U) + 2 * u + q) * W + px];
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 8 * u + 4 * q + j;
                const float xh = (xs[j] - mean[gs + c]) * rstd[gs + c];
                const float z = gamma[c] * xh + beta[c];
                float g = gv[4 * q + j] * (z > 0.f ? 1.f : slope);
                if (training) g = g - dbeta[gs + c] * inv_m - xh * dgamma[gs + c] * inv_m;
                v[4 * q + j] = (__bf16)(gamma[c] * rstd[gs + c] * g);
            }
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(dx)[(row * dx_units + dx_unit_off + u) * W + px] = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2x2 / stride 2 max pooling on blocked bf16 [N][H][U][W][8] -> [N][H/2][U][W/2][8] (odd trailing row / column dropped).
__global__ void maxpool2x2_fwd_kernel(const uint4 *__restrict__ in, int H, int W, int U, size_t n_out,
                                      uint4 *__restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        size_t t = i / Wo;
        const int u = (int)(t % U);
        t /= U;
        const int y = (int)(t % Ho);
        const size_t n = t / Ho;
        __bf16 w[4][8], m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 raw = in[((n * H + 2 * y + (k >> 1)) * U + u) * W + 2 * x + (k & 1)];
            __builtin_memcpy(w[k], &raw, 16);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float best = (float)w[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k) best = (float)w[k][e] > best ? (float)w[k][e] : best;
            m[e] = (__bf16)best;
        }
        uint4 pk;
        __builtin_memcpy(&pk, m, 16);
        out[i] = pk;
    }
}

// dx of the same pooling: dy goes to the first maximum of its window (row-major order), zeros elsewhere and on an odd
// trailing row / column.  One thread per INPUT item, so every byte of dx is written.
__global__ void maxpool2x2_bwd_kernel(const uint4 *__restrict__ in, const uint4 *__restrict__ dy, int H, int W, int U,
                                      size_t n_in, uint4 *__restrict__ dx) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int u = (int)(t % U);
        t /= U;
        const int y = (int)(t % H);
        const size_t n = t / H;
        uint4 pk = make_uint4(0, 0, 0, 0);
        const int yo = y >> 1, xo = x >> 1;
        if (yo < Ho && xo < Wo) {
            __bf16 w[4][8], g[8], r[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 raw = in[((n * H + 2 * yo + (k >> 1)) * U + u) * W + 2 * xo + (k & 1)];
                __builtin_memcpy(w[k], &raw, 16);
            }
            const uint4 graw = dy[((n * Ho + yo) * U + u) * Wo + xo];
            __builtin_memcpy(g, &graw, 16);
            const int me = (y & 1) * 2 + (x & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int arg = 0;
                float best = (float)w[0][e];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if ((float)w[k][e] > best) best = (float)w[k][e], arg = k;
                r[e] = arg == me ? g[e] : (__bf16)0.f;
            }
            __builtin_memcpy(&pk, r, 16);
        }
        dx[i] = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mean |a - b| (mode 0) or mean (a - b)^2 (mode 1); b == null: the constant b_const.  grad (optional) receives
// d(loss)/da = sign(a - b) / n  or  2 (a - b) / n.
__global__ void loss_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, float b_const, size_t n,
                                    int mode, float inv_n, float *__restrict__ grad, float *__restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 av = reinterpret_cast<const float4 *>(a)[i];
        const float4 bv = b ? reinterpret_cast<const float4 *>(b)[i] : make_float4(b_const, b_const, b_const, b_const);
        const float d[4] = {av.x - bv.x, av.y - bv.y, av.z - bv.z, av.w - bv.w};
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (mode == 0) {
                s += fabsf(d[j]);
                g[j] = d[j] > 0.f ? inv_n : (d[j] < 0.f ? -inv_n : 0.f);
            } else {
                s += d[j] * d[j];
                g[j] = 2.f * d[j] * inv_n;
            }
        }
        if (grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    if (blockIdx.x == 0) {   // tail of a length that is not a multiple of 4
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            const float d = a[i] - (b ? b[i] : b_const);
            if (mode == 0) {
                s += fabsf(d);
                if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
            } else {
                s += d * d;
                if (grad) grad[i] = 2.f * d * inv_n;
            }
        }
    }
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void loss_finalize_kernel(const float *__restrict__ partial, int parts, float inv_n, float *__restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < parts; i += blockDim.x) s += partial[i];
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = tot * inv_n;
}

unsigned aux_grid(size_t items) {
    size_t b = (items + AUX_THREADS - 1) / AUX_THREADS;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
int bn_parts(int rows) { return rows < BN_MAX_PARTS ? rows : BN_MAX_PARTS; }
}  // namespace

size_t bn_workspace_floats(int N, int H, int W, int C) {
    (void)W;
    return (size_t)2 * bn_parts(N * H) * C;
}

// `groups` (a divisor of N): the batch is that many consecutive sub-batches, each normalised with statistics of its own and
// counted as one call by the running buffers, in order -- what `groups` calls on the slices of x do, in one set of launches.
// save_mean / save_rstd (and dgamma / dbeta below) are [groups][C]; the workspace is `groups` times a sub-batch's.
int launch_bn_lrelu_fwd(const float *x, int N, int H, int W, int C, const float *gamma, const float *beta, float eps,
                        float slope, int training, float momentum, float *running_mean, float *running_var,
                        long long *num_batches, float *save_mean, float *save_rstd, void *out_act, int out_units,
                        int out_unit_off, float *workspace, hipStream_t stream, int groups) {
    const int rows_g = (N / groups) * H, Q = C / 4, U = C / 8, parts = bn_parts(rows_g);
    const float m = (float)((double)(N / groups) * H * W);
    if (training) {
        float *pmean = workspace, *pm2 = workspace + (size_t)groups * parts * C;
        hipLaunchKernelGGL(bn_partial_stats_kernel, dim3(parts, Q, groups), dim3(AUX_THREADS), 0, stream, (const float4 *)x,
                           rows_g, Q, W, parts, pmean, pm2);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, pmean, pm2, parts, C, rows_g, W, m,
                           eps, momentum, running_mean, running_var, num_batches, save_mean, save_rstd, groups);
    } else {
        // (eval mode: every group reads the running buffers)
        for (int g = 0; g < groups; ++g)
            hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_mean, running_var,
                               C, eps, save_mean + (size_t)g * C, save_rstd + (size_t)g * C);
    }
    const size_t n_items = (size_t)N * H * U * W;
    hipLaunchKernelGGL(bn_lrelu_apply_kernel, dim3(aux_grid(n_items)), dim3(AUX_THREADS), 0, stream, (const float4 *)x,
                       n_items, U, W, save_mean, save_rstd, gamma, beta, slope, (__bf16 *)out_act, out_units, out_unit_off,
                       rows_g);
    return check_hip(hipGetLastError(), "bn_lrelu_fwd launch");
}

int launch_bn_lrelu_bwd(const float *x, const void *dy, int dy_units, int dy_unit_off, int N, int H, int W, int C,
                        const float *gamma, const float *beta, const float *save_mean, const float *save_rstd,
                        float slope, int training, void *dx, int dx_units, int dx_unit_off, float *dgamma, float *dbeta,
                        float *workspace, hipStream_t stream, int groups) {
    const int rows_g = (N / groups) * H, Q = C / 4, U = C / 8, parts = bn_parts(rows_g);
    const float m = (float)((double)(N / groups) * H * W);
    float *pg = workspace, *pgx = workspace + (size_t)groups * parts * C;
    const bool dy_f32 = dy_units == 0;     // the gradient arrives as fp32 [N][H][C/4][W][4]
    if (dy_f32)
        hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(parts, Q, groups), dim3(AUX_THREADS), 0, stream,
                           (const float4 *)x, dy, dy_units, dy_unit_off, rows_g, Q, W, parts, save_mean, save_rstd, gamma,
                           beta, slope, pg, pgx);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(parts, Q, groups), dim3(AUX_THREADS), 0, stream,
                           (const float4 *)x, dy, dy_units, dy_unit_off, rows_g, Q, W, parts, save_mean, save_rstd, gamma,
                           beta, slope, pg, pgx);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4, groups), dim3(256), 0, stream, pg, pgx, parts, C, dbeta,
                       dgamma);
    const size_t n_items = (size_t)N * H * U * W;
    if (dy_f32)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(aux_grid(n_items)), dim3(AUX_THREADS), 0, stream,
                           (const float4 *)x, dy, dy_units, dy_unit_off, n_items, U, W, save_mean, save_rstd, gamma, beta,
                           dbeta, dgamma, 1.0f / m, slope, training, (__bf16 *)dx, dx_units, dx_unit_off, rows_g);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(aux_grid(n_items)), dim3(AUX_THREADS), 0, stream,
                           (const float4 *)x, dy, dy_units, dy_unit_off, n_items, U, W, save_mean, save_rstd, gamma, beta,
                           dbeta, dgamma, 1.0f / m, slope, training, (__bf16 *)dx, dx_units, dx_unit_off, rows_g);
    return check_hip(hipGetLastError(), "bn_lrelu_bwd launch");
}

int launch_maxpool2x2_fwd(const void *in, int N, int H, int W, int U, void *out, hipStream_t stream) {
    const size_t n_out = (size_t)N * (H >> 1) * U * (W >> 1);
    if (n_out == 0) return SR355_OK;
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(aux_grid(n_out)), dim3(AUX_THREADS), 0, stream, (const uint4 *)in, H, W,
                       U, n_out, (uint4 *)out);
    return check_hip(hipGetLastError(), "maxpool2x2_fwd launch");
}

int launch_maxpool2x2_bwd(const void *in, const void *dy, int N, int H, int W, int U, void *dx, hipStream_t stream) {
    const size_t n_in = (size_t)N * H * U * W;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(aux_grid(n_in)), dim3(AUX_THREADS), 0, stream, (const uint4 *)in,
                       (const uint4 *)dy, H, W, U, n_in, (uint4 *)dx);
    return check_hip(hipGetLastError(), "maxpool2x2_bwd launch");
}

// ---------------------------------------------------------------------------------------------------------------
// nn.PixelShuffle(2) on the blocked layout (the sub-pixel stages of /root/reference/CycleGAN/CycleGAN_w_v1/CycleGAN.py:
// 117-129 and /root/reference/condtional VAE/model.py:118-133): in [N][H][4C/c][W][c] -> out [N][2H][C/c][2W][c] with
// out[n][2y+dy][ch][2x+dx] = in[n][y][4 ch + 2 dy + dx][x]; c = 8 (bf16) or 4 (fp32) channels per 16-byte unit.
// One thread per 16-byte unit of the LARGER-resolution tensor; INVERSE scatters the other way (the gradient).
template <typename T, bool INVERSE>
__global__ void pixel_shuffle2_kernel(const T *__restrict__ src, T *__restrict__ dst, int H, int W, int Uo, size_t n_units) {
    constexpr int CPU = 16 / sizeof(T);
    const int Ho = 2 * H, Wo = 2 * W, Ui = 4 * Uo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_units; i += (size_t)gridDim.x * blockDim.x) {
        // (n, yo, uo, xo): a unit of the high-resolution tensor
        const int xo = (int)(i % Wo);
        size_t t = i / Wo;
        const int uo = (int)(t % Uo);
        t /= Uo;
        const int yo = (int)(t % Ho);
        const size_t n = t / Ho;
        const int y = yo >> 1, x = xo >> 1, off = 2 * (yo & 1) + (xo & 1);
        T v[CPU];
        if (INVERSE) {
            const uint4 raw = reinterpret_cast<const uint4 *>(src)[i];
            __builtin_memcpy(v, &raw, 16);
        }
#pragma unroll
        for (int j = 0; j < CPU; ++j) {
            const int ci = 4 * (uo * CPU + j) + off;       // channel of the low-resolution tensor
            const size_t lo = ((((n * H + y) * Ui + ci / CPU) * W + x) * CPU) + ci % CPU;
            if (INVERSE) dst[lo] = v[j];
            else v[j] = src[lo];
        }
        if (!INVERSE) {
            uint4 pk;
            __builtin_memcpy(&pk, v, 16);
            reinterpret_cast<uint4 *>(dst)[i] = pk;
        }
    }
}

int launch_pixel_shuffle2(const void *src, int N, int H, int W, int c_out, int dtype, int inverse, void *dst,
                          hipStream_t stream) {
    const int cpu = unit_channels(dtype);
    const int Uo = c_out / cpu;
    const size_t n_units = (size_t)N * 2 * H * Uo * 2 * W;
    if (n_units == 0) return SR355_OK;
    const dim3 grid(aux_grid(n_units)), block(AUX_THREADS);
    if (dtype == SR355_DTYPE_BF16) {
        if (inverse)
            hipLaunchKernelGGL((pixel_shuffle2_kernel<__bf16, true>), grid, block, 0, stream, (const __bf16 *)src,
                               (__bf16 *)dst, H, W, Uo, n_units);
        else
            hipLaunchKernelGGL((pixel_shuffle2_kernel<__bf16, false>), grid, block, 0, stream, (const __bf16 *)src,
                               (__bf16 *)dst, H, W, Uo, n_units);
    } else {
        if (inverse)
            hipLaunchKernelGGL((pixel_shuffle2_kernel<float, true>), grid, block, 0, stream, (const float *)src,
                               (float *)dst, H, W, Uo, n_units);
        else
            hipLaunchKernelGGL((pixel_shuffle2_kernel<float, false>), grid, block, 0, stream, (const float *)src,
                               (float *)dst, H, W, Uo, n_units);
    }
    return check_hip(hipGetLastError(), "pixel_shuffle2 launch");
}

int loss_parts(size_t n) {
    const size_t b = (n / 4 + AUX_THREADS - 1) / AUX_THREADS;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int launch_loss(const float *a, const float *b, float b_const, size_t n, int mode, float *loss, float *grad,
                float *workspace, hipStream_t stream) {
    const int parts = loss_parts(n);
    const float inv_n = (float)(1.0 / (double)n);
    hipLaunchKernelGGL(loss_partial_kernel, dim3(parts), dim3(AUX_THREADS), 0, stream, a, b, b_const, n, mode, inv_n, grad,
                       workspace);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(AUX_THREADS), 0, stream, workspace, parts, inv_n, loss);
    return check_hip(hipGetLastError(), "loss launch");
}

}  // namespace sr355
