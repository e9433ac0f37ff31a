// Layout conversion at the module boundary and weight packing.
//
// The reference keeps NCHW fp32 activations and OIHW fp32 weights everywhere
// (/root/reference/esrgan/models/esrgan_model.py:10-14, :55-62).  The kernels here move
// between that and the blocked activation layout [N][H][U][W][16 B] and the packed weight
// layout [cin_units][9][cout_pad][16 B] consumed by the MFMA convolution.
#include "sr355_internal.h"

namespace sr355 {

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
template <> __device__ __forceinline__ _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }

// One thread per (n, y, unit, x): gathers CPU channels from NCHW and writes one 16-byte unit.
// split != 0 (bf16): channels [0,C) hold bf16(x) and channels [C,2C) the rounding residue
// bf16(x - bf16(x)), so a conv over both sees x to ~16 bits.
// U8: src is uint8 [N][H][W][C] (an image as PIL / numpy hold it) and the value is byte / 255 in fp32, what
// torchvision's ToTensor hands the reference (/root/reference/esrgan/test.py:69, esrgan/dataset.py:67-68).
template <typename T, bool U8 = false>
__global__ void nchw_to_blocked_kernel(const void *__restrict__ src_, T *__restrict__ dst, int N, int C,
                                       int H, int W, int U, int split) {
    const float *src = static_cast<const float *>(src_);
    const unsigned char *src8 = static_cast<const unsigned char *>(src_);
    auto value = [&](int n, int c, int y, int x) -> float {
        if constexpr (U8) return (float)src8[(((size_t)n * H + y) * W + x) * C + c] / 255.0f;
        else return src[(((size_t)n * C + c) * H + y) * W + x];
    };
    constexpr int CPU = 16 / sizeof(T);
    const size_t total = (size_t)N * H * U * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int u = (int)(t % U);
        t /= U;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        T v[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) {
            const int c = u * CPU + e;
            float f = 0.f;
            if (c < C) {
                f = value(n, c, y, x);
            } else if (split && c < 2 * C) {
                const float full = value(n, c - C, y, x);
                f = full - (float)from_f32<T>(full);
            }
            v[e] = from_f32<T>(f);
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(dst)[i] = pk;
    }
}

// One thread per (n, c, y, x) of the NCHW destination.
template <typename T>
__global__ void blocked_to_nchw_kernel(const T *__restrict__ src, float *__restrict__ dst, int N, int C,
                                       int H, int W, int U, int unit_off) {
    constexpr int CPU = 16 / sizeof(T);
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int y = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const int u = unit_off + c / CPU, e = c % CPU;
        dst[i] = (float)src[((((size_t)n * H + y) * U + u) * W + x) * CPU + e];
    }
}

// One thread per packed unit (u, tap, co): CPU input channels of one output channel and tap.
// Image 0 (units [0,U)) holds the rounded weights; with a split mode image 1 (units [U,2U)) holds
// the rounding residue w - round(w).  SR355_SPLIT_INPUT additionally repeats the rounded weights on
// channels [cin, 2cin) of image 0, where the input carries its own residue.
template <typename T>
__device__ __forceinline__ void pack_weight_body(const float *__restrict__ w, const float *__restrict__ bias,
                                                 T *__restrict__ wp, float *__restrict__ bp, int cout, int cin,
                                                 int U, int cout_pad, int split) {
    constexpr int CPU = 16 / sizeof(T);
    const int total = U * (split ? 2 : 1) * 9 * cout_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        // consecutive threads take consecutive (unit, tap) pairs of one output channel: together their eight reads
        // cover whole cache lines of w (the 16-byte stores are scattered instead)
        const int UU = U * (split ? 2 : 1);
        const int tap = i % 9;
        const int t = i / 9;
        const int uu = t % UU;
        const int co = t / UU;
        const int image = uu >= U ? 1 : 0;
        const int u = uu - image * U;
        T v[CPU];
#pragma unroll
        for (int e = 0; e < CPU; ++e) {
            int ci = u * CPU + e;
            if (split == SR355_SPLIT_INPUT && image == 0 && ci >= cin) ci -= cin;
            float f = 0.f;
            if (co < cout && ci < cin && (u * CPU + e) < (split == SR355_SPLIT_INPUT && image == 0 ? 2 * cin : cin)) {
                f = w[((size_t)co * cin + ci) * 9 + tap];
                if (image) f = f - (float)from_f32<T>(f);
            }
            v[e] = from_f32<T>(f);
        }
        uint4 pk;
        __builtin_memcpy(&pk, v, 16);
        reinterpret_cast<uint4 *>(wp)[((size_t)uu * 9 + tap) * cout_pad + co] = pk;
        if (i < cout_pad && bp) bp[i] = (i < cout && bias) ? bias[i] : 0.f;
    }
}

template <typename T>
__global__ void pack_weight_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                   T *__restrict__ wp, float *__restrict__ bp, int cout, int cin,
                                   int U, int cout_pad, int split) {
    pack_weight_body<T>(w, bias, wp, bp, cout, cin, U, cout_pad, split);
}

struct PackJobTable {
    PackJob j[PACK_BATCH];
};

// blockIdx.y = job; the blocks of a row stride over that job's items
template <typename T>
__global__ void pack_weight_batch_kernel(const PackJobTable t) {
    const PackJob &j = t.j[blockIdx.y];
    constexpr int CPU = 16 / sizeof(T);
    const int split = sizeof(T) == 2 ? j.split : 0;   // only the bf16 path carries rounding residues
    const int real = split == SR355_SPLIT_INPUT ? 2 * j.cin : j.cin;
    const int U = ((real + CPU - 1) / CPU + 1) / 2 * 2;   // whole stages of two units
    if constexpr (sizeof(T) == 2) {
        if (j.f16) {
            pack_weight_body<_Float16>(j.w, j.bias, (_Float16 *)j.wp, j.bp, j.cout, j.cin, U, (j.cout + 31) / 32 * 32, 0);
            return;
        }
    }
    pack_weight_body<T>(j.w, j.bias, (T *)j.wp, j.bp, j.cout, j.cin, U, (j.cout + 31) / 32 * 32, split);
}

// The same with the jobs in device memory (any number of them in ONE launch; blockIdx.y = job): a training step
// repacks every conv after the optimizer update, and 64 jobs per launch in the kernel arguments made that 6 launches.
template <typename T>
__global__ void pack_weight_table_kernel(const PackJob *__restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    constexpr int CPU = 16 / sizeof(T);
    const int split = sizeof(T) == 2 ? j.split : 0;
    const int real = split == SR355_SPLIT_INPUT ? 2 * j.cin : j.cin;
    const int U = ((real + CPU - 1) / CPU + 1) / 2 * 2;
    if constexpr (sizeof(T) == 2) {
        if (j.f16) {
            pack_weight_body<_Float16>(j.w, j.bias, (_Float16 *)j.wp, j.bp, j.cout, j.cin, U, (j.cout + 31) / 32 * 32, 0);
            return;
        }
    }
    pack_weight_body<T>(j.w, j.bias, (T *)j.wp, j.bp, j.cout, j.cin, U, (j.cout + 31) / 32 * 32, split);
}

static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_pack_weight_table(const PackJob *jobs_host, int n, int dtype, void *table_dev, hipStream_t stream) {
    if (n < 1) return SR355_OK;
    if (n > 65535) {
        set_error("pack_weight table: %d jobs exceed one launch", n);
        return SR355_ERR_UNSUPPORTED;
    }
    // (a pageable source: the runtime has staged it when the call returns, so the caller's vector may go)
    if (check_hip(hipMemcpyAsync(table_dev, jobs_host, (size_t)n * sizeof(PackJob), hipMemcpyHostToDevice, stream),
                  "hipMemcpyAsync"))
        return SR355_ERR_HIP;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_table_kernel<__bf16>, dim3(24, n), dim3(256), 0, stream, (const PackJob *)table_dev);
    else
        hipLaunchKernelGGL(pack_weight_table_kernel<float>, dim3(24, n), dim3(256), 0, stream, (const PackJob *)table_dev);
    return check_hip(hipGetLastError(), "pack_weight table launch");
}

int launch_nchw_to_blocked(const float *src, int N, int C, int H, int W, int dtype, void *dst,
                           int dst_units, int split, hipStream_t stream) {
    const size_t total = (size_t)N * H * dst_units * W;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL((nchw_to_blocked_kernel<__bf16, false>), dim3(grid_for(total)), dim3(256), 0, stream,
                           (const void *)src, (__bf16 *)dst, N, C, H, W, dst_units, split);
    else
        hipLaunchKernelGGL((nchw_to_blocked_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, stream,
                           (const void *)src, (float *)dst, N, C, H, W, dst_units, 0);
    return check_hip(hipGetLastError(), "nchw_to_blocked launch");
}

int launch_u8hwc_to_blocked(const unsigned char *src, int N, int C, int H, int W, int dtype, void *dst,
                            int dst_units, int split, hipStream_t stream) {
    const size_t total = (size_t)N * H * dst_units * W;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL((nchw_to_blocked_kernel<__bf16, true>), dim3(grid_for(total)), dim3(256), 0, stream,
                           (const void *)src, (__bf16 *)dst, N, C, H, W, dst_units, split);
    else
        hipLaunchKernelGGL((nchw_to_blocked_kernel<float, true>), dim3(grid_for(total)), dim3(256), 0, stream,
                           (const void *)src, (float *)dst, N, C, H, W, dst_units, 0);
    return check_hip(hipGetLastError(), "u8hwc_to_blocked launch");
}

int launch_blocked_to_nchw(const void *src, int src_units, int unit_off, int dtype, int N, int C, int H,
                           int W, float *dst, hipStream_t stream) {
    const size_t total = (size_t)N * C * H * W;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(blocked_to_nchw_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, stream,
                           (const __bf16 *)src, dst, N, C, H, W, src_units, unit_off);
    else
        hipLaunchKernelGGL(blocked_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                           (const float *)src, dst, N, C, H, W, src_units, unit_off);
    return check_hip(hipGetLastError(), "blocked_to_nchw launch");
}

int launch_pack_weight(const float *w, const float *bias, int cout, int cin, int dtype, int split,
                       void *wp, float *bp, hipStream_t stream) {
    if (dtype == SR355_DTYPE_BF16 && split == SR355_PACK_F16) {
        const int cu16 = real_units_for(cin, dtype, 0), cp16 = cout_pad_for(cout);
        hipLaunchKernelGGL(pack_weight_kernel<_Float16>, dim3(grid_for((size_t)cu16 * 9 * cp16)), dim3(256), 0, stream,
                           w, bias, (_Float16 *)wp, bp, cout, cin, cu16, cp16, 0);
        return check_hip(hipGetLastError(), "pack_weight launch");
    }
    const int cu = real_units_for(cin, dtype, split), cp = cout_pad_for(cout);
    const size_t total = (size_t)packed_units_for(cin, dtype, split) * 9 * cp;
    if (dtype == SR355_DTYPE_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, stream, w, bias,
                           (__bf16 *)wp, bp, cout, cin, cu, cp, split);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, w, bias,
                           (float *)wp, bp, cout, cin, cu, cp, 0);
    return check_hip(hipGetLastError(), "pack_weight launch");
}

// Adam over flat fp32 buffers (torch.optim.Adam without amsgrad / maximize, the optimizer of
// /root/reference/esrgan/train.py:117-122), four elements per thread; `step_size` = lr / (1 - beta1^t) and
// `bc2_sqrt` = sqrt(1 - beta2^t) come from the host, as do omb1 = 1 - beta1 and omb2 = 1 - beta2 (taken in double,
// as torch takes them, before rounding to fp32).
__global__ void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                 float *__restrict__ v, size_t n, float beta1, float beta2, float omb1,
                                 float omb2, float eps, float weight_decay, float step_size, float bc2_sqrt) {
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto one = [&](float &pp, float gg, float &mm, float &vv) {
        if (weight_decay != 0.f) gg += weight_decay * pp;
        mm = mm + (gg - mm) * omb1;
        vv = vv * beta2 + omb2 * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * (mm / denom);
    };
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i],
               vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        one(pp.x, gg.x, mm.x, vv.x);
        one(pp.y, gg.y, mm.y, vv.y);
        one(pp.z, gg.z, mm.z, vv.z);
        one(pp.w, gg.w, mm.w, vv.w);
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (size_t i = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
        one(p[i], g[i], m[i], v[i]);
}

int launch_adam_flat(float *p, const float *g, float *m, float *v, size_t n, float beta1, float beta2, float omb1,
                     float omb2, float eps, float weight_decay, float step_size, float bc2_sqrt, hipStream_t stream) {
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, p, g, m, v, n, beta1,
                       beta2, omb1, omb2, eps, weight_decay, step_size, bc2_sqrt);
    return check_hip(hipGetLastError(), "adam_flat launch");
}

int launch_pack_weight_batch(const PackJob *jobs, int n, int dtype, hipStream_t stream) {
    for (int at = 0; at < n; at += PACK_BATCH) {
        PackJobTable t;
        const int m = n - at < PACK_BATCH ? n - at : PACK_BATCH;
        size_t biggest = 0;      // items of the largest job: the blocks of a row stride over their job's items
        for (int i = 0; i < m; ++i) {
            t.j[i] = jobs[at + i];
            const size_t items = (size_t)((jobs[at + i].cin + 7) / 8 + 1) * 9 * ((jobs[at + i].cout + 31) / 32 * 32);
            if (items > biggest) biggest = items;
        }
        // 48 blocks a job serve the generator's convs (at most 192 x 64); a 512 x 512 conv wants ten times as many
        unsigned gx = (unsigned)((biggest + 1023) / 1024);
        gx = gx < 48 ? 48 : gx > 512 ? 512 : gx;
        if (dtype == SR355_DTYPE_BF16)
            hipLaunchKernelGGL(pack_weight_batch_kernel<__bf16>, dim3(gx, m), dim3(256), 0, stream, t);
        else
            hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(gx, m), dim3(256), 0, stream, t);
        if (check_hip(hipGetLastError(), "pack_weight batch launch")) return SR355_ERR_HIP;
    }
    return SR355_OK;
}

}  // namespace sr355
