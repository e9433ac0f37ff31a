// Weight / bias gradient of the 3x3 convolution on the exact-fp32 path (autograd of nn.Conv2d in
// /root/reference/esrgan/models/esrgan_model.py:10-14 under loss.backward(), /root/reference/esrgan/train.py:161,
// when the module is built with precision="fp32"):
//   dW[co][ci][tap] = sum_pixels dY[co][p] * X[ci][p + tap]
// on v_mfma_f32_32x32x2f32: M = 32 output channels, N = 32 input channels, K = 2 pixels per instruction, one
// accumulator tile per tap.  A lane supplies one float of each operand (output channel lane & 31 of pixel x + (lane >> 5)
// for dY, input channel lane & 31 of the tap-shifted pixel for X), read straight from the blocked fp32 tensors
// [N][H][units][W][4]: at 64 cycles per MFMA the ten dword loads per pixel pair are not what bounds the kernel, and
// the tap-shifted reads of a row hit the vector cache.  One wave per workgroup; a workgroup owns (a block of 32 output
// channels, a chunk of 32 input channels, every G-th image row) and writes one partial slab in the layout of the bf16
// kernel, so the same fixed-order reduction (launch_wgrad_reduce) produces dW / db: deterministic, no float atomics.
#include <string.h>

#include "sr355_internal.h"

namespace sr355 {

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WgradF32Args {
    const float *in;      // blocked fp32 [N][H >> ups][in_units][W >> ups][4]
    const float *dy;      // blocked fp32 [N][H][dy_units][W][4]
    float *slabs;         // [chunks][G][9][CO][32]
    float *bias_slabs;    // [G][CO], or null
    int N, H, W;
    int in_units, ups;
    int dy_units, dy_unit_off;
    int CO, G;
};

constexpr int PAIRS = 4;   // pixel pairs whose operands are loaded before their MFMAs are issued

__global__ __launch_bounds__(64) void wgrad3x3_f32_kernel(const WgradF32Args a) {
    const int lane = threadIdx.x;
    const int m = lane & 31, k = lane >> 5;
    const int g = blockIdx.x, chunk = blockIdx.y, cob = blockIdx.z;
    const int H = a.H, W = a.W, ups = a.ups;
    const int Hin = H >> ups, Win = W >> ups;
    const int co = cob * 32 + m, ci = chunk * 32 + m;
    const int dy_unit = a.dy_unit_off + (co >> 2), in_unit = ci >> 2;
    const bool co_ok = dy_unit < a.dy_units, ci_ok = in_unit < a.in_units;   // channels beyond the real ones but inside
                                                                             // the buffers are summed and never stored
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    const int rows = a.N * H;
    for (int r = g; r < rows; r += a.G) {
        const int n = r / H, y = r - n * H;
        const float *dy_row = a.dy + (((size_t)n * H + y) * a.dy_units + dy_unit) * W * 4 + (co & 3);
        const float *in_row[3];
        bool row_ok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            row_ok[ky] = ci_ok && yy >= 0 && yy < H;
            in_row[ky] = a.in + (((size_t)n * Hin + ((row_ok[ky] ? yy : 0) >> ups)) * a.in_units + (ci_ok ? in_unit : 0)) * Win * 4 +
                         (ci & 3);
        }
        for (int x0 = 0; x0 < W; x0 += 2 * PAIRS) {
            float av[PAIRS], bv[PAIRS][9];
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                const int x = x0 + 2 * p + k;
                av[p] = (co_ok && x < W) ? dy_row[(size_t)x * 4] : 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        // (a pixel beyond the row end contributes nothing: its dY operand is zero)
                        bv[p][ky * 3 + kx] = (row_ok[ky] && xx >= 0 && xx < W) ? in_row[ky][(size_t)(xx >> ups) * 4] : 0.f;
                    }
            }
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                bsum += av[p];
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv[p][t], acc[t], 0, 0, 0);
            }
        }
    }

    // slab [tap][co][ci32]; accumulator element e of a lane: row (co) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5),
    // column (ci) = lane & 31
    float *slab = a.slabs + ((size_t)chunk * a.G + g) * 9 * a.CO * 32 + (size_t)cob * 32 * 32;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rowc = (e & 3) + 8 * (e >> 2) + 4 * k;
            slab[((size_t)t * a.CO + rowc) * 32 + m] = acc[t][e];
        }
    if (a.bias_slabs && chunk == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (k == 0) a.bias_slabs[(size_t)g * a.CO + co] = bsum;
    }
}

void wgrad_f32_geometry(int cout, int cin, int N, int H, int *chunks, int *G, int *CO) {
    *chunks = (cin + 31) / 32;
    *CO = cout_pad_for(cout);
    const long long rows = (long long)N * H;
    long long g = 1024 / ((long long)*chunks * (*CO / 32));   // about one wave per SIMD of the chip
    if (g > rows) g = rows;
    if (g < 1) g = 1;
    *G = (int)g;
}
}  // namespace

size_t wgrad_f32_workspace_bytes(int cout, int cin, int N, int H, int W) {
    (void)W;
    int chunks, G, CO;
    wgrad_f32_geometry(cout, cin, N, H, &chunks, &G, &CO);
    return ((size_t)chunks * G * 9 * 32 + G) * CO * sizeof(float);
}

int launch_wgrad3x3_f32(const float *in, int in_units, int ups, const float *dy, int dy_units, int dy_unit_off, int N,
                        int H, int W, int cout, int cin, float scale, int accumulate, float *dw, float *db, float *slabs,
                        hipStream_t stream) {
    int chunks, G, CO;
    wgrad_f32_geometry(cout, cin, N, H, &chunks, &G, &CO);
    WgradF32Args a;
    memset(&a, 0, sizeof(a));
    a.in = in, a.dy = dy, a.slabs = slabs;
    a.bias_slabs = db ? slabs + (size_t)chunks * G * 9 * CO * 32 : nullptr;
    a.N = N, a.H = H, a.W = W;
    a.in_units = in_units, a.ups = ups;
    a.dy_units = dy_units, a.dy_unit_off = dy_unit_off;
    a.CO = CO, a.G = G;
    hipLaunchKernelGGL(wgrad3x3_f32_kernel, dim3(G, chunks, CO / 32), dim3(64), 0, stream, a);
    if (check_hip(hipGetLastError(), "wgrad3x3 f32 launch")) return SR355_ERR_HIP;
    return launch_wgrad_reduce(slabs, dw, cout, cin, CO, G, scale, accumulate, a.bias_slabs, db, stream);
}

}  // namespace sr355
