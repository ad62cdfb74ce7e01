// One layer of the StyleGAN2 mapping network as one launch forward and two backward (ABI v18).
//
// Reference: implementations/StyleGAN2/model.py:71-78 (MapLinear: ``(x * coef @ W^T + b) * lr``) followed by nn.LeakyReLU(0.2) (:263-282):
//     y[b, o] = lrelu( alpha * sum_k x[b, k] W[o, k] + beta * bias[o] ),      alpha = coef * lr,  beta = lr,      fp32 throughout
// x [B, Din], W [Dout, Din], B = 64..128 rows, Din = Dout = 512: 34-67 MFLOP per layer -- nothing for the chip, and exactly why the
// layer was launch-bound: the library GEMM needs a broadcast copy of the bias, the GEMM and an activation kernel forward (3 launches, 21 us)
// and seven launches backward (45 us), 8 layers, two generator passes per iteration.  A column-split VALU kernel does a layer in one
// launch: a block owns 8 output columns (its slice of W, 16 KB, staged in LDS once) for 64 rows; x is streamed through LDS in chunks of
// 32 inputs, transposed so that the 64 row-lanes of a wave read consecutive banks while the weight is a broadcast.
//   backward:  g = dy * lrelu'(y);   dx = alpha * g @ W  (same kernel shape, W read by columns);
//              dW = alpha * g^T @ x  and  db = beta * sum_b g  (32 x 32 tiles of dW, reduction over the batch rows in LDS).
#include "agf_common.h"

namespace {
constexpr int MAP_OC = 8;       // output columns per block
constexpr int MAP_KC = 32;      // reduction chunk staged in LDS
constexpr int MAP_MAXD = 1024;
}

// MODE 0: forward (y = lrelu(alpha x W^T + beta b));  MODE 1: data gradient (dx = alpha (dy * lrelu'(yref)) W)
template <int MODE>
__global__ void __launch_bounds__(256) map_layer_kernel(const float* __restrict__ x, const float* __restrict__ yref, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out, int B, int Din, int Dout,
                                                        float alpha, float beta, float slope) {
    // forward: reduction length R = Din, outputs O = Dout; backward: R = Dout, O = Din
    const int R = MODE == 0 ? Din : Dout, O = MODE == 0 ? Dout : Din;
    extern __shared__ float smem[];
    float* ws = smem;                                   // [MAP_OC][R]
    float* xs = smem + MAP_OC * R;                      // [MAP_KC][65]
    const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;
    const int o0 = blockIdx.x * MAP_OC, r0 = blockIdx.y * 64;
    // this block's slice of the weights: ws[c][k] = W[o0 + c][k] (forward) or W[k][o0 + c] (backward)
    for (int i = tid; i < MAP_OC * R; i += 256) {
        if (MODE == 0) { const int c = i / R, k = i - c * R; ws[c * R + k] = (o0 + c < O) ? W[(int64_t)(o0 + c) * Din + k] : 0.f; }
        else { const int k = i / MAP_OC, c = i - k * MAP_OC; ws[c * R + k] = (o0 + c < O) ? W[(int64_t)k * Din + o0 + c] : 0.f; }
    }
    float acc0 = 0.f, acc1 = 0.f;
    for (int k0 = 0; k0 < R; k0 += MAP_KC) {
        __syncthreads();
        for (int i = tid; i < 64 * MAP_KC; i += 256) {
            const int row = i / MAP_KC, kk = i - row * MAP_KC;
            float v = 0.f;
            if (r0 + row < B && k0 + kk < R) {
                const int64_t idx = (int64_t)(r0 + row) * R + k0 + kk;
                v = x[idx];
                if (MODE == 1) v = yref[idx] > 0.f ? v : v * slope;         // g = dy * lrelu'(y)
            }
            xs[kk * 65 + row] = v;
        }
        __syncthreads();
        const float* w0 = ws + (q * 2) * R + k0;
        const float* w1 = w0 + R;
#pragma unroll 8
        for (int kk = 0; kk < MAP_KC; kk++) {
            const float xv = xs[kk * 65 + r];
            acc0 = fmaf(xv, w0[kk], acc0);
            acc1 = fmaf(xv, w1[kk], acc1);
        }
    }
    if (r0 + r < B) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int o = o0 + q * 2 + c;
            if (o >= O) continue;
            float v = (c ? acc1 : acc0) * alpha;
            if (MODE == 0) {
                v += beta * (bias ? bias[o] : 0.f);
                v = v > 0.f ? v : v * slope;
            }
            out[(int64_t)(r0 + r) * O + o] = v;
        }
    }
}

// dW[o, k] = alpha * sum_b g[b, o] x[b, k],  db[o] = beta * sum_b g[b, o];  block = a 32 x 32 tile of dW, all batch rows through LDS
__global__ void __launch_bounds__(256) map_layer_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yref, const float* __restrict__ x,
                                                              float* __restrict__ dW, float* __restrict__ db, int B, int Din, int Dout,
                                                              float alpha, float beta, float slope) {
    __shared__ float gs[64][33], xs[64][33];
    const int tid = threadIdx.x;
    const int o0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int ol = tid >> 3, kq = (tid & 7) * 4;         // this thread: row ol of the tile, columns kq .. kq+3
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * 32; i += 256) {
            const int row = i >> 5, c = i & 31;
            float g = 0.f, xv = 0.f;
            if (b0 + row < B) {
                if (o0 + c < Dout) { const int64_t idx = (int64_t)(b0 + row) * Dout + o0 + c; g = dy[idx]; g = yref[idx] > 0.f ? g : g * slope; }
                if (k0 + c < Din) xv = x[(int64_t)(b0 + row) * Din + k0 + c];
            }
            gs[row][c] = g; xs[row][c] = xv;
        }
        __syncthreads();
#pragma unroll 8
        for (int b = 0; b < 64; b++) {
            const float g = gs[b][ol];
            bsum += g;
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = fmaf(g, xs[b][kq + j], acc[j]);
        }
    }
    if (o0 + ol < Dout) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (k0 + kq + j < Din) dW[(int64_t)(o0 + ol) * Din + k0 + kq + j] = acc[j] * alpha;
        if (db && blockIdx.x == 0 && (tid & 7) == 0) db[o0 + ol] = bsum * beta;
    }
}

static int map_check(const void* a, const void* b, const void* c, int B, int Din, int Dout) {
    AGF_CHECK(a && b && c, "map_layer: null pointer");
    AGF_CHECK(B >= 1 && B <= 65535 * 64 && Din >= 1 && Dout >= 1 && Din <= MAP_MAXD && Dout <= MAP_MAXD, "map_layer: bad shape (dims up to 1024)");
    return AGF_OK;
}

extern "C" int agf_map_layer_fwd(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t Din, int32_t Dout,
                                 float alpha, float beta, float slope, void* stream) {
    int rc = map_check(x, W, y, B, Din, Dout);
    if (rc != AGF_OK) return rc;
    const size_t lds = (size_t)(MAP_OC * Din + MAP_KC * 65) * sizeof(float);
    hipLaunchKernelGGL((map_layer_kernel<0>), dim3((Dout + MAP_OC - 1) / MAP_OC, (B + 63) / 64), dim3(256), lds, (hipStream_t)stream,
                       x, (const float*)nullptr, W, bias, y, B, Din, Dout, alpha, beta, slope);
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

extern "C" int agf_map_layer_bwd(const float* dy, const float* y, const float* x, const float* W, float* dx, float* dW, float* db,
                                 int32_t B, int32_t Din, int32_t Dout, float alpha, float beta, float slope, void* stream) {
    int rc = map_check(dy, y, W, B, Din, Dout);
    if (rc != AGF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const size_t lds = (size_t)(MAP_OC * Dout + MAP_KC * 65) * sizeof(float);
        hipLaunchKernelGGL((map_layer_kernel<1>), dim3((Din + MAP_OC - 1) / MAP_OC, (B + 63) / 64), dim3(256), lds, st,
                           dy, y, W, (const float*)nullptr, dx, B, Din, Dout, alpha, beta, slope);
        AGF_LAUNCH_CHECK();
    }
    if (dW) {
        AGF_CHECK(x, "map_layer_bwd: the weight gradient needs x");
        hipLaunchKernelGGL(map_layer_wgrad_kernel, dim3((Din + 31) / 32, (Dout + 31) / 32), dim3(256), 0, st,
                           dy, y, x, dW, db, B, Din, Dout, alpha, beta, slope);
        AGF_LAUNCH_CHECK();
    }
    return AGF_OK;
}
