// ToImage ("ToRGB") of the StyleGAN2 generator as ONE streaming pass each way (ABI v16).
//
// Reference: implementations/StyleGAN2/model.py:239-250 -- a 1x1 ModulatedConv2d WITHOUT demodulation (model.py:91-135) from C feature
// channels to IC <= 4 image channels, plus the skip sum with the previous level's image:
//     out[n,co,p] = coef * sum_c w[co,c] * (s_raw[n,c] + 1) * x[n,c,p] + bias[co] + pre[n,co,p]
// x is channels-last bf16 (the conv layout), the image is planar NCHW bf16 (the FIR layout of the x2 upsample that follows).
//
// With 3 output channels this is 3*C FMAs per pixel against 2*C bytes read: HBM-bound on the packed-fp32 VALU, no MFMA (padding the 3
// outputs to an 8-wide MFMA tile needs a zero-padded weight copy, a padded bias, a crop of the result and, in backward, a zero-padded
// gradient tensor + a second read of x for the weight gradient: ~11 launches and 5 passes over the feature map per level).  Here:
//   forward : one read of x, one planar write (3/C of the input);
//   backward: one read of x and dy, one write of dx = t * s with t[c] = coef * sum_co dy[co] w[co,c]; the per-image moments
//             Q_n[co,c] = sum_p dy[n,co,p] x[n,c,p] are reduced in registers -> LDS -> one plain store per block (no atomics), and a
//             finish kernel turns them into  ds_raw[n,c] = coef * sum_co w[co,c] Q_n[co,c],  dw[co,c] = coef * sum_n s[n,c] Q_n[co,c]
//             and db[co] = sum dy  (deterministic: fixed summation order).
// A lane owns 8 consecutive channels (one 16-byte vector) of a pixel; C/8 lanes (a power of two <= 64) share a pixel and reduce with
// wavefront shuffles.  s_raw is read through a row stride (a column block of the batched style GEMM, Synthesis._batched_affines).
#include "agf_common.h"

namespace {

constexpr int TB = 256;
constexpr int MAXC = 512;

struct ToRgbParams {
    const uint16_t* x; const float* w; const float* bias; const float* s_raw; int64_t s_stride;
    const uint16_t* pre; uint16_t* out;           // forward
    const uint16_t* dy; uint16_t* dx; float* part; // backward: part [N][chunks][IC + 1][C]  (row IC: per-lane-group sums of dy, c < IC used)
    int N, HW, C, chunks, iters;                   // iters = pixel groups per block
    float coef;
};

template <int LPP>
static __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = LPP / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace

template <int IC, int LPP>
__global__ void __launch_bounds__(TB) torgb_fwd_kernel(ToRgbParams p) {
    constexpr int PPI = TB / LPP;                       // pixels per block iteration
    const int tid = threadIdx.x, g = tid % LPP, sub = tid / LPP;
    const int n = blockIdx.y;
    const int c0 = g * 8;
    float ws[IC][8];
    {
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = (p.s_raw[(int64_t)n * p.s_stride + c0 + j] + 1.0f) * p.coef;
#pragma unroll
        for (int co = 0; co < IC; co++)
#pragma unroll
            for (int j = 0; j < 8; j++) ws[co][j] = p.w[co * p.C + c0 + j] * s[j];
    }
    float b[IC];
#pragma unroll
    for (int co = 0; co < IC; co++) b[co] = p.bias ? p.bias[co] : 0.0f;
    const uint16_t* xb = p.x + (int64_t)n * p.HW * p.C;
    const int64_t plane = p.HW;
    const int px0 = blockIdx.x * p.iters * PPI;
    for (int it = 0; it < p.iters; it++) {
        const int px = px0 + it * PPI + sub;
        float acc[IC];
#pragma unroll
        for (int co = 0; co < IC; co++) acc[co] = 0.0f;
        if (px < p.HW) {
            float v[8];
            VecIO<bf16_t, 8>::load((const bf16_t*)(xb + (int64_t)px * p.C + c0), v);
#pragma unroll
            for (int co = 0; co < IC; co++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[co] = fmaf(v[j], ws[co][j], acc[co]);
        }
#pragma unroll
        for (int co = 0; co < IC; co++) acc[co] = group_sum<LPP>(acc[co]);
        if (g == 0 && px < p.HW) {
#pragma unroll
            for (int co = 0; co < IC; co++) {
                const int64_t o = ((int64_t)n * IC + co) * plane + px;
                float r = acc[co] + b[co];
                if (p.pre) r += bf16_bits_to_f32(p.pre[o]);
                p.out[o] = (uint16_t)f32_to_bf16_bits(r);
            }
        }
    }
}

template <int IC, int LPP>
__global__ void __launch_bounds__(TB) torgb_bwd_kernel(ToRgbParams p) {
    constexpr int PPI = TB / LPP;
    __shared__ float red[PPI][(IC + 1) * 8 * LPP + 1];   // [pixel sub-group][(IC + 1) x C], +1 word: the row pitch is odd in banks
    const int tid = threadIdx.x, g = tid % LPP, sub = tid / LPP;
    const int n = blockIdx.y;
    const int c0 = g * 8;
    float wt[IC][8], s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = p.s_raw[(int64_t)n * p.s_stride + c0 + j] + 1.0f;
#pragma unroll
    for (int co = 0; co < IC; co++)
#pragma unroll
        for (int j = 0; j < 8; j++) wt[co][j] = p.w[co * p.C + c0 + j] * p.coef * s[j];      // dx = sum_co dy[co] * (coef w s)
    float q[IC][8], bs[IC];
#pragma unroll
    for (int co = 0; co < IC; co++) {
        bs[co] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) q[co][j] = 0.0f;
    }
    const uint16_t* xb = p.x + (int64_t)n * p.HW * p.C;
    uint16_t* dxb = p.dx + (int64_t)n * p.HW * p.C;
    const uint16_t* dyb = p.dy + (int64_t)n * IC * p.HW;
    const int px0 = blockIdx.x * p.iters * PPI;
    for (int it = 0; it < p.iters; it++) {
        const int px = px0 + it * PPI + sub;
        if (px >= p.HW) break;                                   // (uniform per pixel group; no shuffles in this loop)
        float v[8], d[IC];
        VecIO<bf16_t, 8>::load((const bf16_t*)(xb + (int64_t)px * p.C + c0), v);
#pragma unroll
        for (int co = 0; co < IC; co++) d[co] = bf16_bits_to_f32(dyb[(int64_t)co * p.HW + px]);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float t = 0.0f;
#pragma unroll
            for (int co = 0; co < IC; co++) t = fmaf(d[co], wt[co][j], t);
            o[j] = t;
        }
        VecIO<bf16_t, 8>::store((bf16_t*)(dxb + (int64_t)px * p.C + c0), o);
#pragma unroll
        for (int co = 0; co < IC; co++) {
            bs[co] += d[co];
#pragma unroll
            for (int j = 0; j < 8; j++) q[co][j] = fmaf(d[co], v[j], q[co][j]);
        }
    }
    // block reduction over the PPI pixel sub-groups (fixed order), one plain store per (co, c) and block
    const int CC = 8 * LPP;
#pragma unroll
    for (int co = 0; co < IC; co++)
#pragma unroll
        for (int j = 0; j < 8; j++) red[sub][co * CC + c0 + j] = q[co][j];
    if (g == 0) {
#pragma unroll
        for (int co = 0; co < IC; co++) red[sub][IC * CC + co] = bs[co];
    }
    __syncthreads();
    float* dst = p.part + ((int64_t)n * p.chunks + blockIdx.x) * (IC + 1) * CC;
    for (int i = tid; i < IC * CC + IC; i += TB) {
        float a = 0.0f;
#pragma unroll 4
        for (int r = 0; r < PPI; r++) a += red[r][i];
        dst[i] = a;
    }
}

// finish: thread (c, nslice); block = 64 channels x 16 n-slices
template <int IC>
__global__ void __launch_bounds__(1024) torgb_finish_kernel(const float* __restrict__ part, const float* __restrict__ w, const float* __restrict__ s_raw,
                                                           int64_t s_stride, float* __restrict__ ds, float* __restrict__ dw, float* __restrict__ db,
                                                           int N, int C, int chunks, float coef) {
    __shared__ float red[16][IC][64];
    const int cl = threadIdx.x % 64, ns = threadIdx.x / 64;
    const int c = blockIdx.x * 64 + cl;
    float acc[IC];
#pragma unroll
    for (int co = 0; co < IC; co++) acc[co] = 0.0f;
    if (c < C) {
        float wc[IC];
#pragma unroll
        for (int co = 0; co < IC; co++) wc[co] = w[co * C + c] * coef;
        for (int n = ns; n < N; n += 16) {
            float Q[IC], v[8][IC];
            const float* src = part + (int64_t)n * chunks * (IC + 1) * C + c;
#pragma unroll
            for (int k = 0; k < 8; k++)                    // (chunks <= 8: all loads of an image in flight together)
#pragma unroll
                for (int co = 0; co < IC; co++) v[k][co] = k < chunks ? src[((int64_t)k * (IC + 1) + co) * C] : 0.0f;
#pragma unroll
            for (int co = 0; co < IC; co++) {
                Q[co] = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; k++) Q[co] += v[k][co];
            }
            float d = 0.0f;
            const float sc = s_raw[(int64_t)n * s_stride + c] + 1.0f;
#pragma unroll
            for (int co = 0; co < IC; co++) { d = fmaf(wc[co], Q[co], d); acc[co] = fmaf(sc, Q[co], acc[co]); }
            if (ds) ds[(int64_t)n * C + c] = d;
        }
    }
#pragma unroll
    for (int co = 0; co < IC; co++) red[ns][co][cl] = acc[co];
    __syncthreads();
    if (ns == 0 && c < C) {
#pragma unroll
        for (int co = 0; co < IC; co++) {
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) a += red[r][co][cl];
            if (dw) dw[co * C + c] = a * coef;
        }
    }
    // bias gradient: block 0, the same (column, n-slice) scheme on row IC of the partials (columns 0..IC-1 hold the sums of dy)
    if (db && blockIdx.x == 0) {
        __syncthreads();
        float a = 0.0f;
        if (cl < IC)
            for (int n = ns; n < N; n += 16)
                for (int k = 0; k < chunks; k++) a += part[((int64_t)n * chunks + k) * (IC + 1) * C + IC * C + cl];
        red[ns][0][cl] = a;
        __syncthreads();
        if (ns == 0 && cl < IC) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) t += red[r][0][cl];
            db[cl] = t;
        }
    }
}

namespace {

static int pick_chunks(int HW, int ppi, int N) {
    // enough blocks to fill 256 CUs a few times over, at least 4 pixel groups per block
    int groups = (HW + ppi - 1) / ppi;
    int chunks = groups / 4;
    if (chunks < 1) chunks = 1;
    int want = (2048 + N - 1) / N;
    if (chunks > want) chunks = want;
    if (chunks > 64) chunks = 64;
    return chunks;
}

template <int IC, int LPP>
static void launch_fwd(ToRgbParams& p, hipStream_t st) {
    const int ppi = TB / LPP;
    p.chunks = pick_chunks(p.HW, ppi, p.N);
    const int groups = (p.HW + ppi - 1) / ppi;
    p.iters = (groups + p.chunks - 1) / p.chunks;
    p.chunks = (groups + p.iters - 1) / p.iters;
    hipLaunchKernelGGL((torgb_fwd_kernel<IC, LPP>), dim3(p.chunks, p.N), dim3(TB), 0, st, p);
}
template <int IC, int LPP>
static void launch_bwd(ToRgbParams& p, hipStream_t st) {
    hipLaunchKernelGGL((torgb_bwd_kernel<IC, LPP>), dim3(p.chunks, p.N), dim3(TB), 0, st, p);
}

static void bwd_geometry(ToRgbParams& p) {
    const int lpp = p.C / 8, ppi = TB / lpp;
    p.chunks = pick_chunks(p.HW, ppi, p.N);
    if (p.chunks > 8) p.chunks = 8;                     // the finish kernel loads the chunks of an image as one batch of 8
    const int groups = (p.HW + ppi - 1) / ppi;
    p.iters = (groups + p.chunks - 1) / p.chunks;
    p.chunks = (groups + p.iters - 1) / p.iters;
}

#define TORGB_DISPATCH(FN, IC_, p, st)                                  \
    switch (p.C / 8) {                                                   \
        case 1: FN<IC_, 1>(p, st); break;                                \
        case 2: FN<IC_, 2>(p, st); break;                                \
        case 4: FN<IC_, 4>(p, st); break;                                \
        case 8: FN<IC_, 8>(p, st); break;                                \
        case 16: FN<IC_, 16>(p, st); break;                              \
        case 32: FN<IC_, 32>(p, st); break;                              \
        default: FN<IC_, 64>(p, st); break;                              \
    }

static int check_common(const void* x, const float* w, const float* s_raw, int dtype, int N, int H, int W, int C, int IC) {
    AGF_CHECK(x && w && s_raw, "torgb: null pointer");
    AGF_CHECK(dtype == AGF_BF16, "torgb: activations must be bfloat16 (the fp32 reference-precision path composes the separate operators)");
    AGF_CHECK(N >= 1 && N < 65536 && H >= 1 && W >= 1 && (int64_t)H * W < (1ll << 30), "torgb: bad shape");
    AGF_CHECK(C >= 8 && C <= MAXC && (C & (C - 1)) == 0, "torgb: the channel count must be a power of two in 8..512");
    AGF_CHECK(IC >= 1 && IC <= 4, "torgb: 1..4 image channels");
    AGF_CHECK(((uintptr_t)x % 16) == 0, "torgb: x must be 16-byte aligned");
    return AGF_OK;
}

}  // namespace

extern "C" int agf_torgb_covers(int32_t C, int32_t IC) { return C >= 8 && C <= MAXC && (C & (C - 1)) == 0 && IC >= 1 && IC <= 4; }

extern "C" int agf_torgb_fwd(const void* x, const float* w, const float* bias, const float* s_raw, int64_t s_stride, const void* pre,
                             void* out, int dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t IC, float coef, void* stream) {
    int rc = check_common(x, w, s_raw, dtype, N, H, W, C, IC);
    if (rc != AGF_OK) return rc;
    AGF_CHECK(out, "torgb_fwd: null output");
    ToRgbParams p = {};
    p.x = (const uint16_t*)x; p.w = w; p.bias = bias; p.s_raw = s_raw; p.s_stride = s_stride; p.pre = (const uint16_t*)pre; p.out = (uint16_t*)out;
    p.N = N; p.HW = H * W; p.C = C; p.coef = coef;
    hipStream_t st = (hipStream_t)stream;
    switch (IC) {
        case 1: TORGB_DISPATCH(launch_fwd, 1, p, st); break;
        case 2: TORGB_DISPATCH(launch_fwd, 2, p, st); break;
        case 3: TORGB_DISPATCH(launch_fwd, 3, p, st); break;
        default: TORGB_DISPATCH(launch_fwd, 4, p, st); break;
    }
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

extern "C" int64_t agf_torgb_bwd_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t C, int32_t IC) {
    if (!agf_torgb_covers(C, IC) || N < 1 || H < 1 || W < 1) return 0;
    ToRgbParams p = {};
    p.N = N; p.HW = H * W; p.C = C;
    bwd_geometry(p);
    return (int64_t)N * p.chunks * (IC + 1) * C;
}

extern "C" int agf_torgb_bwd(const void* dy, const void* x, const float* w, const float* s_raw, int64_t s_stride, void* dx, float* ds, float* dw,
                             float* db, float* workspace, int64_t workspace_floats, int dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                             int32_t IC, float coef, void* stream) {
    int rc = check_common(x, w, s_raw, dtype, N, H, W, C, IC);
    if (rc != AGF_OK) return rc;
    AGF_CHECK(dy && dx && workspace, "torgb_bwd: null pointer");
    AGF_CHECK(((uintptr_t)dx % 16) == 0, "torgb_bwd: dx must be 16-byte aligned");
    ToRgbParams p = {};
    p.x = (const uint16_t*)x; p.w = w; p.s_raw = s_raw; p.s_stride = s_stride; p.dy = (const uint16_t*)dy; p.dx = (uint16_t*)dx; p.part = workspace;
    p.N = N; p.HW = H * W; p.C = C; p.coef = coef;
    bwd_geometry(p);
    AGF_CHECK(workspace_floats >= (int64_t)N * p.chunks * (IC + 1) * C, "torgb_bwd: workspace too small (agf_torgb_bwd_workspace_floats)");
    hipStream_t st = (hipStream_t)stream;
    switch (IC) {
        case 1: TORGB_DISPATCH(launch_bwd, 1, p, st); break;
        case 2: TORGB_DISPATCH(launch_bwd, 2, p, st); break;
        case 3: TORGB_DISPATCH(launch_bwd, 3, p, st); break;
        default: TORGB_DISPATCH(launch_bwd, 4, p, st); break;
    }
    AGF_LAUNCH_CHECK();
    const dim3 fg((C + 63) / 64);
    switch (IC) {
        case 1: hipLaunchKernelGGL((torgb_finish_kernel<1>), fg, dim3(1024), 0, st, workspace, w, s_raw, s_stride, ds, dw, db, N, C, p.chunks, coef); break;
        case 2: hipLaunchKernelGGL((torgb_finish_kernel<2>), fg, dim3(1024), 0, st, workspace, w, s_raw, s_stride, ds, dw, db, N, C, p.chunks, coef); break;
        case 3: hipLaunchKernelGGL((torgb_finish_kernel<3>), fg, dim3(1024), 0, st, workspace, w, s_raw, s_stride, ds, dw, db, N, C, p.chunks, coef); break;
        default: hipLaunchKernelGGL((torgb_finish_kernel<4>), fg, dim3(1024), 0, st, workspace, w, s_raw, s_stride, ds, dw, db, N, C, p.chunks, coef); break;
    }
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}
