// Fused bias + activation + gain + clamp and its first / second derivatives, gfx950.
//
// Semantics: reference bias_act.cu:17-141 (order: +b -> act -> *gain*dy -> clamp; grad 1/2 use the saved
// yref/gain or xref; clamp grad zeroes where |yref| >= clamp).  Pure streaming op: HBM-bound, so the only
// design points are 16-byte accesses per lane, a grid-stride loop, and computing the bias index without
// a division in the common channels-last case (step_b == 1).
#include "agf_common.h"

struct BiasActParams {
    const void *x, *b, *xref, *yref, *dy;
    void* y;
    int64_t sizeX;
    int sizeB;
    int64_t stepB;
    int grad;
    float alpha, gain, clamp;
};

template <class S, int A>
static __device__ __forceinline__ S act_eval(S x, S xref, S yref, S dy, int G, S alpha, S gain, S clamp) {
    const S one = 1, two = 2, expRange = 80, halfExpRange = 40;
    const S seluScale = (S)1.0507009873554804934193349852946, seluAlpha = (S)1.6732632423543772848170429916717;
    S yy = (gain != 0) ? yref / gain : 0;
    S y = 0;
    if (A == 1) { if (G <= 1) y = x; }
    if (A == 2) { if (G == 0) y = (x > 0) ? x : 0; if (G == 1) y = (yy > 0) ? x : 0; }
    if (A == 3) { if (G == 0) y = (x > 0) ? x : x * alpha; if (G == 1) y = (yy > 0) ? x : x * alpha; }
    if (A == 4) {
        if (G == 0) { S c = exp(x); S d = one / c; y = (x < -expRange) ? -one : (x > expRange) ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    }
    if (A == 5) {
        if (G == 0) y = (x < -expRange) ? 0 : one / (exp(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    }
    if (A == 6) {
        if (G == 0) y = (x >= 0) ? x : exp(x) - one;
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + one);
    }
    if (A == 7) {
        if (G == 0) y = (x >= 0) ? seluScale * x : (seluScale * seluAlpha) * (exp(x) - one);
        if (G == 1) y = (yy >= 0) ? x * seluScale : x * (yy + seluScale * seluAlpha);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + seluScale * seluAlpha);
    }
    if (A == 8) {
        if (G == 0) y = (x > expRange) ? x : log(exp(x) + one);
        if (G == 1) y = x * (one - exp(-yy));
        if (G == 2) { S c = exp(-yy); y = x * c * (one - c); }
    }
    if (A == 9) {
        if (G == 0) y = (x < -expRange) ? 0 : x / (exp(-x) + one);
        else {
            S c = exp(xref), d = c + one;
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);
            else        y = (xref > halfExpRange) ? 0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -expRange) ? 0 : xref / (exp(-xref) + one) * gain;
        }
    }
    y *= gain * dy;
    if (clamp >= 0) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0) ? clamp : -clamp;
        else        y = (yref > -clamp && yref < clamp) ? y : 0;
    }
    return y;
}

// scalar kernel (any alignment, fp64, tails)
template <class T, int A>
__global__ void __launch_bounds__(256) bias_act_scalar(BiasActParams p, int64_t begin) {
    typedef typename Elem<T>::acc_t S;
    for (int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.sizeX; i += (int64_t)gridDim.x * 256) {
        S x = Elem<T>::load((const T*)p.x + i);
        S b = p.b ? Elem<T>::load((const T*)p.b + (i / p.stepB) % p.sizeB) : (S)0;
        S xref = p.xref ? Elem<T>::load((const T*)p.xref + i) : (S)0;
        S yref = p.yref ? Elem<T>::load((const T*)p.yref + i) : (S)0;
        S dy = p.dy ? Elem<T>::load((const T*)p.dy + i) : (S)1;
        if (p.grad == 0) x += b; else xref += b;
        S y = act_eval<S, A>(x, xref, yref, dy, p.grad, (S)p.alpha, (S)p.gain, (S)p.clamp);
        Elem<T>::store((T*)p.y + i, y);
    }
}

// 16-byte vector kernel: VEC elements per lane per access; requires all pointers 16-byte aligned.
// Access pattern chosen by measurement (tools/probe/stream_variants.hip, 64 x 64 x 256 x 256 bias + lrelu, of 8 TB/s):
//   grid-stride loop over a capped grid, one access per lane and iteration (the first version)      bf16 0.60-0.67   fp32 0.59-0.64
//   the same with 4 stride-separated accesses in flight per lane                                      0.48-0.54        0.56-0.59
//   FULL grid, a block owns 4 x 256 CONSECUTIVE vectors, all loads issued before the first use       0.73             0.78
//   ... with non-temporal loads and stores                                                            0.82             0.85
// i.e. on this chip a streaming pass wants every workgroup short and its bytes contiguous; capping the grid (the usual "2 048 blocks and
// stride the rest") costs 15-25 %.  All indices are 32-bit (size_x <= INT32_MAX is part of the contract); the bias index of the
// channels-last case advances by a constant with one conditional subtract instead of a 64-bit modulo per access.  Non-temporal accesses
// are used for tensors that cannot stay in the 256 MB last-level cache anyway (NT); small ones keep the default policy so that the
// consumer kernel still finds them there.
template <class T, int VEC> struct RawVec;
template <> struct RawVec<float, 4> {
    static __device__ __forceinline__ void unpack(u32x4 t, float (&v)[4]) {
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
};
template <class T> struct RawVec<T, 8> {
    static __device__ __forceinline__ void unpack(u32x4 t, float (&v)[8]) {
        Pack16<T>::unpack(t.x, v[0], v[1]); Pack16<T>::unpack(t.y, v[2], v[3]);
        Pack16<T>::unpack(t.z, v[4], v[5]); Pack16<T>::unpack(t.w, v[6], v[7]);
    }
};

// BMODE: 0 no bias, 1 channels-last (VEC consecutive channels per access), 2 one channel per access (NCHW, H*W % VEC == 0), 3 per element
// REFS: any of xref / yref / dy present (the gradient passes); the forward instantiation carries no registers for them (134 -> ~50 VGPRs:
// three waves per SIMD are too few for a streaming pass)
template <class T, int VEC, int A, int U, int BMODE, bool NT, bool REFS>
__global__ void __launch_bounds__(256) bias_act_vec(BiasActParams p, uint32_t nvec, uint32_t bstep) {
    const uint32_t v0 = blockIdx.x * (256u * U) + threadIdx.x;         // a block owns U * 256 consecutive vectors
    const uint32_t sizeB = (uint32_t)p.sizeB, stepB = (uint32_t)p.stepB;
    uint32_t bidx = BMODE == 1 ? (v0 * VEC) % sizeB : 0u;               // < 2^31: v0 * VEC <= size_x
    const u32x4* px = (const u32x4*)p.x;
    const u32x4* pxr = (const u32x4*)p.xref;
    const u32x4* pyr = (const u32x4*)p.yref;
    const u32x4* pdy = (const u32x4*)p.dy;
    auto ld = [](const u32x4* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
    u32x4 rx[U], rxr[REFS ? U : 1], ryr[REFS ? U : 1], rdy[REFS ? U : 1], rb[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t v = v0 + k * 256u;
        if (v < nvec) {
            rx[k] = ld(px + v);
            if (REFS && p.xref) rxr[k] = ld(pxr + v);
            if (REFS && p.yref) ryr[k] = ld(pyr + v);
            if (REFS && p.dy) rdy[k] = ld(pdy + v);
        }
        if (BMODE == 1) {                                             // (the index is valid for every k: no bounds condition)
            rb[k] = *(const u32x4*)((const T*)p.b + bidx);
            bidx += bstep;
            bidx = bidx >= sizeB ? bidx - sizeB : bidx;
        }
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        const uint32_t v = v0 + k * 256u;
        if (v >= nvec) break;
        float x[VEC], xr[VEC], yr[VEC], dy[VEC], b[VEC], y[VEC];
        RawVec<T, VEC>::unpack(rx[k], x);
        if (REFS && p.xref) RawVec<T, VEC>::unpack(rxr[k], xr);
        if (REFS && p.yref) RawVec<T, VEC>::unpack(ryr[k], yr);
        if (REFS && p.dy) RawVec<T, VEC>::unpack(rdy[k], dy);
        if (BMODE == 1) RawVec<T, VEC>::unpack(rb[k], b);
        if (BMODE == 2) {
            const float bv = (float)Elem<T>::load((const T*)p.b + ((v * VEC) / stepB) % sizeB);
#pragma unroll
            for (int e = 0; e < VEC; e++) b[e] = bv;
        }
        if (BMODE == 3) {
#pragma unroll
            for (int e = 0; e < VEC; e++) b[e] = (float)Elem<T>::load((const T*)p.b + ((v * VEC + e) / stepB) % sizeB);
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            float xx = x[e], xref = (REFS && p.xref) ? xr[e] : 0.f, yref = (REFS && p.yref) ? yr[e] : 0.f, d = (REFS && p.dy) ? dy[e] : 1.f;
            const float bb = BMODE ? b[e] : 0.f;
            if (p.grad == 0) xx += bb; else xref += bb;
            y[e] = act_eval<float, A>(xx, xref, yref, d, p.grad, p.alpha, p.gain, p.clamp);
        }
        u32x4 out;
        if constexpr (VEC == 4) { out.x = __float_as_uint(y[0]); out.y = __float_as_uint(y[1]); out.z = __float_as_uint(y[2]); out.w = __float_as_uint(y[3]); }
        else { out.x = Pack16<T>::pack(y[0], y[1]); out.y = Pack16<T>::pack(y[2], y[3]); out.z = Pack16<T>::pack(y[4], y[5]); out.w = Pack16<T>::pack(y[6], y[7]); }
        if (NT) __builtin_nontemporal_store(out, (u32x4*)p.y + v); else *((u32x4*)p.y + v) = out;
    }
}

template <class T, int A>
static void launch_act(const BiasActParams& p, hipStream_t st) {
    constexpr int VEC = sizeof(T) == 4 ? 4 : (sizeof(T) == 2 ? 8 : 0);
    int64_t done = 0;
    if constexpr (VEC > 0) {
        auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
        if (al(p.x) && al(p.y) && al(p.xref) && al(p.yref) && al(p.dy) && al(p.b)) {
            int64_t nvec = p.sizeX / VEC;
            if (nvec > 0) {
                const bool refs = p.xref || p.yref || p.dy;
                constexpr int U = 4, UR = 2;
                const dim3 grid((unsigned)agf_ceil_div(nvec, 256 * (refs ? UR : U))), block(256);
                const uint32_t n = (uint32_t)nvec;
                const int mode = !p.b ? 0 : (p.stepB == 1 && p.sizeB % VEC == 0) ? 1 : (p.stepB % VEC == 0) ? 2 : 3;
                const uint32_t bstep = mode == 1 ? (uint32_t)((256u * VEC) % (uint32_t)p.sizeB) : 0u;
                const bool nt = nvec * 16 > (int64_t)(64 << 20);     // x and y together exceed half of the last-level cache
#define BA_LAUNCH(M, NTV) do { if (refs) hipLaunchKernelGGL((bias_act_vec<T, VEC, A, UR, M, false, true>), grid, block, 0, st, p, n, bstep); \
                               else hipLaunchKernelGGL((bias_act_vec<T, VEC, A, U, M, NTV, false>), grid, block, 0, st, p, n, bstep); } while (0)
                switch (mode) {
                    case 0: if (nt) BA_LAUNCH(0, true); else BA_LAUNCH(0, false); break;
                    case 1: if (nt) BA_LAUNCH(1, true); else BA_LAUNCH(1, false); break;
                    case 2: if (nt) BA_LAUNCH(2, true); else BA_LAUNCH(2, false); break;
                    default: if (nt) BA_LAUNCH(3, true); else BA_LAUNCH(3, false); break;
                }
#undef BA_LAUNCH
                done = nvec * VEC;
            }
        }
    }
    if (done < p.sizeX) {
        int64_t rest = p.sizeX - done;
        int64_t blocks = agf_ceil_div(rest, 256);
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL((bias_act_scalar<T, A>), dim3((unsigned)blocks), dim3(256), 0, st, p, done);
    }
}

template <class T>
static int launch_typed(const BiasActParams& p, int act, hipStream_t st) {
    switch (act) {
        case 1: launch_act<T, 1>(p, st); break;
        case 2: launch_act<T, 2>(p, st); break;
        case 3: launch_act<T, 3>(p, st); break;
        case 4: launch_act<T, 4>(p, st); break;
        case 5: launch_act<T, 5>(p, st); break;
        case 6: launch_act<T, 6>(p, st); break;
        case 7: launch_act<T, 7>(p, st); break;
        case 8: launch_act<T, 8>(p, st); break;
        case 9: launch_act<T, 9>(p, st); break;
        default: agf_set_error("no kernel found for the specified activation func (%d)", act); return AGF_EINVAL;
    }
    return AGF_OK;
}

extern "C" int agf_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                            int dtype, int64_t size_x, int32_t size_b, int64_t step_b,
                            int grad, int act, float alpha, float gain, float clamp, void* stream) {
    // validation mirrors bias_act.cpp:29-44
    AGF_CHECK(x && y, "bias_act: null pointer");
    AGF_CHECK(dtype >= AGF_F32 && dtype <= AGF_F64, "bias_act: unsupported dtype %d", dtype);
    AGF_CHECK(size_x >= 0 && size_x <= INT32_MAX, "x is too large");
    AGF_CHECK(grad >= 0 && grad <= 2, "grad must be 0, 1 or 2");
    AGF_CHECK(b == nullptr || (size_b >= 1 && step_b >= 1), "b has wrong number of elements");
    if (size_x == 0) return AGF_OK;
    BiasActParams p;
    p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y;
    p.sizeX = size_x; p.sizeB = b ? size_b : 1; p.stepB = b ? step_b : 1;
    p.grad = grad; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case AGF_F32:  rc = launch_typed<float>(p, act, st); break;
        case AGF_F16:  rc = launch_typed<f16_t>(p, act, st); break;
        case AGF_BF16: rc = launch_typed<bf16_t>(p, act, st); break;
        default:       rc = launch_typed<double>(p, act, st); break;
    }
    if (rc != AGF_OK) return rc;
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}
