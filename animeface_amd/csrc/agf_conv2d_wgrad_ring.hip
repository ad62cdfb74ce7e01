// Weight gradient of the 3x3 convolution as a multi-stage direct-to-LDS ring (3x3, stride 1, "same"; 128-pixel tiles, ragged maps with
// >= 70 % tile coverage, and 8x8 maps as one half-empty tile per image: every StyleGAN2 map from 8x8 up).
//
// conv2d_wgrad_kernel<3, true, true> (agf_conv2d.hip) double-buffers whole 256-pixel tile sets: it issues tile t+1, contracts tile t and
// then drains `vmcnt(0)` -- 76 KB in flight per CU right after the issue and nothing towards the end of the phase; PMC showed its waves
// parked on vmcnt 45 % of the time with the matrix pipe 37 % busy.  Here the same contraction (64 co x 64 ci x 9 taps per block, one
// 32 x 32 quadrant x 9 taps = 144 accumulator registers per wave, transposing LDS reads) runs over 128-pixel tiles through a ring of
// NS = 3 stages of 42.6 KB: while tile t is contracted the loads of tiles t+1 and t+2 are in flight, and the wait at the top of a tile
// is PARTIAL -- `s_waitcnt vmcnt(6)`: vector-memory operations retire in order and every wave issues exactly 6 per tile (2 dy + 4 x
// wave-loads; the ones beyond the tile or beyond the work list are issued with out-of-range offsets into a 1 KB dummy region), so "all
// but the youngest 6 have completed" is "tile t has landed".  One bare s_barrier per tile (no fence: __syncthreads() would drain vmcnt).
#include "agf_conv2d_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradRingParams {
    const bf16_t* x;          // [N,H,W,Cin]
    const bf16_t* dy;         // [N,H,W,Cout]
    float* dw;                // [Cout,3,3,Cin] fp32, accumulated into
    const float* in_scale;    // [N,Cin] or null   (applied to the fp32 partial sums of image n: blocks stay inside one image then)
    const float* out_scale;   // [N,Cout] or null
    int N, H, W, Cin, Cout;
    int TH, TW, twShift;      // pixel tile: TH x TW = 128, TW in {16, 32}
    int tilesW, tilesH, pixTiles, lgTilesW, lgTilesH;   // tiles per row / column of an image; lg* >= 0: powers of two (shift decode)
    int cutX, cutY;           // first patch column / row (halo included) that lies beyond the image in the LAST tile column / row
    int tilesCo, tilesCi, splitK;
    float scale;
    int epiScale, perImage;
    float* part;              // two-stage combine: [splitK][Cout,3,3,Cin] partial sums written with plain stores (null: fp32 atomics into dw)
    int64_t dwNumel;
};

static __device__ __forceinline__ bf16x8 ring_tr_frag(const bf16_t* base) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * 32));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

#define RING_OOB 0x7fff0000
#define RING_DYR 128           // dy rows (pixels) of a tile
#define RING_XRMAX 208         // x rows of a tile incl. halo, rounded to 16: 6 x 34 = 204 -> 208 (TW = 32), 10 x 18 = 180 -> 192 (TW = 16)

// NK = k-steps per wave and tile = 8 / (k subsets): 4 when the block has four 32 x 32 quadrants, 2 with two (Cin <= 32 or Cout <= 32), 1 with one.
// SC: per-image operand scales applied to the fragments in registers (a wave's lane holds ONE channel of each operand: one scalar per
//     lane and tile, fetched a tile ahead) -- for the launches whose blocks cannot stay inside one image (small maps with many images),
//     where the scales cannot ride on the partial sums; ~230 VALU operations per tile in the shadow of its 36 MFMAs.
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t ring_scale2(uint32_t v, float s) {
    return Pack16<bf16_t>::pack(__uint_as_float(v << 16) * s, __uint_as_float(v & 0xffff0000u) * s);
}
template <int NS, int NK, bool SC = false>
__global__ void __launch_bounds__(512, 2) conv2d_wgrad_ring_kernel(WgradRingParams p) {
    constexpr int TAPS = 9;
    constexpr int DYR = RING_DYR;
    constexpr int DV = 2, XV = 4;                                        // wave-loads per wave and tile (8 waves: 16 dy, up to 32 x slots)
    constexpr int LPW = DV + XV + (SC ? 2 : 0);                           // SC: + the two scale vectors of the tile's image
    static_assert(LPW * (NS - 2) <= 63, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int PW = p.TW + 2, PH = p.TH + 2;
    const int P = PH * PW;
    const int XR = (P + 15) & ~15;
    const int STAGE_E = (2 * DYR + 2 * XR) * 32;                          // elements of one stage: dy [2][DYR][32], x [2][XR][32]
    bf16_t* sBase = (bf16_t*)smem_raw;
    bf16_t* sDummy = sBase + NS * STAGE_E;                                // 1 KB: target of the wave-loads that carry nothing
    float* sSide = (float*)(sDummy + 512);                                // SC: [NS][2][8 waves][64 lanes] per-lane operand scales of a stage's tile

    int bid = blockIdx.x;
    const int ks = bid % p.splitK; bid /= p.splitK;
    const int tci = bid % p.tilesCi;
    const int tco = bid / p.tilesCi;
    const int co0 = tco * 64, ci0 = tci * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int realQ = (p.Cout > 32 ? 2 : 1) * (p.Cin > 32 ? 2 : 1);
    const int quad = wave & (realQ - 1), kidx = wave / realQ, kSplit = 8 / realQ;
    const int wa = p.Cout > 32 ? (p.Cin > 32 ? quad >> 1 : quad) : 0;      // co block
    const int wb = p.Cin > 32 ? (quad & 1) : 0;                            // ci block
    const int li = lane & 15, lg = (lane >> 4) & 1, lk = lane >> 5;
    const int laneOff = (8 * lk + (li >> 2)) * 32 + 16 * lg + 4 * (li & 3);
    const int aOff = wa * DYR * 32 + laneOff;
    const int bOff = 2 * DYR * 32 + wb * XR * 32 + laneOff;
    // k-step (row r, 16-pixel column run h) of the tile -> wave subset kidx owns column run kidx % (TW / 16), rows rowStart .. + NK - 1
    const int stepsPerRow = p.TW >> 4;
    const int colHalf = kidx % stepsPerRow, rowStart = (kidx / stepsPerRow) * NK;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // Addressing.  Everything that does not depend on the tile is computed once: per wave-load slot v = (i * 8 + wave) * 64 + lane (LDS-
    // linear order [32-channel block][row][16-byte chunk]) the byte offset of the lane's vector relative to the tile origin and a flag
    // word -- bit 0..3: the vector lies in the top halo row / at or below the row that leaves the image in the last tile row / in the left halo
    // column / at or beyond the column that leaves the image in the last tile column (for maps that are whole tiles: the halo ring), bit 4: never loaded (channel tail, slot padding),
    // bit 5: always set.  A tile contributes one scalar mask (which halo sides fall outside the image, bit 4, bit 5 for tiles beyond
    // the work list): offset = (flags & mask) ? out of range : origin + relative -- four vector instructions per load, placed between
    // the MFMAs of the tile being contracted (when this block was issued in one piece right after the barrier, with a division-based
    // tile decode, the eight waves spent ~40 % of a tile's time in it in lock step with the matrix pipe idle).
    int dRel[DV], dFlag[DV], xRel[XV], xFlag[XV];
    int dLds[DV], xLds[XV];         // LDS element offset of the wave-load inside a stage (wave-uniform), or -1: dummy region
#pragma unroll
    for (int i = 0; i < DV; i++) {
        const int v = (i * 8 + wave) * 64 + lane;                         // < 1024 = DYR * 8 always
        const int blk = v / (DYR * 4), rem = v - blk * (DYR * 4);
        const int q = rem >> 2, ch = (blk * 4 + (rem & 3)) * 8;
        const int r = q >> p.twShift, c = q & (p.TW - 1);
        dRel[i] = ((r * p.W + c) * p.Cout + co0 + ch) * 2;
        dFlag[i] = 32 | (co0 + ch < p.Cout ? 0 : 16) | (r >= p.cutY - 1 ? 2 : 0) | (c >= p.cutX - 1 ? 8 : 0);
        dLds[i] = (i * 8 + wave) * 512;
    }
#pragma unroll
    for (int i = 0; i < XV; i++) {
        const int wl = i * 8 + wave;
        const int v = wl * 64 + lane;
        const bool slot = wl * 8 < XR;                                    // XR * 8 vectors = XR / 8 wave-loads
        const int blk = v / (XR * 4), rem = v - blk * (XR * 4);
        const int q = rem >> 2, ch = (blk * 4 + (rem & 3)) * 8;
        const int pr = q / PW, pc = q - pr * PW;
        xRel[i] = (((pr - 1) * p.W + pc - 1) * p.Cin + ci0 + ch) * 2;
        xFlag[i] = 32 | ((slot && q < P && ci0 + ch < p.Cin) ? 0 : 16) | (pr == 0 ? 1 : 0) | (pr >= p.cutY ? 2 : 0) | (pc == 0 ? 4 : 0) | (pc >= p.cutX ? 8 : 0);
        xLds[i] = slot ? 2 * DYR * 32 + wl * 512 : -1;
    }

    const int tilesPerImage = p.tilesH * p.tilesW;
    // one buffer resource per tensor (the launcher checks that both are below 2 GB)
    const __amdgpu_buffer_rsrc_t dRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, RING_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t xRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, RING_OOB, 0x00020000);
    // SC: this lane's channel of the A operand (dy: co) and of the B operand (x: ci); tail lanes hold zero fragments: any finite scale
    const int chA = co0 + wa * 32 + (lane & 31), chB = ci0 + wb * 32 + (lane & 31);
    const int chAc = chA < p.Cout ? chA : p.Cout - 1, chBc = chB < p.Cin ? chB : p.Cin - 1;
    // (either scale may be absent -- e.g. a gradient stored already times the demodulation scale: an empty resource then, every load of
    //  it returns zero without touching memory, and the fragment scale below is 1)
    const __amdgpu_buffer_rsrc_t aScRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.out_scale, 0, (SC && p.out_scale) ? p.N * p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bScRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.in_scale, 0, (SC && p.in_scale) ? p.N * p.Cin * 4 : 0, 0x00020000);
    auto issue_tile = [&](int pt, bool live, int stage) {
        int tw, th, n;
        if (p.lgTilesW >= 0) { tw = pt & (p.tilesW - 1); th = (pt >> p.lgTilesW) & (p.tilesH - 1); n = pt >> (p.lgTilesW + p.lgTilesH); }
        else { n = pt / tilesPerImage; const int rem = pt - n * tilesPerImage; th = rem / p.tilesW; tw = rem - th * p.tilesW; }
        const int pix = (n * p.H + th * p.TH) * p.W + tw * p.TW;
        const int dOrigin = pix * p.Cout * 2, xOrigin = pix * p.Cin * 2;
        const int mask = !live ? 63 : 16 | (th == 0 ? 1 : 0) | (th == p.tilesH - 1 ? 2 : 0) | (tw == 0 ? 4 : 0) | (tw == p.tilesW - 1 ? 8 : 0);
        bf16_t* sS = sBase + stage * STAGE_E;
#pragma unroll
        for (int i = 0; i < DV; i++) {
            const int off = (dFlag[i] & mask) ? RING_OOB : dOrigin + dRel[i];
            bf16_t* dst = sS + dLds[i];                                   // (operands of the builtin go through locals: see agf_conv2d_pipe.hip)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dRes, (lds_ptr)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XV; i++) {
            const int off = (xFlag[i] & mask) ? RING_OOB : xOrigin + xRel[i];
            bf16_t* dst = xLds[i] >= 0 ? sS + xLds[i] : sDummy;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xRes, (lds_ptr)dst, 16, off, 0, 0, 0);
        }
        if (SC) {
            // this lane's scale of each operand for the tile's image, as two more DMA operations of the same group: they land with the tile
            // (a register-carried asynchronous load cannot be made safe here: the compiler may copy the destination register before the
            //  data has arrived -- it does not know the inline-asm load is one)
            const int offA = live ? (n * p.Cout + chAc) * 4 : RING_OOB, offB = live ? (n * p.Cin + chBc) * 4 : RING_OOB;
            float* dA = sSide + ((stage * 2 + 0) * 8 + wave) * 64;
            float* dB = sSide + ((stage * 2 + 1) * 8 + wave) * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(aScRes, (lds_ptr)dA, 4, offA, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bScRes, (lds_ptr)dB, 4, offB, 0, 0, 0);
        }
    };

    // work list of this block: with per-image scales a contiguous run of tiles of ONE image, otherwise every splitK-th tile
    const int curN = p.epiScale ? ks / p.perImage : 0;
    const int run = (tilesPerImage + p.perImage - 1) / p.perImage;
    const int ptStep = p.epiScale ? 1 : p.splitK;
    const int ptBegin = p.epiScale ? curN * tilesPerImage + (ks % p.perImage) * run : ks;
    int ptEnd = p.pixTiles;
    if (p.epiScale) { ptEnd = ptBegin + run; if (ptEnd > (curN + 1) * tilesPerImage) ptEnd = (curN + 1) * tilesPerImage; }
    if (ptBegin >= ptEnd && !p.part) return;             // block-uniform (two-stage combine: an idle block still writes its zeros)

    int ptIssue = ptBegin, stIssue = 0;
#pragma unroll
    for (int s = 0; s < NS - 1; s++) {
        issue_tile(ptIssue, ptIssue < ptEnd, stIssue);
        ptIssue += ptStep; stIssue = stIssue + 1 == NS ? 0 : stIssue + 1;
    }
    int cur = 0;
    for (int pt = ptBegin; pt < ptEnd; pt += ptStep) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPW * (NS - 2)) : "memory");   // this wave's part of tile pt has landed ...
        __builtin_amdgcn_s_barrier();                                            // ... everyone's has, and everyone has left tile pt - 1
        __builtin_amdgcn_sched_barrier(0);
        float scA = 1.f, scB = 1.f;
        if (SC) {
            scA = p.out_scale ? sSide[((cur * 2 + 0) * 8 + wave) * 64 + lane] : 1.f;
            scB = p.in_scale ? sSide[((cur * 2 + 1) * 8 + wave) * 64 + lane] : 1.f;
        }
        const bf16_t* aCur = sBase + cur * STAGE_E + aOff + (rowStart * p.TW + colHalf * 16) * 32;
        const bf16_t* bCur = sBase + cur * STAGE_E + bOff + (rowStart * PW + colHalf * 16) * 32;
        // Fragment reads: this wave contracts NK k-steps = 16-pixel runs of NK consecutive tile rows at one column offset.  The three
        // kw taps of a patch row are the SAME 18 pixels shifted by 0 / 1 / 2 along k, and the three kh taps of consecutive k-steps share
        // patch rows, so a patch row is fetched once -- k 0..7 | 8..15 (two transposing reads) plus k 8..11 | 16..19 (one) = five
        // dwords per lane -- and the shifted operands are built in registers (kw = 2: dword renaming, kw = 1: four v_alignbit):
        // (NK + 2) * 3 + NK * 2 transposing reads per NK * 9 MFMAs instead of NK * 20.  The reads were this kernel's bottleneck:
        // without global loads the loop ran at 46 % of the MFMA rate with 20 reads per k-step and at 62 % with none.
        u32x4 rowF[NK + 2]; uint32_t rowE[NK + 2];
        bf16x8 aF[NK];
#pragma unroll
        for (int j = 0; j < NK + 2; j++) {
            const bf16_t* rp = bCur + j * PW * 32;
            const bf16x8 f0 = ring_tr_frag(rp);
            const s16x4 e = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rp + 8 * 32));
            rowF[j] = __builtin_bit_cast(u32x4, f0);
            rowE[j] = __builtin_bit_cast(uint2_t, e).x;
            if (j < NK) aF[j] = ring_tr_frag(aCur + j * p.TW * 32);
        }
        if (SC) {
#pragma unroll
            for (int j = 0; j < NK + 2; j++) {
                rowF[j].x = ring_scale2(rowF[j].x, scB); rowF[j].y = ring_scale2(rowF[j].y, scB);
                rowF[j].z = ring_scale2(rowF[j].z, scB); rowF[j].w = ring_scale2(rowF[j].w, scB);
                rowE[j] = ring_scale2(rowE[j], scB);
                if (j < NK) {
                    u32x4 a = __builtin_bit_cast(u32x4, aF[j]);
                    a.x = ring_scale2(a.x, scA); a.y = ring_scale2(a.y, scA); a.z = ring_scale2(a.z, scA); a.w = ring_scale2(a.w, scA);
                    aF[j] = __builtin_bit_cast(bf16x8, a);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NK; j++) {
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const u32x4 d = rowF[j + kh];
                const uint32_t e0 = rowE[j + kh];
                u32x4 s1, s2;
                s1.x = __builtin_amdgcn_alignbit(d.y, d.x, 16); s1.y = __builtin_amdgcn_alignbit(d.z, d.y, 16);
                s1.z = __builtin_amdgcn_alignbit(d.w, d.z, 16); s1.w = __builtin_amdgcn_alignbit(e0, d.w, 16);
                s2.x = d.y; s2.y = d.z; s2.z = d.w; s2.w = e0;
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aF[j], __builtin_bit_cast(bf16x8, d), acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aF[j], __builtin_bit_cast(bf16x8, s1), acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aF[j], __builtin_bit_cast(bf16x8, s2), acc[kh * 3 + 2], 0, 0, 0);
            }
            if (j == 0) {                                                 // the DMA of tile pt + (NS - 1) steps, into the stage tile pt - 1 occupied
                issue_tile(ptIssue, ptIssue < ptEnd, stIssue);
                ptIssue += ptStep; stIssue = stIssue + 1 == NS ? 0 : stIssue + 1;
            }
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing (empty) loads still target LDS
    // k-step subsets hold partial sums of the same quadrants: add them through LDS, then one wave per quadrant issues the atomics
    {
        float* sRed = (float*)smem_raw;
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            __syncthreads();
            if (kidx != 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) sRed[(wave * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (kidx == 0) {
                for (int k = 1; k < kSplit; k++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[t][r] += sRed[((k * realQ + quad) * 16 + r) * 64 + lane];
                }
            }
        }
    }
    if (kidx != 0) return;
    const int ci = ci0 + wb * 32 + (lane & 31);
    if (ci >= p.Cin) return;
    float si = p.scale;
    if (p.epiScale && p.in_scale) si *= p.in_scale[(int64_t)curN * p.Cin + ci];
    float* part = p.part ? p.part + (int64_t)ks * p.dwNumel : nullptr;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int co = co0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co >= p.Cout) continue;
        float sc = si;
        if (p.epiScale && p.out_scale) sc *= p.out_scale[(int64_t)curN * p.Cout + co];
        if (part) {
#pragma unroll
            for (int t = 0; t < TAPS; t++) part[((int64_t)co * TAPS + t) * p.Cin + ci] = acc[t][r] * sc;
        } else {
#pragma unroll
            for (int t = 0; t < TAPS; t++) unsafeAtomicAdd(p.dw + ((int64_t)co * TAPS + t) * p.Cin + ci, acc[t][r] * sc);
        }
    }
}

// second stage of the two-stage combine: dw[i] = sum_ks part[ks][i].  256 threads = 64 float4 lanes x 4 interleaved subsets of ks.
// oihwCin > 0: dw is written in the parameter's own [Cout][Cin][3][3] order (the partial sums are [Cout][3][3][Cin]), so that autograd can
// hand the tensor to the optimizer without a layout copy.
__global__ void __launch_bounds__(256) conv2d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int64_t numel, int splitK,
                                                                  int oihwCin) {
    __shared__ f32x4 red[3][64];
    const int l = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const int64_t i4 = (int64_t)blockIdx.x * 64 + l;
    const bool live = i4 * 4 < numel;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = part + i4 * 4;
        int ks = kp;
        for (; ks + 4 < splitK; ks += 8) {
            const f32x4 u = *(const f32x4*)(src + (int64_t)ks * numel), v = *(const f32x4*)(src + (int64_t)(ks + 4) * numel);
            a += u; b += v;
        }
        if (ks < splitK) a += *(const f32x4*)(src + (int64_t)ks * numel);
    }
    a += b;
    if (kp) red[kp - 1][l] = a;
    __syncthreads();
    if (kp == 0 && live) {
        a += red[0][l]; a += red[1][l]; a += red[2][l];
        if (oihwCin > 0) {
            const int64_t i = i4 * 4;                            // = (co * 9 + t) * Cin + ci, four consecutive ci
            const int ci = (int)(i % oihwCin);
            const int64_t r = i / oihwCin;
            const int t = (int)(r % 9);
            float* dst = dw + ((r / 9) * oihwCin + ci) * 9 + t;
            dst[0] = a.x; dst[9] = a.y; dst[18] = a.z; dst[27] = a.w;
        } else *(f32x4*)(dw + i4 * 4) = a;
    }
}

static int ring_pow2_floor_log2(int v) { int s = 0; while ((2 << s) <= v) s++; return s; }

// Tile geometry, split and scale mode of a launch; false: shape not covered (the caller falls through to conv2d_wgrad_kernel)
static bool ring_plan(WgradRingParams& p, bool scales, bool both, int N, int H, int W, int Cin, int Cout) {
    constexpr int on = 1;
    if (!on) return false;
    if (W < 8 || H < 4) return false;
    {   // 4 x 32 or 8 x 16 pixel tiles: whichever wastes less of its area on this map
        auto cover = [&](int tw) { const int th = RING_DYR / tw; return (double)H * W / ((double)((W + tw - 1) / tw) * tw * ((H + th - 1) / th) * th); };
        p.TW = (W >= 32 && cover(32) >= cover(16)) ? 32 : 16;
    }
    p.TH = RING_DYR / p.TW;
    if ((int64_t)N * H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fff0000ll) return false;     // 32-bit byte offsets into the whole tensor
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.twShift = ring_pow2_floor_log2(p.TW);
    p.tilesW = (W + p.TW - 1) / p.TW; p.tilesH = (H + p.TH - 1) / p.TH;
    p.lgTilesW = ring_pow2_floor_log2(p.tilesW); p.lgTilesH = ring_pow2_floor_log2(p.tilesH);
    if ((1 << p.lgTilesW) != p.tilesW || (1 << p.lgTilesH) != p.tilesH) p.lgTilesW = p.lgTilesH = -1;      // ragged maps: division decode
    p.cutX = W - (p.tilesW - 1) * p.TW + 1; p.cutY = H - (p.tilesH - 1) * p.TH + 1;
    {   // tiles that hang over the image contract zeros: not worth it when more than ~30 % of the tile area is padding
        constexpr int ragged = 1;
        constexpr double minCover = 0.7;
        const double cover = (double)H * W / ((double)p.tilesW * p.TW * p.tilesH * p.TH);
        // (8x8 maps are ONE half-empty 8x16 tile per image: still 75 -> 60 us for 512 -> 512 channels at batch 64 against the staging kernel)
        const double need = (p.tilesW * p.tilesH == 1 && minCover > 0.45) ? 0.45 : minCover;
        if ((W % p.TW || H % p.TH) && (!ragged || cover < need)) return false;
    }
    const int tpi = p.tilesW * p.tilesH;
    p.pixTiles = tpi * N;
    p.tilesCo = (Cout + 63) / 64; p.tilesCi = (Cin + 63) / 64;
    const int base = p.tilesCo * p.tilesCi;
    constexpr int wantBlocks = 0;
    const int want = ((wantBlocks ? wantBlocks : 256) + base - 1) / base;
    const int cap = p.pixTiles / 8 < 1 ? 1 : p.pixTiles / 8;          // >= 8 tiles of 128 pixels per block (its 64x64x9 partial sums must stay cheap)
    p.splitK = want < 1 ? 1 : (want > cap ? cap : want);
    p.epiScale = 0; p.perImage = 1;
    if (scales) {
        // per-image scales ride on the partial sums: every block stays inside one image
        int m = (want + N - 1) / N;
        if (m > tpi / 8) m = tpi / 8;
        if (m < 1) m = 1;
        constexpr int sc_on = 1;
        if ((N * m > want + want / 2 && tpi / m < 24) || tpi < 8) {
            if (!sc_on || !both) return false;
            p.epiScale = 0;                                  // blocks span images: scales on the operands (SC kernel), split as without scales
        } else { p.epiScale = 1; p.perImage = m; p.splitK = N * m; }
    }
    p.dwNumel = (int64_t)Cout * 9 * Cin;
    const int XR = ((p.TH + 2) * (p.TW + 2) + 15) & ~15;
    return (size_t)3 * (2 * RING_DYR + 2 * XR) * 32 * sizeof(bf16_t) + 1024 + 3 * 2 * 8 * 64 * 4 <= 160 * 1024;
}

// bytes of the [splitK][Cout,3,3,Cin] fp32 scratch of the two-stage combine, 0 = shape not covered
int64_t agf_conv2d_wgrad_ring_workspace(bool scales, int N, int H, int W, int Cin, int Cout) {
    constexpr int two = 1;
    WgradRingParams p;
    if (!two || !ring_plan(p, scales, scales, N, H, W, Cin, Cout) || p.splitK < 2) return 0;
    return (int64_t)p.splitK * p.dwNumel * 4;
}

// AGF_ENOKERNEL: shape not covered.  With a workspace of agf_conv2d_wgrad_ring_workspace() bytes the blocks write their partial sums
// there with plain stores and a second launch adds them into dw (which is then OVERWRITTEN, no zero-initialisation needed): the fp32
// atomics of the one-stage combine ran at 0.5 TB/s -- 76 us of a 190 us launch for the 37 MB that 256 blocks x 64x64x9 produce.
int agf_conv2d_wgrad_ring_launch(const void* x, const void* dy, float* dw, const float* in_scale, const float* out_scale,
                                 int N, int H, int W, int Cin, int Cout, float scale, float* workspace, int64_t workspaceBytes, int oihw, hipStream_t st) {
    WgradRingParams p;
    if (!ring_plan(p, in_scale || out_scale, in_scale && out_scale, N, H, W, Cin, Cout)) return AGF_ENOKERNEL;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dw = dw; p.in_scale = in_scale; p.out_scale = out_scale; p.scale = scale;
    p.part = (workspace && p.splitK >= 2 && workspaceBytes >= (int64_t)p.splitK * p.dwNumel * 4) ? workspace : nullptr;
    const int base = p.tilesCo * p.tilesCi;
    constexpr int NS = 3;
    const int XR = ((p.TH + 2) * (p.TW + 2) + 15) & ~15;
    const size_t lds = (size_t)NS * (2 * RING_DYR + 2 * XR) * 32 * sizeof(bf16_t) + 1024 + (size_t)NS * 2 * 8 * 64 * sizeof(float);
    const int realQ = (Cout > 32 ? 2 : 1) * (Cin > 32 ? 2 : 1);
    const bool sc = (in_scale || out_scale) && !p.epiScale;
    void (*kern)(WgradRingParams) = realQ == 4 ? (sc ? conv2d_wgrad_ring_kernel<NS, 4, true> : conv2d_wgrad_ring_kernel<NS, 4, false>)
                                  : realQ == 2 ? (sc ? conv2d_wgrad_ring_kernel<NS, 2, true> : conv2d_wgrad_ring_kernel<NS, 2, false>)
                                               : (sc ? conv2d_wgrad_ring_kernel<NS, 1, true> : conv2d_wgrad_ring_kernel<NS, 1, false>);
    static bool attr[6] = {false, false, false, false, false, false};
    const int slot = (realQ == 4 ? 0 : realQ == 2 ? 1 : 2) + (sc ? 3 : 0);
    if (!attr[slot]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { agf_set_error("conv2d_wgrad ring: cannot reserve LDS: %s", hipGetErrorString(e)); return AGF_ELAUNCH; }
        attr[slot] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(base * p.splitK)), dim3(512), lds, st, p);
    if (p.part)
        hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3((unsigned)((p.dwNumel / 4 + 63) / 64)), dim3(256), 0, st, p.part, dw, p.dwNumel, p.splitK, oihw ? Cin : 0);
    else if (workspace) return AGF_EINVAL;               // the caller asked for the overwriting mode but the plan changed: cannot happen
    return AGF_OK;
}
