// Persistent, multi-stage direct-to-LDS MFMA convolution for the high-resolution, few-channel layers (3x3, stride 1, "same";
// Cin in {32, 64, 128}, Cout <= 64 per block) -- the layers whose time is set by HBM streaming, not by the matrix pipe.
//
// What limited conv2d_fwd_dl_kernel on these shapes (64 -> 64 @256x256: 27 % MFMA busy, 2.2 TB/s): one K chunk of look-ahead against
// a loaded-memory latency of 2-3 us, a full `vmcnt(0)` drain per chunk, and at every tile boundary a cold prologue (loads issued,
// then waited for) plus an epilogue during which the block has nothing in flight.  Here
//   * one block per CU lives for the whole launch and walks its tiles; the (tile, K chunk) stages form ONE continuous pipeline of
//     NSTAGE LDS buffers filled by `buffer_load ... lds`: the loads of the next tile's first chunks are in flight while the current
//     tile is contracted and stored -- NSTAGE-1 stages (~3 x 38 KB per CU) of look-ahead everywhere;
//   * waits are partial: `s_waitcnt vmcnt(N)` with N = the number of vector-memory operations this wave issued AFTER the stage it
//     needs (LOADS retire in order among themselves, so the N youngest loads may stay in flight; the tile's stores are not counted
//     among those N -- a store may be acknowledged before an older load returns -- which makes the wait exact when no store is pending
//     and early otherwise: measured cost 0-4 % per launch).  For N to be a
//     compile-time constant every operation is issued unconditionally: DMA beyond the work list, absent epilogue operands and
//     out-of-image pixels use out-of-range buffer offsets (loads return zeros, stores are dropped); the chunk loop is unrolled
//     (NCH = Cin / 16 is a template parameter) and `__builtin_amdgcn_sched_barrier` pins the issue order the counts assume;
//   * epilogue operands (lrelu mask / pooled residual of agf_conv2d_fwd_mask, or demodulation scale + bias + noise) are requested
//     one or two chunks before the tile's last MFMA, the results leave as 16-byte vectors straight from registers
//     (v_permlane32_swap, see conv_epilogue_pl in agf_conv2d.hip) -- no LDS round trip, no per-tile block barrier;
//   * the blocks of one XCD sweep a contiguous band of pixel tiles together, so halo rows are re-read from that XCD's L2.
// Style-modulated layers use per-image weights (w + n * wImgStride: W * s[n] prepared by agf_prep_weights_mod), so the activation
// path stays a pure DMA stream.
#include "agf_conv2d_common.h"
#include <utility>

typedef __attribute__((address_space(3))) void* lds_ptr;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) (the chunk index feeds s_waitcnt immediates)
template <int... I, class F>
static __device__ __forceinline__ void pipe_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

struct PipeParams {
    ConvParams c;
    int tilesWl2, tilesHl2;     // log2(tiles per row), log2(tiles per column) of one image
    int band;                   // pixel tiles per XCD (contiguous)
    long long wImgStride;       // elements between per-image weight tensors (0 = one shared tensor)
    int yPix;                   // elements per pixel of y (= Cout unless this launch writes a channel slice of a wider tensor: EPI 0 only)
    int countStores;            // A/B: count a tile's stores among the operations a partial wait leaves in flight (see the wait)
    int dbg;                    // timing experiments (AGF_PIPE_DBG): bits 0-1: 1 = no epilogue at all, 2 = stores of zeros without the epilogue arithmetic; +4: weight DMA out of range (zero fill, no memory traffic); +8: the same for the activations
};

#define PIPE_OOB 0x7fff0000
#define PIPE_ISSUE_TAP 1          // the tap of a chunk after whose MFMAs the next DMA group is issued

template <int V> static __device__ __forceinline__ void pipe_wait_vm() {
    static_assert(V >= 0 && V <= 63, "vmcnt immediate out of range");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(V) : "memory");
}

// vector-memory operations a wave issues after the DMA group of the stage it waits for at chunk c of a tile (steady state):
// per iteration [P prefetch loads if chunk == PF] [LPW DMA] ... [S stores if chunk == NCH-1]; the group was issued NS-1 iterations ago
template <int NCH, int NS>
static constexpr int pipe_younger(int c, int lpw, int P, int S, int PF) {
    int n = (NS - 2) * lpw;
    for (int j = 1; j <= NS - 2; j++) if ((((c - j) % NCH) + NCH) % NCH == PF) n += P;
    for (int j = 1; j <= NS - 1; j++) if ((((c - j) % NCH) + NCH) % NCH == NCH - 1) n += S;
    return n;
}

// MT x NWM: 32-channel blocks per wave / waves along co; NWN waves along pixels x NJ 32-pixel sub-tiles each; NCH K chunks of 16.
// EPI: 0 = forward epilogue (demodulation scale, bias, noise, lrelu, gain), 1 = + lrelu mask of the layer below (agf_conv2d_fwd_mask),
//      2 = mask + pooled residual.  The per-channel operands of EPI 0 (out_scale[n], bias) travel as two extra 256-byte DMA loads
//      into a 512-byte tail of every stage buffer and are read from LDS: no registers, no vmcnt entanglement.
// WS (weights stationary; shared weights that fit: Cin <= 64): the whole Cout x 9 x Cin tensor is loaded ONCE per block into a fixed LDS
//      region and only the activation chunks stream through the ring.  Measured on 64 -> 64 @256x256, B = 128 with the epilogue off:
//      0.73 ms with the 74 KB of weights re-streamed from L2 for every tile, 0.58 ms without that traffic -- and the two streams cost
//      more together than the sum of each alone.
template <int KS, int MT, int NWM, int NWN, int NJ, int NCH, int NSTAGE, int PMAX, int EPI, bool WS>
__global__ void __launch_bounds__(64 * NWM * NWN, 2) conv2d_fwd_pipe_kernel(PipeParams pp) {
    // EPI 3 (agf_conv2d_fwd_pool): the operands of EPI 0, but what leaves the tile is its 2x2 average and the 1-bit sign mask of the
    // full-resolution result (see the pooled branch of the epilogue); everywhere else it is EPI 0
    // EPI 4: EPI 0 + the sign bits of the stored output (agf_conv2d_fwd_bits); EPI 5 / 6: EPI 1 / 2 with the lrelu mask read as bits
    // (agf_conv2d_fwd_maskbits: MT * NJ dwords per lane instead of 2 * MT * NJ 16-byte vectors)
    constexpr bool POOL = EPI == 3;
    constexpr bool BOUT = EPI == 4, MBITS = EPI == 5 || EPI == 6;
    constexpr int EP = (POOL || BOUT) ? 0 : EPI == 5 ? 1 : EPI == 6 ? 2 : EPI;
    const ConvParams& p = pp.c;
    constexpr int NW = NWM * NWN;
    constexpr int TAPS = KS * KS, HALO = KS / 2, BM = 32 * NWM * MT;
    constexpr int WTOT = TAPS * BM * 2;                  // 16-byte vectors of one weight chunk
    constexpr int XTOT = PMAX * 2;
    static_assert(WTOT % 64 == 0, "weight chunk must be whole wave-loads");
    constexpr int WG = WTOT / 64;                        // wave-loads of one weight chunk
    constexpr int WGS = WS ? 0 : WG;                     // ... that travel with every stage
    constexpr int NG = WGS + (XTOT + 63) / 64;           // 1 KB wave-loads per stage
    constexpr int SIDE_E = NG * 64 * 8;                  // after the wave-loads (whose last, partial one zero-fills up to here):
    constexpr int STAGE_E = SIDE_E + 256;                //   float side[2][64] = out_scale[n], bias; elements per stage buffer
    constexpr int NOPS = NG + 2;                         // DMA operations per stage: the wave-loads + the two side loads
    constexpr int LPW_HI = (NOPS + NW - 1) / NW, LPW_LO = NOPS / NW;
    constexpr int KG = LPW_HI;
    constexpr int PA = MT * NJ * 2;                      // 16-byte vectors of the tile's mask (pooled residual) per lane
    constexpr int PCNT = EP == 0 ? NJ : MBITS ? MT * NJ + (EP - 1) * PA : EP * PA;       // prefetch loads per wave and tile
    constexpr int SCNT = POOL ? MT * NJ : 2 * MT * NJ + (BOUT ? (MT == 2 ? NJ : NJ / 2) : 0);    // stores per wave and tile (pooled: NJ / 2 * MT vectors + as many mask words)
    static_assert(!BOUT || MT == 2 || NJ % 2 == 0, "sign-bit words of the 32-channel tile leave in pixel pairs");
    constexpr int PF = NCH >= 2 ? NCH - 2 : 0;           // chunk at whose start the epilogue operands are requested
    constexpr int TCONS = (NSTAGE - 1 + NCH - 1) / NCH;  // first tiles: the conservative wait count (operations of the prologue)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sWfix = (bf16_t*)smem_raw;                   // WS: [NCH][WTOT * 8]
    bf16_t* sS = sWfix + (WS ? NCH * WTOT * 8 : 0);      // [NSTAGE][STAGE_E]
    float* red = (float*)(sS + (size_t)NSTAGE * STAGE_E);                  // [NW][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int PW = p.TW + 2 * HALO;
    const int P = (p.TH + 2 * HALO) * PW;

    // ---- this block's tiles: XCD x sweeps the band [x * band, (x+1) * band) with all of its blocks side by side ----
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, nPer = gridDim.x >> 3;
    const int bandEnd = (xcd + 1) * pp.band < p.pixTiles ? (xcd + 1) * pp.band : p.pixTiles;
    const int tFirst = xcd * pp.band + kx;
    const int mTiles = tFirst < bandEnd ? (bandEnd - tFirst + nPer - 1) / nPer : 0;
    if (mTiles == 0) return;
    const bool hiWave = wave < (NOPS % NW == 0 ? NW : NOPS % NW);   // this wave issues LPW_HI (else LPW_LO) DMA loads per stage

    // ---- chunk-invariant DMA geometry of this lane's wave-loads ----
    int goff[KG];         // weights: byte offset at chunk 0 (or OOB); activations: byte offset relative to the tile's first pixel
    int gpos[KG];         // activations: (patch row << 16) | patch column, -1 = beyond the patch
#pragma unroll
    for (int k = 0; k < KG; k++) {
        const int g64 = wave + k * NW;
        const int v = g64 * 64 + lane;
        goff[k] = PIPE_OOB; gpos[k] = -1;
        if (g64 < WGS) {
            const int row = v >> 1, half = (v & 1) ^ ((row >> 3) & 1);
            const int tap = row / BM, co = row - tap * BM;
            if (co < p.Cout) goff[k] = ((co * TAPS + tap) * p.Cin + half * 8) * 2;
        } else if (g64 < NG) {
            const int vx = v - WGS * 64;
            const int pix = vx >> 1, half = (vx & 1) ^ ((pix >> 3) & 1);
            if (pix < P) {
                const int pr = pix / PW, pc = pix - pr * PW;
                gpos[k] = (pr << 16) | pc;
                goff[k] = (((pr - HALO) * p.W + (pc - HALO)) * p.Cin + half * 8) * 2;
            }
        }
    }
    const int imgBytes = p.H * p.W * p.Cin * 2;
    const int wBytes = p.Cout * TAPS * p.Cin * 2;

    auto issue = [&](int ord, int ich, int buf) {
        const bool live = ord < mTiles;
        const int pt = tFirst + (live ? ord : 0) * nPer;
        const int tw = pt & ((1 << pp.tilesWl2) - 1), th = (pt >> pp.tilesWl2) & ((1 << pp.tilesHl2) - 1), n0 = pt >> (pp.tilesWl2 + pp.tilesHl2);
        const int h0 = th * p.TH, w0 = tw * p.TW;
        const bool interior = h0 >= HALO && w0 >= HALO && h0 + p.TH + HALO <= p.H && w0 + p.TW + HALO <= p.W;
        const __amdgpu_buffer_rsrc_t xRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n0 * p.H * p.W * p.Cin), 0, live ? imgBytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t wRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)n0 * pp.wImgStride), 0, live ? wBytes : 0, 0x00020000);
        // (the chunk's byte offset is made opaque: as a compile-time constant it would be folded into NCH x KG loop-invariant
        //  per-lane offsets that live in registers across the whole tile loop)
        int chOff = ich * 32;
        asm volatile("" : "+s"(chOff));
        const int tileOrg = (h0 * p.W + w0) * p.Cin * 2 + chOff;
        bf16_t* dst = sS + buf * STAGE_E;
#pragma unroll
        for (int k = 0; k < KG; k++) {
            const int g64 = wave + k * NW;
            if (k < LPW_LO || hiWave) {
                if (g64 >= NG) {
                    // operation NG: out_scale[n0][lane], operation NG + 1: bias[lane] -> side[g64 - NG][lane] (zeros when absent)
                    const bool isB = g64 > NG;
                    const float* src = isB ? p.bias : (p.out_scale ? p.out_scale + (int64_t)n0 * p.Cout : nullptr);
                    const __amdgpu_buffer_rsrc_t sRes = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (live && src) ? p.Cout * 4 : 0, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(sRes, (lds_ptr)(dst + SIDE_E + (g64 - NG) * 128), 4, lane * 4, 0, 0, 0);
                } else if (g64 < WGS) {
                    // (the offset goes through a local: passing an element of a template-sized array straight into the builtin from inside a
                    //  lambda makes this clang drop the kernel's HOST stub without a diagnostic)
                    const int off = (pp.dbg & 4) ? PIPE_OOB : goff[k] + chOff;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wRes, (lds_ptr)(dst + g64 * 64 * 8), 16, off, 0, 0, 0);
                } else {
                    int off = PIPE_OOB;
                    if (gpos[k] >= 0) {
                        const int h = h0 + (gpos[k] >> 16) - HALO, w = w0 + (gpos[k] & 0xffff) - HALO;
                        if ((interior || (h >= 0 && h < p.H && w >= 0 && w < p.W)) && !(pp.dbg & 8)) off = tileOrg + goff[k];
                    }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xRes, (lds_ptr)(dst + g64 * 64 * 8), 16, off, 0, 0, 0);
                }
            }
        }
    };

    // ---- fragment addresses (tile-invariant): swizzled 32-byte rows as in conv2d_fwd_dl_kernel ----
    int bPix[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int q = wn * (32 * NJ) + j * 32 + l31;
        bPix[j] = ((q >> p.twShift) * PW) + (q & (p.TW - 1));
    }
    int aBase[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) aBase[i] = (wm * 32 * MT + i * 32 + l31) * 16 + ((lhi ^ ((l31 >> 3) & 1)) << 3);

    // ---- epilogue operands that live in registers: the tile's lrelu mask / pooled residual (EP 1, 2) or its noise (EP 0) ----
    const int coW = wm * 32 * MT;
    u32x4 preA[(EP >= 1 && !MBITS) ? PA : 1], preB[EP == 2 ? PA : 1];
    uint32_t preM[MBITS ? MT * NJ : 1];                              // MBITS: the mask words of (pixel j, 32-channel block i)
    float nzv[NJ];
    int pixOff[NJ];                                                  // byte offset of the lane's pixel j in an image of y (Cout channels), or OOB
    int cellOff[POOL ? NJ : 1];                                      // POOL: byte offset of the pixel's 2x2 cell in an image of the pooled tensor, or OOB
    auto prefetch = [&](int ord) {
        const int pt = tFirst + ord * nPer;
        const int tw = pt & ((1 << pp.tilesWl2) - 1), th = (pt >> pp.tilesWl2) & ((1 << pp.tilesHl2) - 1), n0 = pt >> (pp.tilesWl2 + pp.tilesHl2);
        const int h0 = th * p.TH, w0 = tw * p.TW;
        const int outImg = p.H * p.W * p.Cout * 2;
        int hw2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int q = wn * (32 * NJ) + j * 32 + l31;
            const int h = h0 + (q >> p.twShift), w = w0 + (q & (p.TW - 1));
            const bool valid = h < p.H && w < p.W;
            pixOff[j] = valid ? (h * p.W + w) * pp.yPix * 2 : PIPE_OOB;
            hw2[j] = valid ? ((h >> 1) * (p.W >> 1) + (w >> 1)) * p.Cout * 2 : PIPE_OOB;
            if (POOL) cellOff[j] = hw2[j];
            if (EP == 0) {
                const __amdgpu_buffer_rsrc_t nRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.noise + (int64_t)n0 * p.H * p.W), 0, p.noise ? p.H * p.W * 4 : 0, 0x00020000);
                nzv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(nRes, valid ? (h * p.W + w) * 4 : PIPE_OOB, 0, 0));
            }
        }
        if (EP >= 1) {
            const __amdgpu_buffer_rsrc_t mRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask_y + (int64_t)n0 * p.H * p.W * p.Cout), 0, p.mask_y ? outImg : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res_pooled + (int64_t)n0 * (p.H >> 1) * (p.W >> 1) * p.Cout), 0, p.res_pooled ? outImg >> 2 : 0, 0x00020000);
            if (MBITS) {
                // (a pixel's y is Cout * 2 bytes, its mask words Cout / 8 bytes: the word offset is the pixel offset / 16)
                const __amdgpu_buffer_rsrc_t bRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask_bits + (int64_t)n0 * p.H * p.W * (p.Cout >> 5)), 0, outImg >> 4, 0x00020000);
#pragma unroll
                for (int s = 0; s < MT * NJ; s++) {
                    const int j = s / MT, i = s % MT;
                    const int off = (pixOff[j] != PIPE_OOB && coW + i * 32 < p.Cout) ? (pixOff[j] >> 4) + ((coW + i * 32) >> 3) : PIPE_OOB;
                    preM[s] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(bRes, off, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < PA; s++) {                           // slot s = (j, i, q): this lane's post-swap vector
                const int j = s / (MT * 2), i = (s / 2) % MT, q = s & 1;
                const int cb = coW + i * 32 + (2 * q + lhi) * 8;
                const int offA = cb < p.Cout ? pixOff[j] + cb * 2 : PIPE_OOB, offB = cb < p.Cout ? hw2[j] + cb * 2 : PIPE_OOB;
                if (!MBITS) preA[s] = __builtin_amdgcn_raw_buffer_load_b128(mRes, offA, 0, 0);
                if (EP == 2) preB[s] = __builtin_amdgcn_raw_buffer_load_b128(rRes, offB, 0, 0);
            }
        }
    };

    float msumAcc = 0.f;                                             // EP >= 1: this lane's share of the masked channel sums (see the flush)
    int msumIdx = 0;

    f32x16 acc[MT][NJ];
    auto epilogue = [&](int ord, const bf16_t* stage) {
        const int pt = tFirst + ord * nPer;
        const int n0 = pt >> (pp.tilesWl2 + pp.tilesHl2);
        const __amdgpu_buffer_rsrc_t yRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (int64_t)n0 * p.H * p.W * pp.yPix), 0, p.H * p.W * pp.yPix * 2, 0x00020000);
        const float* side = (const float*)(stage + SIDE_E);         // [2][64]: out_scale[n0], bias (landed with this stage)
        if ((pp.dbg & 3) == 1) return;
        if ((pp.dbg & 3) == 2) {
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int cb = coW + i * 32 + (2 * q + lhi) * 8;
                        const int offY = cb < p.Cout ? pixOff[j] + cb * 2 : PIPE_OOB;
                        u32x4 val = {0u, 0u, 0u, 0u};
                        __builtin_amdgcn_raw_buffer_store_b128(val, yRes, offY, 0, 0);
                    }
            return;
        }
        if (POOL) {
            // lane pixels j, j + 1 are rows h, h + 1 of one column (TW == 32), lane ^ 1 holds the neighbouring column: a 2x2 cell is two
            // registers here and two there.  Rounded to bf16 first and summed as agf_pool2x2 does, ((a + b) + c) + d: bit-identical to
            // conv -> pool2x2 without the full-resolution write and re-read.  Stores per wave and tile: NJ / 2 * MT vectors + as many mask
            // words (SCNT, which the counted waits use).
            const int cells = (p.H >> 1) * (p.W >> 1);
            const __amdgpu_buffer_rsrc_t pRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (int64_t)n0 * cells * p.Cout), 0, cells * p.Cout * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t kRes = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pool_mask + (int64_t)n0 * cells * (p.Cout >> 3)), 0, cells * (p.Cout >> 3) * 4, 0x00020000);
            auto value = [&](int j, int i, int q) -> u32x4 {
                uint32_t Pk[2][2];
#pragma unroll
                for (int r2 = 0; r2 < 2; r2++) {
                    const int rg = 2 * q + r2;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = acc[i][j][rg * 4 + e];
                    const int co = coW + i * 32 + rg * 8 + lhi * 4;
                    const f32x4 bb = *(const f32x4*)(side + 64 + co);
                    if (p.out_scale) {
                        const f32x4 os = *(const f32x4*)(side + co);
                        v[0] *= os.x; v[1] *= os.y; v[2] *= os.z; v[3] *= os.w;
                    }
                    v[0] += bb.x + nzv[j]; v[1] += bb.y + nzv[j]; v[2] += bb.z + nzv[j]; v[3] += bb.w + nzv[j];
                    if (p.act == 3) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] *= p.gain;
                    Pk[r2][0] = Pack16<bf16_t>::pack(v[0], v[1]);
                    Pk[r2][1] = Pack16<bf16_t>::pack(v[2], v[3]);
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(Pk[0][0], Pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(Pk[0][1], Pk[1][1], false, false);
                return u32x4{s0[0], s1[0], s0[1], s1[1]};
            };
            auto unpack8 = [](u32x4 t, float (&f)[8]) {
                Pack16<bf16_t>::unpack(t.x, f[0], f[1]); Pack16<bf16_t>::unpack(t.y, f[2], f[3]);
                Pack16<bf16_t>::unpack(t.z, f[4], f[5]); Pack16<bf16_t>::unpack(t.w, f[6], f[7]);
            };
            auto bits8 = [](const float (&f)[8]) {
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 8; e++) m |= (f[e] > 0.f ? 1u : 0u) << e;
                return m;
            };
#pragma unroll
            for (int jp = 0; jp < NJ / 2; jp++) {
                const int j0 = 2 * jp;
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    // the lanes of a pair share the work: the even lane (column w) pools channel group 2*0 + lhi, the odd one (column w + 1) group
                    // 2*1 + lhi; each sends the other the two vectors it does not pool itself (DPP) and every lane stores
                    const bool odd = (l31 & 1) != 0;
                    const u32x4 a0 = value(j0, i, 0), a1 = value(j0, i, 1), c0 = value(j0 + 1, i, 0), c1 = value(j0 + 1, i, 1);
                    u32x4 mineA, mineC, recvA, recvC;
                    mineA.x = odd ? a1.x : a0.x; mineA.y = odd ? a1.y : a0.y; mineA.z = odd ? a1.z : a0.z; mineA.w = odd ? a1.w : a0.w;
                    mineC.x = odd ? c1.x : c0.x; mineC.y = odd ? c1.y : c0.y; mineC.z = odd ? c1.z : c0.z; mineC.w = odd ? c1.w : c0.w;
                    recvA.x = agf_swap1(odd ? a0.x : a1.x); recvA.y = agf_swap1(odd ? a0.y : a1.y); recvA.z = agf_swap1(odd ? a0.z : a1.z); recvA.w = agf_swap1(odd ? a0.w : a1.w);
                    recvC.x = agf_swap1(odd ? c0.x : c1.x); recvC.y = agf_swap1(odd ? c0.y : c1.y); recvC.z = agf_swap1(odd ? c0.z : c1.z); recvC.w = agf_swap1(odd ? c0.w : c1.w);
                    // cell = (a b / c d) with a, c in the even column: the even lane owns a, c; the odd lane owns b, d.  Sum in agf_pool2x2's order
                    float fa[8], fb[8], fc[8], fd[8], o[8];
                    unpack8(mineA, fa); unpack8(recvA, fb); unpack8(mineC, fc); unpack8(recvC, fd);
            #pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float s = fa[e] + fb[e];                          // a + b (commutative: same bits on both lanes)
                        const float c = odd ? fd[e] : fc[e], d = odd ? fc[e] : fd[e];
                        o[e] = ((s + c) + d) * p.pool_gain;
                    }
                    const unsigned bA = bits8(fa), bB = bits8(fb), bC = bits8(fc), bD = bits8(fd);
                    const unsigned word = odd ? (bB | (bA << 8) | (bD << 16) | (bC << 24)) : (bA | (bB << 8) | (bC << 16) | (bD << 24));
                    const int cb = coW + i * 32 + (2 * (odd ? 1 : 0) + lhi) * 8;
                    {
                        const bool keep = cellOff[j0] != PIPE_OOB && cb < p.Cout;
                        u32x4 out;
                        out.x = Pack16<bf16_t>::pack(o[0], o[1]); out.y = Pack16<bf16_t>::pack(o[2], o[3]);
                        out.z = Pack16<bf16_t>::pack(o[4], o[5]); out.w = Pack16<bf16_t>::pack(o[6], o[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(out, pRes, keep ? cellOff[j0] + cb * 2 : PIPE_OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(word, kRes, keep ? (cellOff[j0] >> 2) + (cb >> 1) : PIPE_OOB, 0, 0);
                    }
                }
            }
            return;
        }
        float msum[EP >= 1 ? MT * 16 : 1];
        if (EP >= 1) {
#pragma unroll
            for (int e = 0; e < MT * 16; e++) msum[e] = 0.f;
        }
        // BOUT: the sign-bit words go to [pixel][yPix / 32] dwords: a pixel's words start at its y offset / 16 (yPix * 2 bytes of y per pixel,
        // yPix / 8 bytes of bits), this launch's first block at word bitsOrg (a channel slice of a wider tensor: the 128-channel split)
        const __amdgpu_buffer_rsrc_t oRes = __builtin_amdgcn_make_buffer_rsrc((void*)((BOUT && p.bits_out) ? p.bits_out + (int64_t)n0 * p.H * p.W * (pp.yPix >> 5) : nullptr), 0,
                                                                              (BOUT && p.bits_out) ? (p.H * p.W * pp.yPix) >> 3 : 0, 0x00020000);
        uint32_t wprev = 0u;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            uint32_t wbits[MT];
#pragma unroll
            for (int i = 0; i < MT; i++) wbits[i] = 0u;
#pragma unroll
            for (int i = 0; i < MT; i++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    uint32_t Pk[2][2];
#pragma unroll
                    for (int r2 = 0; r2 < 2; r2++) {
                        const int rg = 2 * q + r2;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = acc[i][j][rg * 4 + e];
                        if (EP == 0) {
                            const int co = coW + i * 32 + rg * 8 + lhi * 4;
                            const f32x4 bb = *(const f32x4*)(side + 64 + co);
                            if (p.out_scale) {
                                const f32x4 os = *(const f32x4*)(side + co);
                                v[0] *= os.x; v[1] *= os.y; v[2] *= os.z; v[3] *= os.w;
                            }
                            v[0] += bb.x + nzv[j]; v[1] += bb.y + nzv[j]; v[2] += bb.z + nzv[j]; v[3] += bb.w + nzv[j];
                        }
                        if (p.act == 3) {
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                        }
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] *= p.gain;
                        Pk[r2][0] = Pack16<bf16_t>::pack(v[0], v[1]);
                        Pk[r2][1] = Pack16<bf16_t>::pack(v[2], v[3]);
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(Pk[0][0], Pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(Pk[0][1], Pk[1][1], false, false);
                    u32x4 val = {s0[0], s1[0], s0[1], s1[1]};
                    const int cb = coW + i * 32 + (2 * q + lhi) * 8;
                    if (EP >= 1) {
                        const int s = (j * MT + i) * 2 + q;
                        float g[8];
                        Pack16<bf16_t>::unpack(val.x, g[0], g[1]); Pack16<bf16_t>::unpack(val.y, g[2], g[3]);
                        Pack16<bf16_t>::unpack(val.z, g[4], g[5]); Pack16<bf16_t>::unpack(val.w, g[6], g[7]);
                        if (EP == 2) {   // pooled residual (zeros when absent)
                            float rv[8];
                            Pack16<bf16_t>::unpack(preB[s].x, rv[0], rv[1]); Pack16<bf16_t>::unpack(preB[s].y, rv[2], rv[3]);
                            Pack16<bf16_t>::unpack(preB[s].z, rv[4], rv[5]); Pack16<bf16_t>::unpack(preB[s].w, rv[6], rv[7]);
#pragma unroll
                            for (int e = 0; e < 8; e++) g[e] += rv[e] * p.res_scale;
                        }
                        if constexpr (MBITS) {
                            if (p.mask_bits) {
                                const uint32_t byte = preM[j * MT + i] >> (8 * (2 * q + lhi));
                                const bool live = pixOff[j] != PIPE_OOB && cb < p.Cout;
#pragma unroll
                                for (int e = 0; e < 8; e++) {
                                    g[e] = ((byte >> e) & 1u) ? g[e] : g[e] * p.mask_alpha;
                                    msum[(i * 2 + q) * 8 + e] += live ? g[e] : 0.f;
                                }
                            }
                        } else if (p.mask_y) {
                            float a[8];
                            Pack16<bf16_t>::unpack(preA[s].x, a[0], a[1]); Pack16<bf16_t>::unpack(preA[s].y, a[2], a[3]);
                            Pack16<bf16_t>::unpack(preA[s].z, a[4], a[5]); Pack16<bf16_t>::unpack(preA[s].w, a[6], a[7]);
                            const bool live = pixOff[j] != PIPE_OOB && cb < p.Cout;
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                g[e] = a[e] > 0.f ? g[e] : g[e] * p.mask_alpha;
                                msum[(i * 2 + q) * 8 + e] += live ? g[e] : 0.f;
                            }
                        }
                        val.x = Pack16<bf16_t>::pack(g[0], g[1]); val.y = Pack16<bf16_t>::pack(g[2], g[3]);
                        val.z = Pack16<bf16_t>::pack(g[4], g[5]); val.w = Pack16<bf16_t>::pack(g[6], g[7]);
                    }
                    const int offY = cb < p.Cout ? pixOff[j] + cb * 2 : PIPE_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(val, yRes, offY, 0, 0);
                    if constexpr (BOUT) {
                        uint32_t byte = 0u;
                        byte |= ((int16_t)(val.x & 0xffffu) > 0 ? 1u : 0u) | ((int16_t)(val.x >> 16) > 0 ? 2u : 0u);
                        byte |= ((int16_t)(val.y & 0xffffu) > 0 ? 4u : 0u) | ((int16_t)(val.y >> 16) > 0 ? 8u : 0u);
                        byte |= ((int16_t)(val.z & 0xffffu) > 0 ? 16u : 0u) | ((int16_t)(val.z >> 16) > 0 ? 32u : 0u);
                        byte |= ((int16_t)(val.w & 0xffffu) > 0 ? 64u : 0u) | ((int16_t)(val.w >> 16) > 0 ? 128u : 0u);
                        wbits[i] |= byte << (8 * (2 * q + lhi));
                    }
                }
            }
            if constexpr (BOUT) {
                // two bytes of a (pixel, 32-channel block) word are here, two in lane ^ 32 (see conv_epilogue_pl): one swap + OR assembles the
                // words of blocks 0 / 1 (MT == 2) or of pixels j - 1 / j (MT == 1); lower half-wave stores the first, upper the second
                if constexpr (MT == 2) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(wbits[0], wbits[1], false, false);
                    const int cb32 = coW + lhi * 32;
                    const int off = (pixOff[j] != PIPE_OOB && cb32 < p.Cout) ? (pixOff[j] >> 4) + (cb32 >> 3) : PIPE_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(sw[0] | sw[1], oRes, off, 0, 0);
                } else if (j & 1) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(wprev, wbits[0], false, false);
                    const int po = lhi ? pixOff[j] : pixOff[j - 1];
                    const int off = (po != PIPE_OOB && coW < p.Cout) ? (po >> 4) + (coW >> 3) : PIPE_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(sw[0] | sw[1], oRes, off, 0, 0);
                } else wprev = wbits[0];
            }
        }
        if (EP >= 1) {
            // halving butterfly over the 32 lanes of each half-wave (see conv_epilogue_pl): lane l keeps the total of value index
            // bit0*NV/2 + bit1*NV/4 + ... ; accumulated over the block's tiles in ONE register
            constexpr int NV = MT * 16;
            int live = NV, idx = 0, m = 1;
#pragma unroll
            for (; live > 1; live >>= 1, m <<= 1) {
                const bool up = (lane & m) != 0;
                const int half = live >> 1;
#pragma unroll
                for (int k = 0; k < NV / 2; k++) {
                    if (k < half) {
                        const float send = up ? msum[k] : msum[k + half];
                        const float keep = up ? msum[k + half] : msum[k];
                        msum[k] = keep + __shfl_xor(send, m);
                    }
                }
                idx += up ? half : 0;
            }
#pragma unroll
            for (; m < 32; m <<= 1) msum[0] += __shfl_xor(msum[0], m);
            msumAcc += msum[0];
            msumIdx = idx;
        }
    };

    if (WS) {
        // the whole weight tensor, chunk-major, in the stage layout of a weight chunk (swizzled 32-byte rows)
        const __amdgpu_buffer_rsrc_t wAll = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wBytes, 0x00020000);
        for (int g = wave; g < WG * NCH; g += NW) {
            const int chunk = g / WG, gg = g - chunk * WG;
            const int v = gg * 64 + lane;
            const int row = v >> 1, half = (v & 1) ^ ((row >> 3) & 1);
            const int tap = row / BM, co = row - tap * BM;
            const int off = co < p.Cout ? ((co * TAPS + tap) * p.Cin + half * 8) * 2 + chunk * 32 : PIPE_OOB;
            bf16_t* dst = sWfix + (chunk * WTOT + gg * 64) * 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wAll, (lds_ptr)dst, 16, off, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- the pipeline ----
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; s++) issue(s / NCH, s % NCH, s);
    __builtin_amdgcn_sched_barrier(0);
    int buf = 0;
    for (int t = 0; t < mTiles; t++) {
        pipe_static_for(std::make_integer_sequence<int, NCH>{}, [&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            // wait for this wave's loads of stage (t, ch), then for everyone's; after the barrier buffer (buf - 1) is free again
            if (t < TCONS) { if (hiWave) pipe_wait_vm<(NSTAGE - 2) * LPW_HI>(); else pipe_wait_vm<(NSTAGE - 2) * LPW_LO>(); }
            // (stores are NOT counted among the operations that may stay in flight: loads retire in order among themselves, but the ISA does
            //  not promise that a younger store cannot be acknowledged before an older load returns -- counting the tile's 2*MT*NJ stores as
            //  "still outstanding" would let the wait pass with the stage's loads pending.  With S = 0 the wait is exact when no store is
            //  pending and merely earlier-than-needed (it also drains the stores) otherwise.  AGF_PIPE_COUNT_STORES=1: the old count, A/B.)
            else if (pp.countStores) {
                if (hiWave) pipe_wait_vm<pipe_younger<NCH, NSTAGE>(ch, LPW_HI, PCNT, SCNT, PF)>();
                else pipe_wait_vm<pipe_younger<NCH, NSTAGE>(ch, LPW_LO, PCNT, SCNT, PF)>();
            } else if (hiWave) pipe_wait_vm<pipe_younger<NCH, NSTAGE>(ch, LPW_HI, PCNT, 0, PF)>();
            else pipe_wait_vm<pipe_younger<NCH, NSTAGE>(ch, LPW_LO, PCNT, 0, PF)>();
            // a bare s_barrier: __syncthreads() adds a workgroup fence that the compiler lowers to `s_waitcnt vmcnt(0)` -- a full drain of
            // the pipeline.  Nothing more is needed here: a wave's DMA data is in LDS once ITS vmcnt says so (the wait above), its
            // fragment reads of the previous stage were consumed by MFMAs before it arrived.
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (ch == 0) {
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            }
            const bf16_t* stageBase = sS + buf * STAGE_E;
            const bf16_t* cW = WS ? sWfix + ch * WTOT * 8 : stageBase;
            const bf16_t* cX = stageBase + WGS * 64 * 8;
#pragma unroll
            for (int kh = 0; kh < KS; kh++) {
#pragma unroll
                for (int kw = 0; kw < KS; kw++) {
                    const int tap = kh * KS + kw;
                    bf16x8 af[MT], bfr[NJ];
#pragma unroll
                    for (int i = 0; i < MT; i++) af[i] = *(const bf16x8*)(cW + tap * BM * 16 + aBase[i]);
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        const int pix = bPix[j] + kh * PW + kw;
                        bfr[j] = *(const bf16x8*)(cX + pix * 16 + ((lhi ^ ((pix >> 3) & 1)) << 3));
                    }
#pragma unroll
                    for (int i = 0; i < MT; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    if (tap == PIPE_ISSUE_TAP) {
                        // the address arithmetic of the epilogue-operand requests and of the next DMA group rides in the shadow of this
                        // chunk's MFMAs (issued in one piece right after the barrier, all eight waves spent it in lock step with the
                        // matrix pipe idle); the order [prefetch][DMA] within the iteration is what pipe_younger counts on
                        __builtin_amdgcn_sched_barrier(0);
                        if (ch == PF) prefetch(t);
                        __builtin_amdgcn_sched_barrier(0);
                        constexpr int ahead = NSTAGE - 1;
                        const int nb = buf + ahead >= NSTAGE ? buf + ahead - NSTAGE : buf + ahead;
                        issue(t + (ch + ahead) / NCH, (ch + ahead) % NCH, nb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            buf = buf + 1 == NSTAGE ? 0 : buf + 1;
            asm volatile("" : "+s"(buf));                 // opaque: keeps the next chunks' LDS addresses from being formed (and held) early
            __builtin_amdgcn_sched_barrier(0);
            if (ch == NCH - 1) epilogue(t, stageBase);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    if (EP >= 1 && (MBITS ? p.mask_bits != nullptr : p.mask_y != nullptr) && p.mask_sum) {             // block-uniform
        // add the waves that share the channels through LDS, then ONE atomic per channel and block
        red[wave * 64 + lane] = msumAcc;
        __syncthreads();
        const int ci = msumIdx >> 3, e = msumIdx & 7;                      // value index -> (i * 2 + q, e)
        const int co = coW + (ci >> 1) * 32 + (2 * (ci & 1) + lhi) * 8 + e;
        if (wn == 0 && (MT == 2 || l31 < 16) && co < p.Cout) {
            float v = 0.f;
            for (int k = 0; k < NWN; k++) v += red[(wm * NWN + k) * 64 + lane];
            unsafeAtomicAdd(p.mask_sum + (int64_t)(blockIdx.x & 255) * p.Cout + co, v);
        }
    }
}

template <int KS, int MT, int NWM, int NWN, int NJ, int NCH, int NSTAGE, int PMAX, int EPI, bool WS>
static int launch_pipe_e(const PipeParams& pp, int blocksPerCU, hipStream_t st) {
    constexpr int TAPS = KS * KS, BM = 32 * NWM * MT, NW = NWM * NWN;
    constexpr int WG = TAPS * BM * 2 / 64;
    constexpr int NG = (WS ? 0 : WG) + (PMAX * 2 + 63) / 64;
    const size_t lds = (size_t)(WS ? NCH * WG * 1024 : 0) + (size_t)NSTAGE * (NG * 1024 + 512) + NW * 64 * 4;
    if (lds > 160 * 1024) return AGF_ENOKERNEL;
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return AGF_ELAUNCH; cus = prop.multiProcessorCount; }
    int grid = (cus * blocksPerCU) & ~7;
    if (grid < 8) grid = 8;
    hipError_t e = hipFuncSetAttribute((const void*)conv2d_fwd_pipe_kernel<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, EPI, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { agf_set_error("conv2d_fwd (pipe): cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return AGF_ELAUNCH; }
    hipLaunchKernelGGL((conv2d_fwd_pipe_kernel<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, EPI, WS>), dim3((unsigned)grid), dim3(64 * NW), lds, st, pp);
    return AGF_OK;
}

template <int KS, int MT, int NWM, int NWN, int NJ, int NCH, int NSTAGE, int PMAX, bool WS = false>
static int launch_pipe(const PipeParams& pp, int blocksPerCU, hipStream_t st) {
    if (pp.c.pool_mask) {
        // conv + lrelu + 2x2 average + sign mask: instantiated for the one shape class the networks pool after on this kernel (the
        // discriminator's 64 -> 64 conv on the 256 x 256 / 128 x 128 maps: 33..64 output channels, Cin 64, shared weights)
        if constexpr (MT == 2 && NCH == 4 && WS && NJ % 2 == 0) return launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 3, WS>(pp, blocksPerCU, st);
        else return AGF_ENOKERNEL;
    }
    if (pp.c.mask_bits) {
        // the mask as bits: instantiated for the shapes the discriminator's hand-offs produce on this kernel (shared weights, Cin <= 64)
        if constexpr (WS) return pp.c.res_pooled ? launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 6, WS>(pp, blocksPerCU, st)
                                                 : launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 5, WS>(pp, blocksPerCU, st);
        else return AGF_ENOKERNEL;
    }
    if (pp.c.bits_out) {
        if constexpr (WS) return launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 4, WS>(pp, blocksPerCU, st);
        else return AGF_ENOKERNEL;
    }
    if (pp.c.res_pooled) return launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 2, WS>(pp, blocksPerCU, st);
    if (pp.c.mask_y)     return launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 1, WS>(pp, blocksPerCU, st);
    return launch_pipe_e<KS, MT, NWM, NWN, NJ, NCH, NSTAGE, PMAX, 0, WS>(pp, blocksPerCU, st);
}

static bool pipe_covers(int N, int H, int W, int Cin, int Cout) {
    if (Cout > 64 || (Cout % 8) || (Cin != 32 && Cin != 64 && Cin != 128) || H < 16 || W < 32) return false;
    const int tw = (W + 31) / 32, th = (H + 15) / 16;
    if ((tw & (tw - 1)) || (th & (th - 1)) || (int64_t)tw * th * N < 512) return false;
    return (int64_t)H * W * (Cin > Cout ? Cin : Cout) * 2 < 0x60000000ll;
}

static int pipe_launch(const ConvParams& p0, int64_t wImgStride, hipStream_t st, int yPix = 0);

int agf_conv2d_pipe_launch(const ConvParams& p0, hipStream_t st) {
    // 128 output channels from 32 / 64 inputs on a streaming-size map (the discriminator's 64 -> 128 conv at 128x128): the 8-wave
    // 128-channel kernel has only 2-4 K chunks per tile to amortise its prologue and epilogue over (590 TFLOP/s); two launches of this
    // kernel, one per 64-channel half of the weights, each writing its channel slice of y, are faster (the second reads x from L2 / MALL)
    constexpr int split = 1;
    if (split && !p0.pool_mask && p0.Cout == 128 && (p0.Cin == 32 || p0.Cin == 64) && !p0.mask_y && !p0.mask_bits && !p0.res_pooled && !p0.out_scale && !p0.in_scale && !p0.residual &&
        !p0.noise && ((uintptr_t)p0.y % 16) == 0 && pipe_covers(p0.N, p0.H, p0.W, p0.Cin, 64)) {
        for (int half = 0; half < 2; half++) {
            ConvParams q = p0;
            q.Cout = 64;
            q.w = p0.w + (int64_t)half * 64 * 9 * p0.Cin;
            q.y = p0.y + half * 64;
            q.bias = p0.bias ? p0.bias + half * 64 : nullptr;
            q.bits_out = p0.bits_out ? p0.bits_out + half * 2 : nullptr;      // (the half's two words of every pixel's four)
            const int rc = pipe_launch(q, 0, st, 128);
            if (rc != AGF_OK) return half == 0 ? rc : AGF_ELAUNCH;
        }
        return AGF_OK;
    }
    return pipe_launch(p0, 0, st);
}

// Whether a producer conv (Cin -> Cout, 3x3) may hand its consumer the lrelu mask as bits without landing on a slower kernel: every kernel
// family writes / reads them except this file's re-streamed-weight instantiations (Cin = 128 with <= 64 outputs per block)
extern "C" int agf_conv2d_maskbits_covers(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (Cout % 32) return 0;
    if (Cin > 64 && Cout <= 64 && pipe_covers(N, H, W, Cin, Cout)) return 0;
    return 1;
}

// wmod[n][co][tap][ci] = w[co][tap][ci] * s[n][ci]: the style modulation folded into one weight tensor per image (what the reference
// materialises as `weight * style`, implementations/StyleGAN2/model.py:115 -- here only for the few-channel layers, a few MB)
__global__ void __launch_bounds__(256) modulate_weights_kernel(const bf16_t* w, const float* s, bf16_t* out, int N, int rows, int Cin) {
    const int vpr = Cin >> 3;
    const int64_t total = (int64_t)N * rows * vpr;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int cv = (int)(v % vpr);
        const int64_t r = v / vpr;
        const int row = (int)(r % rows), n = (int)(r / rows);
        const u32x4 val = *(const u32x4*)(w + ((int64_t)row * Cin + cv * 8));
        *(u32x4*)(out + (((int64_t)n * rows + row) * Cin + cv * 8)) = scale_vec8(val, s + (int64_t)n * Cin + cv * 8);
    }
}

extern "C" int agf_modulate_weights(const void* w, const float* s, void* wmod, int dtype, int32_t N, int32_t Cout, int32_t taps, int32_t Cin, void* stream) {
    AGF_CHECK(w && s && wmod, "modulate_weights: null pointer");
    AGF_CHECK(dtype == AGF_BF16, "modulate_weights: bf16 only");
    AGF_CHECK(N >= 1 && Cout >= 1 && taps >= 1 && Cin >= 8 && Cin % 8 == 0, "modulate_weights: Cin must be a positive multiple of 8");
    AGF_CHECK(((uintptr_t)w % 16) == 0 && ((uintptr_t)wmod % 16) == 0 && ((uintptr_t)s % 16) == 0, "modulate_weights: misaligned pointer");
    const int64_t total = (int64_t)N * Cout * taps * (Cin / 8);
    int64_t blocks = agf_ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(modulate_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, s, (bf16_t*)wmod, N, Cout * taps, Cin);
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

extern "C" int agf_conv2d_fwd_wimg_covers(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize) {
    constexpr int mode = 1;
    return (mode && ksize == 3 && pipe_covers(N, H, W, Cin, Cout)) ? 1 : 0;
}

extern "C" int agf_conv2d_fwd_wimg(const void* x, const void* w, void* y, const float* out_scale, const float* bias, const float* noise,
                                   int dtype, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                                   int act, float alpha, float act_gain, int64_t w_image_stride, void* stream) {
    AGF_CHECK(x && w && y, "conv2d_fwd_wimg: null pointer");
    AGF_CHECK(dtype == AGF_BF16, "conv2d_fwd_wimg: bf16 only");
    AGF_CHECK(act == 1 || act == 3, "conv2d_fwd_wimg: act must be 1 (linear) or 3 (lrelu)");
    AGF_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "conv2d_fwd_wimg: misaligned pointer");
    AGF_CHECK(w_image_stride >= (int64_t)Cout * ksize * ksize * Cin && (w_image_stride % 8) == 0, "conv2d_fwd_wimg: bad per-image weight stride");
    if (ksize != 3 || !pipe_covers(N, H, W, Cin, Cout)) { agf_set_error("conv2d_fwd_wimg: shape not covered by the per-image-weight kernel"); return AGF_ENOKERNEL; }
    ConvParams p = {};
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.y = (bf16_t*)y;
    p.out_scale = out_scale; p.bias = bias; p.noise = noise;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.act = act; p.alpha = alpha; p.gain = act_gain;
    const int rc = pipe_launch(p, w_image_stride, (hipStream_t)stream);
    if (rc != AGF_OK) return rc;
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

static int pipe_launch(const ConvParams& p0, int64_t wImgStride, hipStream_t st, int yPix) {
    // covered: 3x3, one co tile (Cout <= 64), Cin in {32, 64, 128}, maps that 16x32 pixel tiles cover with power-of-two tile counts,
    // no input scale (style-modulated layers come with per-image weights instead), no residual operand
    constexpr int mode = 1;
    if (!mode) return AGF_ENOKERNEL;
    ConvParams p = p0;
    if (p.in_scale || p.residual) return AGF_ENOKERNEL;
    if ((p.mask_bits || p.bits_out) && (wImgStride != 0 || p.Cin > 64 || (p.Cout % 32) || p.pool_mask)) return AGF_ENOKERNEL;
    if ((p.mask_y || p.mask_bits || p.res_pooled) && (p.out_scale || p.bias || p.noise)) return AGF_ENOKERNEL;
    if (p.pool_mask && ((p.H & 1) || (p.W & 1) || (p.H % 16) || (p.W % 32) || yPix)) return AGF_ENOKERNEL;      // whole tiles only: every 2x2 cell inside one
    if (((uintptr_t)p.y % 16) || !pipe_covers(p.N, p.H, p.W, p.Cin, p.Cout)) return AGF_ENOKERNEL;
    p.flat = 0; p.TI = 1; p.TW = 32; p.TH = 16; p.twShift = 5; p.thShift = 4;
    p.tilesW = (p.W + 31) / 32; p.tilesH = (p.H + 15) / 16; p.tilesN = p.N; p.tilesCo = 1;
    p.pixTiles = p.tilesW * p.tilesH * p.N;
    PipeParams pp;
    pp.c = p;
    pp.tilesWl2 = 0; while ((1 << pp.tilesWl2) < p.tilesW) pp.tilesWl2++;
    pp.tilesHl2 = 0; while ((1 << pp.tilesHl2) < p.tilesH) pp.tilesHl2++;
    pp.band = (p.pixTiles + 7) / 8;
    pp.wImgStride = wImgStride;
    pp.yPix = yPix ? yPix : p.Cout;
    if (pp.yPix != p.Cout && (p.mask_y || p.mask_bits || p.res_pooled || p.out_scale || wImgStride)) return AGF_ENOKERNEL;
    constexpr int pipe_dbg = 0;
    pp.dbg = pipe_dbg;
    constexpr int cnt_st = 0;
    pp.countStores = cnt_st;
    constexpr int ws_on = 1;
    if (ws_on && wImgStride == 0 && p.Cin <= 64) {       // shared weights that fit next to the activation ring: loaded once per block
        if (p.Cout > 32) {
            if (p.Cin == 32) return launch_pipe<3, 2, 1, 8, 2, 2, 5, 612, true>(pp, 1, st);
            return launch_pipe<3, 2, 1, 8, 2, 4, 4, 612, true>(pp, 1, st);
        }
        if (p.Cin == 32) return launch_pipe<3, 1, 1, 8, 2, 2, 6, 612, true>(pp, 1, st);
        return launch_pipe<3, 1, 1, 8, 2, 4, 5, 612, true>(pp, 1, st);
    }
    if (wImgStride == 0 && p.Cin <= 64) {                // (A/B: AGF_PIPE_WS=0)
        if (p.Cout > 32) {
            if (p.Cin == 32)  return launch_pipe<3, 2, 1, 8, 2, 2, 4, 612>(pp, 1, st);
            return launch_pipe<3, 2, 1, 8, 2, 4, 4, 612>(pp, 1, st);
        }
        if (p.Cin == 32)  return launch_pipe<3, 1, 1, 8, 2, 2, 5, 612>(pp, 1, st);
        return launch_pipe<3, 1, 1, 8, 2, 4, 5, 612>(pp, 1, st);
    }
    if (p.Cout > 32) {
        if (p.Cin == 32)  return launch_pipe<3, 2, 1, 8, 2, 2, 4, 612>(pp, 1, st);
        if (p.Cin == 64)  return launch_pipe<3, 2, 1, 8, 2, 4, 4, 612>(pp, 1, st);
        return launch_pipe<3, 2, 1, 8, 2, 8, 4, 612>(pp, 1, st);
    }
    if (p.Cin == 32)  return launch_pipe<3, 1, 1, 8, 2, 2, 5, 612>(pp, 1, st);
    if (p.Cin == 64)  return launch_pipe<3, 1, 1, 8, 2, 4, 5, 612>(pp, 1, st);
    return launch_pipe<3, 1, 1, 8, 2, 8, 5, 612>(pp, 1, st);
}
