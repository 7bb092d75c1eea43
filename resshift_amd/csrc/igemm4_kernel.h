// Halo-tile implicit GEMM for the 3x3 / stride-1 / pad-1 convolutions (fp16 storage), fourth generation.
//
// igemm2 / igemm3 gather the pixel operand once PER FILTER TAP: a 3x3 conv moves every input pixel nine times from L2 into
// LDS, and the GroupNorm + SiLU in front of the conv (models/unet.py:128-147, ldm/modules/diffusionmodules/model.py:129-137)
// needs its own read + write pass over the tensor because LDS-DMA data never passes through registers.  Here the loop nest
// is turned inside out:
//     for 64-channel chunk c:   halo tile (TH+2) x (TW+2) pixels x 64 channels  -> LDS  ONCE        (LDS-DMA)
//                               [GroupNorm affine (+FiLM) + SiLU applied IN LDS, once per element]   (optional)
//         for tap (ky,kx):      weight tile BC x 64 of W[:, tap, c]            -> LDS ring          (LDS-DMA)
//                               MFMA: the pixel fragments are the SAME halo rows, read at offset ky*(TW+2)+kx
//   * pixel traffic L2 -> LDS drops from 9 x 256 rows to (TH+2)(TW+2) = 396 rows per chunk (5.8 x less at TW = 64); the weight
//     tile is amortised over 256 pixels (igemm2: 128);
//   * the conv reads the RAW producer output: the GroupNorm apply pass (one read + one write of the whole tensor per
//     GroupNorm) disappears, only the statistics pass remains.  The affine is applied exactly where gn_apply_kernel applies it
//     (fp32 fma, SiLU, round to fp16), so results are bit-identical to the two-kernel path; halo rows outside the image stay
//     exact zeros (the conv pads the NORMALISED tensor);
//   * 128-byte LDS rows with the usual (chunk ^ row & 7) swizzle are conflict-free for ANY 16 consecutive rows, so the shifted
//     fragment reads cost no more than the aligned ones.
// Tile: 256 output pixels (TH x TW = 4 x 64 or 8 x 32, one image) x BC channels, 8 waves as 4 pixel-waves x 2 channel-waves
// (wave tile 64 x BC/2), one workgroup per CU; LDS: 2 halo buffers (chunk c computes while chunk c+1 arrives, spread over the
// taps) + 2 weight slots.  Epilogue as igemm2 (bias, activation, residual, fp16 transposition through LDS).
//
// SPLIT = true: the same kernel for split storage (RS_F16S: (hi, lo) fp16 pairs, common.h).  A chunk is 32 channels and an LDS
// row holds [32 ch hi | 32 ch lo] - again 128 bytes / 8 sixteen-byte positions with the same swizzle, so the LDS image, the
// LDS-DMA pieces and every fragment address are those of the fp16 kernel; position q of a row is fetched from the hi (q < 4) or
// the lo plane of the pixel record / weight row.  What were the two k-steps of a stage are now the hi and the lo fragments of ONE
// k-step: acc += (2^11 Wh).Xh + Wh.Xl + Wl.Xh (three MFMAs, ONE accumulator: the hi weight fragment is scaled by 2^11 with
// v_pk_mul_f16 - exact for |w| < 32, checked when the weights are packed - instead of keeping a second accumulator set), and the
// epilogue multiplies by 2^-11.  The GroupNorm pass joins, transforms and re-splits the pairs.
// SEG > 0 (igemm4s.hip): small planes.  The 8 x 32-pixel tile is cut into 32 / SEG segments of SEG pixels, each with its own halo
// columns and separated by ONE shared zero column ([pad | seg 0 | pad | seg 1 | ... | pad]: 1 + (32 / SEG)(SEG + 1) <= 40 columns, the
// halo buffer of the 8 x 32 geometry): SEG = 8: four IMAGES of an 8 x 8 plane side by side; SEG = 16: the two 8-row halves of ONE 16 x 16
// image (each half brings its own top / bottom halo rows).  That puts the 16 x 16 / 8 x 8 UNet levels - latency chains on the generic
// kernels: one LDS-DMA round trip per k-step (profiles/r2_igemm4_ablation.txt §7, §8) - on the halo kernel's schedule (halo once per
// chunk, three weight slots, GroupNorm fold).  Their few tiles fill the chip by SPLIT-K OVER STAGES: grid.z slices the (chunk, tap)
// stage sequence [0, 9 nch) into contiguous ranges - a slice may start or end in the middle of a chunk - and every slice writes an fp32
// partial slab that splitk_reduce_kernel / splitk_reduce_stats_kernel finishes (bias, activation, residual, storage conversion,
// optional GroupNorm statistics of the stored output).
#pragma once
#include "igemm_common.h"
#include "gn_tail.h"
#include <type_traits>

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// NB: keep the LDS-DMA builtin inside a plain __device__ function (see igemm2.hip)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ void lds_dma16s(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {   // + a scalar byte offset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

// k-step 1 of a fragment address (chunk index ^ 4 = byte ^ 64), computed where it is used: as plain C++ the compiler hoists all
// twelve variants out of the tap loop and the kernel spills
__device__ __forceinline__ int xor64(int v) {
    int r;
    asm volatile("v_xor_b32 %0, 64, %1" : "=v"(r) : "v"(v));
    return r;
}

#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
__device__ long long g_ig4_clk[4 * 8192];   // per workgroup: cycle counter at kernel start / K loop start / K loop end / kernel end (ablate builds)
#endif
// ABL (builds with -DRS_SPLIT_ABLATE only, RS_IGEMM4_ABL=n selects): timing ablations, results wrong: bit 0 = no weight loads
// after the first two stages, bit 1 = no halo loads after chunk 0, bit 2 = no MFMAs
// (Round 3's NWV = 4 form - 128-pixel tiles on four waves, two workgroups per CU - measured within 1 % of this one on every shape
// (profiles/r3_negative_results.txt) and left the tree in round 5, as did the GroupNorm pass riding on taps 7 / 8, profiles/r3_igemm4_early_gn.txt.)
//
// SCHED (round 5): where a stage's REFILL code sits.  The refill of a stage - one halo piece of the next chunk and the weight tile two
// stages ahead: ~110 scalar / vector instructions of address arithmetic in front of four LDS-DMA instructions - used to run right behind
// the stage's barrier in all eight waves at once, i.e. in BOTH waves of every SIMD, with the matrix pipe idle until the first fragments
// had been read behind it (profiles/r3_igemm4_phases.txt: 3 016 cycles per split-storage stage for 1 920 cycles of MFMA issue).  With
// SCHED = 1 every wave issues its fragment reads first; the waves 0 - 3 then refill and go on to their MFMAs, the waves 4 - 7 (the
// other wave of each SIMD: a workgroup's waves go to the SIMDs in cyclic order) run the first SCH_AT channel fragments' MFMAs, refill,
// and finish - whichever wave of a SIMD is in its address arithmetic, the other one keeps the matrix pipe fed.
// s_waitcnt vmcnt(n) for an n that is a constant once the tap loop is unrolled (the asm immediate itself must be a literal)
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

template <int TW, int BC, bool SPLIT, int SEG = 0, int ABL = 0, int NWV = 8>
__global__ __launch_bounds__(64 * NWV, 2) void igemm4_kernel(IGemmParams p) {
    static_assert(SEG == 0 || (TW == 32 && (SEG == 8 || SEG == 16)), "segmented tiles use the 8 x 32 geometry");
    static_assert(NWV == 8, "8 waves, one workgroup per CU");
    constexpr int NT = 64 * NWV, WPX = NWV / 2;   // threads; pixel-waves (x 2 channel-waves)
    constexpr int NXB = 2;                        // halo buffers
    constexpr int LDSCAP = 160 * 1024;
    // Measured (profiles/r5_halo_refill_ab.txt, scripts/igemm_bench.py, conv layer mix of one pass): split storage 152.2 ms (round 4) ->
    // 144.9 with SCHED 2 (SCHED 1: 152.7, SCHED 0 on the same sources: 147.8); fp16 66.9 -> 62.8 with SCHED 1 (SCHED 2: 64.1, SCHED 0: 66.3).
    // Default: 2 for split storage, 1 for fp16.  -DRS_IG4_SCHED=n forces one form (A/B builds).
#ifdef RS_IG4_SCHED
    constexpr int SCHED = (ABL == 0) ? RS_IG4_SCHED : 0;
#else
    constexpr int SCHED = (ABL != 0) ? 0 : (SPLIT ? 2 : 1);
#endif
#ifndef RS_IG4_FAST
#define RS_IG4_FAST 1
#endif
    // (Round 5 also measured, on these sources, and dropped: the NEXT tap's pixel fragments read from the halo buffer during the current tap's
    // last MFMAs with the weight fragments one channel fragment ahead - behind a barrier the first MFMA waits for one read instead of
    // nine - : 146.4 -> 144.9 ms on the conv layer mix of the microbenchmark, 109.88 vs 109.86 ms of the family in the pass; and s_setprio 1
    // for waves 4 - 7 over the K loop: 110.1 vs 109.9 ms.  profiles/r5_halo_refill_ab.txt.  Neither start-of-stage latency is what a stage
    // waits for.  Nor is it the LDS-DMA instructions' issue cost: one piece per channel fragment with the two waves of a SIMD half a fragment
    // apart - waves 0 - 3 in front of fragment i's MFMAs, waves 4 - 7 behind them - is SLOWER, 153.1 vs 146.5 ms / 117.9 vs 113.5 ms in the pass:
    // every form that takes the two waves of a SIMD out of step costs split storage 4 - 6 %.)
    // FAST (round 5): the refill's per-lane byte offsets - XPW halo pieces and RW weight row groups of this wave - are computed ONCE and
    // kept in registers (10 VGPRs); what changes from stage to stage (chunk, tap) is a scalar and travels in the buffer instruction's
    // SGPR offset.  A stage's refill shrinks from ~100 instructions (pixel decode, bounds tests, 32-bit multiplies, per piece and row
    // group, in every one of the 9 nch stages) to the four LDS-DMA instructions and a handful of scalar ones.  The folded shortcut's
    // chunks / tiles (another base, pitch and row length; a few stages at the end of the loop) keep the general code.
    constexpr bool FAST = SEG == 0 && ABL == 0 && RS_IG4_FAST;
    constexpr int KC = SPLIT ? 32 : 64;         // input channels per chunk (one 128-byte LDS row per pixel / weight row)
    // halo row pitch HWD: TW + 2 rounded up to a multiple of 8, so that a tap's row shift ky * HWD leaves (row & 7) - the LDS
    // swizzle key - unchanged: the nine shifted fragment addresses of a lane are 3 bases (kx) + an immediate offset (ky)
    constexpr int TH = 32 * NWV / TW, HWD = (TW + 2 + 7) / 8 * 8, HROWS = (TH + 2) * HWD, HROWS_P = HROWS;
    constexpr int XBUF = HROWS_P * 128;
    constexpr int WSLOT = BC * 128;
    constexpr int WBASE = NXB * XBUF;
    // weight ring: three slots wherever 160 KB allow it (tile s+2 is requested at stage s and has two stages to land: with two
    // slots the L2 round trip of tile s+1 - ~1 us under load against a 1.1 us stage - shows up as 8 - 14 % of the kernel time,
    // profiles/r2_igemm4_ablation.txt); (TW = 64, BC = 160) and BC = 192 keep two
    constexpr int NSLOT = (NXB * XBUF + 3 * WSLOT <= LDSCAP) ? 3 : 2;
    constexpr int FP = 4, FC = BC / 32;
    constexpr int XPIECES = HROWS_P / 8;        // 1 KB LDS-DMA pieces (8 halo rows x 128 B) of one chunk
    constexpr int XPW = (XPIECES + NWV - 1) / NWV;      // pieces per wave (wave w owns pieces w, w + NWV, ...)
    constexpr int XPWF = XPIECES / NWV;                 // ... of which every wave has this many
    constexpr int RR = 8 * NWV;                         // weight rows covered by one LDS-DMA instruction of every wave
    constexpr int RWF = BC / RR, RWP = BC % RR, RW = RWF + (RWP ? 1 : 0);
    constexpr int NCELL = (HROWS_P * 8 + NT - 1) / NT;  // 16-byte LDS cells per thread in the in-LDS GroupNorm pass
    static_assert(XPW <= 8 && RWP % 8 == 0 && NXB * XBUF + NSLOT * WSLOT <= LDSCAP, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
    if (tid == 0 && blockIdx.x < 8192) g_ig4_clk[4 * blockIdx.x] = clock64();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave % WPX, wc = wave / WPX;
    const int rr = 8 * wave + (lane >> 3);
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);    // source K-chunk of this lane (swizzle on the source side)
    const bool wpart = !RWP || wave < RWP / 8;

    // ---- tile decode: channel tiles of one pixel tile are adjacent (they share the halo in L2)
    const int nby = (p.Cout + BC - 1) / BC;
    const int txb_n = p.Wo / TW, tyb_n = p.Ho / TH;
    // Small planes (SEG != 0, split-K over stage ranges): the grid is one-dimensional, id = tile * splitk + slice, so that the hardware's
    // round-robin of workgroup ids over the 8 XCDs puts the SLICES on the XCDs (splitk = 8: XCD z runs slice z of every tile) and a slice's
    // share of the weight matrix (640 x 5760 x 4 B / 8 = 1.8 MB) stays in that XCD's 4 MB L2.  With the slices on grid.z every XCD owned
    // whole tiles and fetched the WHOLE matrix: 283 MB of HBM / fabric reads per launch for 25 MB of operands (PMC, profiles/
    // r4_pmc_traffic_parity.json: the 8 x 8 level's convs were fetch-bound on weight re-reads).
    const int nslice = (SEG != 0 && p.splitk > 1) ? p.splitk : 1;
    int tile = SEG != 0 ? (int)blockIdx.x / nslice : xcd_remap(blockIdx.x, gridDim.x);
    const int nb = tile % nby; tile /= nby;
    // SEG: one tile = four images of an 8 x 8 plane (SEG = 8) or one 16 x 16 image (SEG = 16); b = first image of the tile
    const int txb = SEG ? 0 : tile % txb_n;
    if (!SEG) tile /= txb_n;
    const int tyb = SEG ? 0 : tile % tyb_n;
    const int b = SEG ? (SEG == 8 ? 4 * tile : tile) : tile / tyb_n;
    const int n0 = nb * BC, y0 = tyb * TH, x0 = txb * TW;
    const int Cin = p.C0;
    // (scalars used inside the loops / lambdas are copied out of the by-value parameter block: a lambda capture of `p` itself
    // makes the compiler park the whole struct in scratch memory, and scratch loads inside the K loop drain the DMA queue)
    const float* const xcoef = p.xcoef;
    const int xact = p.xact, Hs = p.Hs, Ws = p.Ws;
    // halo row hr of the tile -> source pixel (image index in `img`); false: zero padding / separator column / outside the image
    auto halo_src = [&](int hr, unsigned& pix, int& img) -> bool {
        const int hy = hr / HWD, hx = hr - hy * HWD;
        if constexpr (SEG == 0) {
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            img = b;
            pix = (unsigned)((b * Hs + y) * Ws + x);
            return hr < HROWS && hx < TW + 2 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        } else {
            const int q = hx - 1, sg = q / (SEG + 1), xin = q - sg * (SEG + 1);     // (q = -1: sg = 0, xin = -1)
            const int y = (SEG == 16 ? 8 * sg : 0) - 1 + hy;
            img = b + (SEG == 8 ? sg : 0);
            pix = (unsigned)((img * Hs + y) * Ws + xin);
            return hr < HROWS && hx >= 1 && (unsigned)xin < (unsigned)SEG && sg < 32 / SEG && (unsigned)y < (unsigned)Hs;
        }
    };

    constexpr unsigned INV = 0xF0000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // Refill addresses are recomputed where they are used (a dozen VALU instructions per LDS-DMA piece): kept in registers across
    // the K loop they push the kernel over 256 VGPRs, and a spilled value comes back through a scratch load - a VMEM access whose
    // wait also drains the DMA queue.
    const int Cout = p.Cout, Ktot = p.Ktot, ld0 = p.ld0;
    // Folded 1x1 shortcut (IGemmParams::sx): chunks nch .. nch + nch2 - 1 of the halo sequence are 32-channel chunks of the RAW block
    // input, stages 9 nch .. 9 nch + nch2 - 1 their centre-tap weight tiles [Cout][sC hi | sC lo] - the shortcut's GEMM as nch2 more
    // stages of this accumulator (fp16 storage: 64-channel chunks, rows [Cout][sC]; the last chunk may be half full).  Only in the
    // instantiations the engine asks for it (big planes, 8 waves).
    constexpr bool SKIPOK = SEG == 0 && NWV == 8 && ABL == 0;
    const int nch = (Cin + KC - 1) / KC;
    const int sC = (SKIPOK && p.sx) ? p.sC : 0, sld = p.sld;
    const int nch2 = (sC + KC - 1) / KC, ncht = nch + nch2;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.sx, 0, p.sx_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.sw, 0, p.sw_bytes, 0x00020000);
    auto issue_x = [&](int c, int k) {   // piece k of chunk c -> halo buffer c & 1 (wave w owns pieces w, w + NWV, ...)
        if (wave + NWV * k >= XPIECES) return;                       // wave-uniform
        const int hr = 8 * (wave + NWV * k) + (lane >> 3);
        unsigned pix; int img;
        const bool inside = halo_src(hr, pix, img);
        if constexpr (SKIPOK) {
            // a chunk of the shortcut's source: same pixels, its own base / channel count / pixel pitch - SCALAR selects in front of the
            // same vector code (a branch here would split the nine unrolled taps of the K loop into basic blocks: + 25 VGPRs, spills)
            const bool sk = c >= nch;
            const int cc = sk ? c - nch : c;
            const unsigned ldc = (unsigned)(sk ? sld : ld0), Cc = (unsigned)(sk ? sC : Cin);
            const unsigned cb = SPLIT ? (unsigned)(cc * 32 + (kcp & 3) * 8) : (unsigned)(cc * 64 + kcp * 8);
            const unsigned off = SPLIT ? pix * ldc * 4u + (kcp >> 2) * ldc * 2u + cb * 2u : pix * ldc * 2u + cb * 2u;
            lds_dma16(sk ? rsx : rx, smem + (c & 1) * XBUF + (wave + NWV * k) * 1024, (inside && cb < Cc) ? off : INV);
            return;
        }
        // source of LDS position (lane & 7) of this row = logical chunk kcp: fp16: channels 8 kcp ..; split: plane kcp >> 2
        // (0 = hi, 1 = lo, `ld0` halfs further in the pixel record), channels 8 (kcp & 3) ..
        const unsigned cb = SPLIT ? (unsigned)(c * 32 + (kcp & 3) * 8) : (unsigned)(c * 64 + kcp * 8);
        const bool ok = inside && cb < (unsigned)Cin;
        const unsigned off = SPLIT ? pix * (unsigned)ld0 * 4u + (kcp >> 2) * (unsigned)ld0 * 2u + cb * 2u : pix * (unsigned)ld0 * 2u + cb * 2u;
        lds_dma16(rx, smem + (NXB == 2 ? (c & 1) * XBUF : 0) + (wave + NWV * k) * 1024, ok ? off : INV);
    };
    auto issue_w = [&](int s, int slot) {   // weight tile of stage s = (chunk s / 9, tap s % 9) -> ring slot s % NSLOT (passed in)
        char* sbase = smem + WBASE + slot * WSLOT + (8 * wave) * 128;
        if constexpr (SKIPOK) {
            // stage s >= 9 nch: a tile of the shortcut's weights, rows [sC hi | sC lo] (scalar selects, see issue_x)
            const bool sk = s >= 9 * nch;
            const int c = sk ? nch + (s - 9 * nch) : s / 9, tap = sk ? 0 : s - c * 9;
            const unsigned Kc = (unsigned)(sk ? sC : Ktot), Cc = (unsigned)(sk ? sC : Cin);
            const unsigned cb = SPLIT ? (unsigned)((sk ? c - nch : c) * 32 + (kcp & 3) * 8) : (unsigned)((sk ? c - nch : c) * 64 + kcp * 8);
            const unsigned kb = (unsigned)tap * Cc * 2u + cb * 2u + (SPLIT ? (kcp >> 2) * Kc * 2u : 0u);
            const __amdgpu_buffer_rsrc_t rc = sk ? rsw : rw;
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                if (RWP && i == RW - 1 && !wpart) continue;
                const int n = n0 + RR * i + rr;
                const bool ok = RR * i + rr < BC && n < Cout && cb < Cc;
                lds_dma16(rc, sbase + (RR * i) * 128, ok ? (unsigned)n * Kc * (SPLIT ? 4u : 2u) + kb : INV);
            }
            return;
        }
        const int c = s / 9, tap = s - c * 9;
        const unsigned cb = SPLIT ? (unsigned)(c * 32 + (kcp & 3) * 8) : (unsigned)(c * 64 + kcp * 8);
        // weight rows: fp16 [K]; split [K hi | K lo]
        const unsigned kb = (unsigned)(tap * Cin) * 2u + cb * 2u + (SPLIT ? (kcp >> 2) * (unsigned)Ktot * 2u : 0u);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (RWP && i == RW - 1 && !wpart) continue;
            const int n = n0 + RR * i + rr;
            const bool ok = RR * i + rr < BC && n < Cout && cb < (unsigned)Cin;
            lds_dma16(rw, sbase + (RR * i) * 128, ok ? (unsigned)n * (unsigned)Ktot * (SPLIT ? 4u : 2u) + kb : INV);
        }
    };

    unsigned xv[XPW], wv[RW];   // FAST: byte offset of this lane's 16 bytes in piece k / row group i at chunk 0, tap 0 (INV: zeros)
    if constexpr (FAST) {
        const unsigned lpart = SPLIT ? (unsigned)(kcp >> 2) * (unsigned)ld0 * 2u + (unsigned)(kcp & 3) * 16u : (unsigned)kcp * 16u;
#pragma unroll
        for (int k = 0; k < XPW; ++k) {
            unsigned pix; int img;
            const bool inside = halo_src(8 * (wave + NWV * k) + (lane >> 3), pix, img);
            xv[k] = (inside && wave + NWV * k < XPIECES) ? pix * (unsigned)ld0 * (SPLIT ? 4u : 2u) + lpart : INV;
        }
        const unsigned wpart_l = SPLIT ? (unsigned)(kcp >> 2) * (unsigned)Ktot * 2u + (unsigned)(kcp & 3) * 16u : (unsigned)kcp * 16u;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int n = n0 + RR * i + rr;
            wv[i] = (RR * i + rr < BC && n < Cout) ? (unsigned)n * (unsigned)Ktot * (SPLIT ? 4u : 2u) + wpart_l : INV;
        }
    }
    // main-source piece k of chunk c (c < nch) / main weight tile of stage s (s < 9 nch) from the precomputed offsets
    auto issue_x_fast = [&](int c, int k) __attribute__((always_inline)) {
        if (wave + NWV * k >= XPIECES) return;                       // wave-uniform
        unsigned v = xv[k];
        if constexpr (!SPLIT) v = (c * 64 + 64 > Cin && kcp >= 4) ? INV : v;   // half chunk: its upper 32 channels do not exist
        lds_dma16s(rx, smem + (c & 1) * XBUF + (wave + NWV * k) * 1024, v, (unsigned)(c * KC) * 2u);
    };
    auto issue_w_fast = [&](int s, int slot) __attribute__((always_inline)) {
        char* sbase = smem + WBASE + slot * WSLOT + (8 * wave) * 128;
        const int c = s / 9, tap = s - c * 9;
        const unsigned so = (unsigned)(tap * Cin + c * KC) * 2u;
        const bool halfc = !SPLIT && c * 64 + 64 > Cin;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (RWP && i == RW - 1 && !wpart) continue;
            lds_dma16s(rw, sbase + (RR * i) * 128, (halfc && kcp >= 4) ? INV : wv[i], so);
        }
    };

    // ---- fragment addressing
    const int swz0 = ((lg ^ (lr & 7)) << 4), swz1 = (((4 + lg) ^ (lr & 7)) << 4);
    const int la = (wc * (BC / 2) + lr) * 128;
    int xfo[FP][3];   // LDS byte offset (k-step 0; k-step 1 = ^ 64) of (fragment j, lane) in the CURRENT halo buffer for kx = 0..2, ky = 0
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int ty = TW == 64 ? wp : 2 * wp + (j >> 1);
        const int tx = TW == 64 ? 16 * j : 16 * (j & 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // (SEG: pixel column tx + lr of the tile sits (tx + lr) / SEG separator columns further right)
            const int hr = ty * HWD + tx + lr + (SEG ? (tx + lr) / SEG : 0) + kx;
            xfo[j][kx] = hr * 128 + ((lg ^ (hr & 7)) << 4);
        }
    }

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- in-LDS GroupNorm (+FiLM) + activation of one halo chunk: thread t owns the 16-byte cells t, t+512, ...; their
    // channel group (cell & 7) ^ (row & 7) is the same for all of them, so 16 coefficients per chunk stay in registers
    const int cg = (tid & 7) ^ ((tid >> 3) & 7);
    unsigned cell_in = 0;   // bit k: cell k of this thread is a pixel inside the image (the others must stay exact zeros)
#pragma unroll
    for (int k = 0; k < NCELL; ++k) {
        unsigned pix; int img;
        if (halo_src((tid >> 3) + (NT / 8) * k, pix, img)) cell_in |= 1u << k;
    }
    // split storage: thread t owns, in rows (t >> 2) + 128 k, the position pair (t & 3, (t & 3) + 4) = the hi and the lo half (in
    // swizzle-dependent order) of ONE 8-channel group
    const int sp_hi = ((tid & 3) ^ ((tid >> 2) & 7)) < 4 ? (tid & 3) : (tid & 3) + 4;   // position holding the hi halves
    const int sp_cg = ((tid & 3) ^ ((tid >> 2) & 7)) & 3;                                 // channel group inside the 32-channel chunk
    unsigned sp_in = 0;
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < (HROWS_P + NT / 4 - 1) / (NT / 4); ++k) {
            unsigned pix; int img;
            if (halo_src((tid >> 2) + (NT / 4) * k, pix, img)) sp_in |= 1u << k;
        }
    }
    constexpr int NCELL_S = (HROWS_P + NT / 4 - 1) / (NT / 4);   // cell pairs per thread in the split-storage GroupNorm pass
    // the 16 affine coefficients of a thread's channel group for chunk c (scale a, shift d)
    struct Coef { f32x4 a0, a1, d0, d1; };
    auto load_coef = [&](int c) -> Coef {
        const float* sc = SPLIT ? xcoef + (long long)b * 2 * Cin + c * 32 + sp_cg * 8
                                : xcoef + (long long)b * 2 * Cin + min(c * 64 + cg * 8, Cin - 8);   // (half chunks: upper cells are never multiplied)
        return Coef{*(const f32x4*)sc, *(const f32x4*)(sc + 4), *(const f32x4*)(sc + Cin), *(const f32x4*)(sc + Cin + 4)};
    };
    auto apply_split = [&](int c, auto act_tag, Coef cf, auto k0_tag, auto k1_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
        const float* sc = xcoef + (long long)b * 2 * Cin + c * 32 + sp_cg * 8;
        f32x4 a0 = cf.a0, a1 = cf.a1, d0 = cf.d0, d1 = cf.d1;
        char* xb = smem + (NXB == 2 ? (c & 1) * XBUF : 0) + (tid >> 2) * 128;
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if (!((sp_in >> k) & 1)) continue;
            if constexpr (SEG == 8) {   // four images in the tile: the cell's own image's coefficients (L1-resident)
                const int hr = (tid >> 2) + (NT / 4) * k, hx = hr % HWD;
                const float* si = sc + (long long)((hx - 1) / (SEG + 1)) * 2 * Cin;
                a0 = *(const f32x4*)si; a1 = *(const f32x4*)(si + 4); d0 = *(const f32x4*)(si + Cin); d1 = *(const f32x4*)(si + Cin + 4);
            }
            f16x8* ch_ = (f16x8*)(xb + k * (NT / 4) * 128 + sp_hi * 16);
            f16x8* cl_ = (f16x8*)(xb + k * (NT / 4) * 128 + (sp_hi ^ 4) * 16);
            f16x8 vh = *ch_, vl = *cl_;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = fmaf(rs_join(vh[e], vl[e]), e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? d0[e & 3] : d1[e & 3]);
                // SiLU with v_exp / v_rcp (1 ulp each): fp32-class, a fifth of the IEEE-division sequence
                const float u = ACT == RS_ACT_SILU ? t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) : t;
                f16 hh, ll;
                rs_split(u, hh, ll);
                vh[e] = hh; vl[e] = ll;
            }
            *ch_ = vh; *cl_ = vl;
        }
    };
    auto apply = [&](int c, auto act_tag, Coef cf, auto k0_tag, auto k1_tag) __attribute__((always_inline)) {   // (not inlined, its closure object lives in scratch memory)
        constexpr int ACT = decltype(act_tag)::value;
        constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
        const int ch = min(c * 64 + cg * 8, Cin - 8);   // (half chunks: the upper cells are never multiplied; keep the loads in range)
        const float* sc = xcoef + (long long)b * 2 * Cin + ch;
        f32x4 a0 = cf.a0, a1 = cf.a1, d0 = cf.d0, d1 = cf.d1;
        char* xb = smem + (NXB == 2 ? (c & 1) * XBUF : 0) + tid * 16;
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if (!((cell_in >> k) & 1)) continue;
            if constexpr (SEG == 8) {   // four images in the tile: the cell's own image's coefficients (L1-resident)
                const int hr = (tid >> 3) + (NT / 8) * k, hx = hr % HWD;
                const float* si = sc + (long long)((hx - 1) / (SEG + 1)) * 2 * Cin;
                a0 = *(const f32x4*)si; a1 = *(const f32x4*)(si + 4); d0 = *(const f32x4*)(si + Cin); d1 = *(const f32x4*)(si + Cin + 4);
            }
            f16x8* cell = (f16x8*)(xb + k * NT * 16);
            f16x8 v = *cell;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (f16)rs_act_t<ACT, true>(fmaf((float)v[e], a0[e], d0[e]));
                v[4 + e] = (f16)rs_act_t<ACT, true>(fmaf((float)v[4 + e], a1[e], d1[e]));
            }
            *cell = v;
        }
    };

    // cells [K0, K1) of chunk c's halo buffer: GroupNorm affine (+FiLM) + activation, storage-specific
    auto apply_cells = [&](int c, const Coef& cf, auto k0_tag, auto k1_tag) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
            if (xact == RS_ACT_SILU) apply_split(c, std::integral_constant<int, RS_ACT_SILU>{}, cf, k0_tag, k1_tag);
            else apply_split(c, std::integral_constant<int, RS_ACT_NONE>{}, cf, k0_tag, k1_tag);
        } else {
            if (xact == RS_ACT_SILU) apply(c, std::integral_constant<int, RS_ACT_SILU>{}, cf, k0_tag, k1_tag);
            else apply(c, std::integral_constant<int, RS_ACT_NONE>{}, cf, k0_tag, k1_tag);
        }
    };
    constexpr int NC_ALL = SPLIT ? NCELL_S : NCELL;
    const int nst = nch * 9 + nch2;   // (+ the shortcut's centre-tap stages)
    // One barrier per (chunk, tap) stage: weight tile s+1 and one piece of the next chunk's halo are requested right behind it
    // and have the whole stage (40 - 48 MFMAs per wave) to land.  (A variant with the barrier between the two k-steps and two
    // fragment register sets carried across the nine unrolled taps - igemm3's schedule - needs > 256 VGPRs here: 160 spilled
    // registers, 557 instead of 810 TFLOP/s on the layer mix.)
#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
    if (tid == 0 && blockIdx.x < 8192) g_ig4_clk[4 * blockIdx.x + 1] = clock64();
#endif
    // split-K over stages: this workgroup's slice [s_beg, s_end) of the (chunk, tap) sequence (the whole sequence without split-K);
    // it may begin / end in the middle of a chunk.  (The planner leaves no slice empty.)
    // SLICED is a compile-time property (only the small-plane geometries are ever launched with split-K): without it the nine taps of a
    // chunk stay ONE straight-line block - a per-tap range test costs the scheduler its view across the taps (measured: + 20 % on every
    // big-plane shape, profiles/r3_igemm4_phases.txt)
    constexpr bool SLICED = SEG != 0;
    const int zsl = SLICED ? (int)blockIdx.x % nslice : 0;
    int s_beg = 0, s_end = nst;
    if (SLICED && p.splitk > 1) {
        const int per = (nst + p.splitk - 1) / p.splitk;
        s_beg = min(nst, zsl * per);
        s_end = min(nst, s_beg + per);
    }
    const int c_beg = SLICED ? s_beg / 9 : 0, c_last = SLICED ? (s_end - 1) / 9 : nch - 1;
#pragma unroll
    for (int k = 0; k < XPW; ++k) issue_x(c_beg, k);
    issue_w(s_beg, NSLOT == 3 ? s_beg % 3 : (s_beg & 1));
    if (NSLOT == 3 && s_beg + 1 < s_end) issue_w(s_beg + 1, (s_beg + 1) % 3);
    if (SLICED && NXB == 2 && (c_beg & 1)) {   // the first chunk of the slice sits in halo buffer 1
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) xfo[j][q] += XBUF;
    }
    for (int c = c_beg; c <= c_last; ++c) {
        const bool two = c * 64 + 64 <= Cin;     // fp16: full chunk = two k-steps of 32 channels (half chunk: one)
        const int t_first = SLICED && c == c_beg ? s_beg - 9 * c_beg : 0;   // first tap of this chunk inside the slice
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = c * 9 + tap;
            if constexpr (SLICED) { if (s < s_beg || s >= s_end) continue; }   // (workgroup-uniform)
            // weight tile s has landed.  Sliced (small-plane) kernels: a stage issues [halo piece, tile s+2] and only the loads of tile s+1 -
            // the youngest RW (waves that also own a partial row group) or RWF ones of this wave - may still be in flight, i.e. a halo piece
            // has ONE stage to land.  Big planes (round 5): a stage issues [tile s+2, halo piece]; what a wave issued AFTER tile s is the
            // piece of stage s-2, tile s+1 and the piece of stage s-1, and all of that may fly on (VMEM operations retire in order) - a
            // halo piece has TWO stages, and where in a stage the refill sits no longer decides whether the next barrier waits for HBM.
            // (Pieces are counted with their lower bound over the waves - the last round is partial - which only makes the wait stricter;
            // tap 0 still waits for the whole chunk: the taps 7, 8 in front of it issue no pieces.)
            if constexpr (SLICED) {
                if (NSLOT == 3 && s + 1 < s_end) {
                    if (RWP && wpart) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RWP ? RW - 1 : RW) : "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                const int nx1 = (tap >= 1 && tap - 1 < XPWF) ? 1 : 0, nx2 = (tap >= 2 && tap - 2 < XPWF) ? 1 : 0;
                if (s + 1 < s_end) {
                    if (NSLOT == 3) {
                        if (RWP && wpart) wait_vmcnt_n(RW + nx1 + nx2);
                        else wait_vmcnt_n((RWP ? RW - 1 : RW) + nx1 + nx2);
                    } else wait_vmcnt_n(nx1);   // two slots: tile s went out one stage ago, in front of that stage's piece
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (tap == t_first) {
                if (xcoef) {
                    apply_cells(c, load_coef(c), std::integral_constant<int, 0>{}, std::integral_constant<int, NC_ALL>{});
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (c > c_beg) {   // the fragment offsets move over to the other halo buffer
                    const int flip = (c & 1) ? XBUF : -XBUF;
#pragma unroll
                    for (int j = 0; j < FP; ++j)
#pragma unroll
                        for (int q = 0; q < 3; ++q) xfo[j][q] += flip;
                }
            }
            // refills: one piece of the next chunk's halo (buffer (c+1)&1: chunk c-1 is finished everywhere), then the next weight tile
            // (a slice that enters the chunk at tap t_first > 0 has 9 - t_first stages for the XPW pieces: the rest goes with tap 8)
            auto refill = [&]() __attribute__((always_inline)) {
                if constexpr (SLICED) {
                    if (c < c_last && !(ABL & 2)) {
                        const int k0 = tap - t_first;
                        if (k0 < XPW) issue_x(c + 1, k0);
                        if (tap == 8)
                            for (int kk = k0 + 1; kk < XPW; ++kk) issue_x(c + 1, kk);
                    }
                }
                // (ring slot of stage s: s % 3 = tap % 3 with nine taps per chunk; two slots: s & 1)
                if constexpr (FAST) {
                    if (s + NSLOT - 1 < 9 * nch) issue_w_fast(s + NSLOT - 1, NSLOT == 3 ? (tap + 2) % 3 : ((s + 1) & 1));
                    else if (s + NSLOT - 1 < s_end) issue_w(s + NSLOT - 1, NSLOT == 3 ? (tap + 2) % 3 : ((s + 1) & 1));   // (a shortcut tile)
                    if (tap < XPW) {
                        if (c + 1 < nch) issue_x_fast(c + 1, tap);
                        else issue_x(c + 1, tap);                                                                          // (a shortcut chunk, or nothing)
                    }
                    return;
                }
                if (s + NSLOT - 1 < s_end && (!(ABL & 1) || s < 1)) issue_w(s + NSLOT - 1, NSLOT == 3 ? (tap + 2) % 3 : ((s + 1) & 1));
                if constexpr (!SLICED) {
                    // behind the tile (see the wait above).  Issued in EVERY chunk so that the counted waits see a fixed number of loads per
                    // stage: past the last chunk the piece is all out-of-range lanes (zeros into the halo buffer chunk c - 1 has left)
                    if (tap < XPW && !(ABL & 2)) issue_x(c + 1, tap);
                }
            };
            if constexpr (SCHED == 0) refill();
            const char* wb = smem + WBASE + (NSLOT == 3 ? tap % 3 : (s & 1)) * WSLOT + la;
            const int ky = tap / 3, kx = tap % 3;
            // the stage's MFMAs (fragment reads of the nine shifted halo rows + the weight tile) and, SCHED = 1, its refill between them (see the
            // header): waves 0 - 3 refill in front of their MFMAs, waves 4 - 7 behind the first SCH_AT channel fragments
            const bool first_half = wave < NWV / 2;   // (scalar)
            if constexpr (SPLIT) {
                constexpr int SCH_AT = FC >= 4 ? 3 : 2;
                f16x8 ah[FC], al[FC], bh[FP], bl[FP];
                // (in the order the MFMAs want them: LDS returns are in order, the first product needs ah[0] and the hi pixel fragments)
                ah[0] = *(const f16x8*)(wb + swz0);
#pragma unroll
                for (int j = 0; j < FP; ++j) bh[j] = *(const f16x8*)(smem + xfo[j][kx] + ky * HWD * 128);
#pragma unroll
                for (int j = 0; j < FP; ++j) bl[j] = *(const f16x8*)(smem + xor64(xfo[j][kx]) + ky * HWD * 128);
                al[0] = *(const f16x8*)(wb + swz1);
#pragma unroll
                for (int i = 1; i < FC; ++i) {
                    ah[i] = *(const f16x8*)(wb + swz0 + i * 2048);
                    al[i] = *(const f16x8*)(wb + swz1 + i * 2048);
                }
                auto mm = [&](auto i0_tag, auto i1_tag) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = decltype(i0_tag)::value; i < decltype(i1_tag)::value; ++i) {
                        const f16x8 as = ah[i] * (f16)RS_LO_SCALE;   // v_pk_mul_f16: exact (|w| < 32)
#pragma unroll
                        for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                };
                if constexpr (SCHED == 0) {
                    mm(std::integral_constant<int, 0>{}, std::integral_constant<int, FC>{});
                } else if constexpr (SCHED == 2) {   // (A/B builds: every wave refills behind its first channel fragment's MFMAs)
                    __builtin_amdgcn_sched_barrier(0);
                    mm(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    refill();
                    __builtin_amdgcn_sched_barrier(0);
                    mm(std::integral_constant<int, 1>{}, std::integral_constant<int, FC>{});
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    if (first_half) refill();
                    __builtin_amdgcn_sched_barrier(0);
                    mm(std::integral_constant<int, 0>{}, std::integral_constant<int, SCH_AT>{});
                    __builtin_amdgcn_sched_barrier(0);
                    if (!first_half) refill();
                    __builtin_amdgcn_sched_barrier(0);
                    mm(std::integral_constant<int, SCH_AT>{}, std::integral_constant<int, FC>{});
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks == 1 && !two) break;
                    const int sw = ks ? swz1 : swz0;
                    f16x8 a[FC], bf[FP];
#pragma unroll
                    for (int i = 0; i < FC; ++i) a[i] = *(const f16x8*)(wb + sw + i * 2048);
#pragma unroll
                    for (int j = 0; j < FP; ++j) bf[j] = *(const f16x8*)(smem + (ks ? xor64(xfo[j][kx]) : xfo[j][kx]) + ky * HWD * 128);
                    if constexpr (SCHED == 1) {
                        if (ks == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (first_half) refill();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < FC; ++i)
#pragma unroll
                        for (int j = 0; j < FP; ++j) {
                            if (ABL & 4) { acc[i][j][0] += (float)(a[i][0] * bf[j][0]); continue; }
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bf[j], acc[i][j], 0, 0, 0);
                        }
                    if constexpr (SCHED != 0) {
                        if (ks == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (SCHED == 2 || !first_half) refill();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (SKIPOK) {
        // The shortcut's stages: one per 32-channel chunk of the raw block input (no GroupNorm pass, the centre tap only).  The same ring
        // discipline as above - stage s waits for tile s and chunk c (requested one / two stages ago), refills chunk c + 1 and tile
        // s + NSLOT - 1 behind the barrier - except that a chunk lasts ONE stage, so all of a wave's halo pieces go out together.
        for (int c = nch; c < ncht; ++c) {
            const int s = 9 * nch + (c - nch);
            if (NSLOT == 3 && s + 1 < nst) {
                if (RWP && wpart) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RWP ? RW - 1 : RW) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            {
                const int flip = (c & 1) ? XBUF : -XBUF;
#pragma unroll
                for (int j = 0; j < FP; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q) xfo[j][q] += flip;
            }
            if (c + 1 < ncht) {
#pragma unroll
                for (int k = 0; k < XPW; ++k) issue_x(c + 1, k);
            }
            // (the refills of a wave stay in the order halo pieces, then weight tile: the counted wait above relies on the tile being youngest)
            if (s + NSLOT - 1 < nst) issue_w(s + NSLOT - 1, NSLOT == 3 ? (s + 2) % 3 : ((s + 1) & 1));
            const char* wb = smem + WBASE + (NSLOT == 3 ? s % 3 : (s & 1)) * WSLOT + la;
            if constexpr (SPLIT) {
                f16x8 ah[FC], al[FC], bh[FP], bl[FP];
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    bh[j] = *(const f16x8*)(smem + xfo[j][1] + HWD * 128);
                    bl[j] = *(const f16x8*)(smem + xor64(xfo[j][1]) + HWD * 128);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    ah[i] = *(const f16x8*)(wb + swz0 + i * 2048);
                    al[i] = *(const f16x8*)(wb + swz1 + i * 2048);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const f16x8 as = ah[i] * (f16)RS_LO_SCALE;   // exact (|w| < 30: the engine folds no shortcut with larger weights)
#pragma unroll
                    for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
            } else {
                const bool two2 = (c - nch) * 64 + 64 <= sC;   // (a half chunk: one k-step of 32 channels)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks == 1 && !two2) break;
                    const int sw = ks ? swz1 : swz0;
                    f16x8 a[FC], bf[FP];
#pragma unroll
                    for (int i = 0; i < FC; ++i) a[i] = *(const f16x8*)(wb + sw + i * 2048);
#pragma unroll
                    for (int j = 0; j < FP; ++j) bf[j] = *(const f16x8*)(smem + (ks ? xor64(xfo[j][1]) : xfo[j][1]) + HWD * 128);
#pragma unroll
                    for (int i = 0; i < FC; ++i)
#pragma unroll
                        for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
    if (tid == 0 && blockIdx.x < 8192) g_ig4_clk[4 * blockIdx.x + 2] = clock64();
#endif
    // ---------------------------------------------------------------- epilogue
    // (the barrier that ends the K loop comes after the residual loads below: their latency - the 4 x FC loads of a lane used to
    // be issued per channel fragment, five round trips to L2 in a row, 8 - 10 k cycles of a 60 - 120 k cycle workgroup - overlaps it)
    // global pixel index of row r (0..63) of this wave's pixel tile
    auto pixel = [&](int r) -> long long {
        const int ty = TW == 64 ? wp : 2 * wp + (r >> 5);
        const int tx = TW == 64 ? r : (r & 31);
        if constexpr (SEG == 8) return ((long long)(b + (tx >> 3)) * 8 + ty) * 8 + (tx & 7);           // image b + segment, 8 x 8 plane
        if constexpr (SEG == 16) return ((long long)b * 16 + ty + 8 * (tx >> 4)) * 16 + (tx & 15);      // rows 8 .. 15 in segment 1
        return ((long long)b * p.Ho + y0 + ty) * p.Wo + x0 + tx;
    };
    if (SLICED && p.splitk > 1) {
        // split-K slice: raw fp32 partial sums (split storage: the accumulator carries 2^11 x the sum); scale / bias / activation /
        // residual / storage conversion are applied by the reduce kernel.  No LDS is touched: no barrier with the waves still looping.
        float* part = p.partial + (long long)zsl * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const long long m = pixel(j * 16 + lr);
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
                if (n < p.Cout) *(f32x4*)(part + m * p.Cout + n) = SPLIT ? acc[i][j] * RS_LO_INV : acc[i][j];
            }
        }
        return;
    }
    constexpr int ROWB = (BC / 2) * 2 + 16;
    constexpr int CPR = (BC / 2) / 8;
    constexpr int NITEM = 64 * CPR;
    char* stg = smem + wave * 64 * ROWB;
    f16* y = (f16*)p.y;
    const f16* res = (const f16*)p.res;
    const bool res_ok = res != nullptr;
    long long mres[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) mres[j] = pixel(j * 16 + lr) * p.ldres * (SPLIT ? 2 : 1);
    f32x4 bvs[FC];
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
        bvs[i] = p.bias ? *(const f32x4*)(p.bias + min(n, p.Cout - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (SKIPOK && sC && p.sbias) bvs[i] += *(const f32x4*)(p.sbias + min(n, p.Cout - 4));   // (the shortcut's bias)
    }
    // per-channel statistics of the stored output for the consuming GroupNorm (IGemmParams::ystats): lane (lr, lg) accumulates its
    // 4 channels of every channel fragment over its FP pixel fragments, then the 16 `lr` lanes are reduced with xor-shuffles
    float* const ystats = p.ystats;
    const bool tail_on = p.tail.coef != nullptr;
    // (LDS: the staging tiles end below 8 * 64 * ROWB <= 106 KB; the partials sit behind them)
    float* const sb = (float*)(smem + NWV * 64 * ROWB);
    // wave-level sum of one (channel fragment, register) column over the 16 pixel lanes -> LDS
    auto stats_put = [&](int i, int r, float a, float q) {
        a = rs_sum16(a); q = rs_sum16(q);   // (DPP adds; same operands and bits as the xor-shuffle butterfly they replace)
        if (lr == 0) { sb[(wave * (BC / 2) + i * 16 + lg * 4 + r) * 2] = a; sb[(wave * (BC / 2) + i * 16 + lg * 4 + r) * 2 + 1] = q; }
    };
    // GroupNorm tail (gn_tail.h): with IGemmParams::tail the workgroup that publishes the LAST statistics of image b also writes the
    // consuming GroupNorm's coefficients.  All of it is wave 0's business: it alone gathers and stores the tile's statistics, drains those
    // few stores and draws the ticket while the other seven waves are already staging and storing the output tile (nobody waits for the
    // write-through round trip); at the very end, behind one barrier, the image's last workgroup computes the coefficients with all its threads.
    unsigned* const tail_flag = (unsigned*)(smem + NWV * 64 * ROWB + NWV * (BC / 2) * 2 * sizeof(float));   // behind the staging tiles and the partials
    auto stats_out = [&]() {   // the pixel-waves' partials -> one pair per channel of the tile -> global
        __syncthreads();
        if (wave != 0) return;
        for (int c = lane; c < BC; c += 64) {
            if (n0 + c >= p.Cout) break;
            const int hw_ = c / (BC / 2), cl = c - hw_ * (BC / 2);   // channel-wave, channel inside its half
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < WPX; ++w4) { a += sb[((hw_ * WPX + w4) * (BC / 2) + cl) * 2]; q += sb[((hw_ * WPX + w4) * (BC / 2) + cl) * 2 + 1]; }
            // (SEG = 16: the tile IS the image; SEG = 8 never produces statistics here - four images per tile)
            float* dst = ystats + ((SEG ? (long long)b : (long long)b * (tyb_n * txb_n) + tyb * txb_n + txb) * p.ystats_ld + n0 + c) * 2;
            if (tail_on) rs_pub_pair(dst, a, q);   // write-through: another workgroup (the image's last arriver) reads it inside this launch
            else { dst[0] = a; dst[1] = q; }
        }
        if (tail_on) { const bool last = rs_gn_tail_arrive(p.tail, b); if (lane == 0) *tail_flag = last ? 1u : 0u; }
    };
    if constexpr (SPLIT) {
        // values finished in place (exact fp32 arithmetic), then two staging passes: the hi halves, then the lo halves
        const float osc = p.out_scale * RS_LO_INV;   // the accumulator carries 2^11 x the sum (see the header)
        const int act = p.act, ldres = p.ldres;
#ifdef RS_IG4_SPLIT_RES_FRAGS   // (A/B builds: the round-3 form everywhere)
        constexpr bool RES_FRAGS = true;
#else
        constexpr bool RES_FRAGS = SEG != 0;   // (the small-plane instantiations have no registers to spare for the row form: 4 - 5 spills; their tiles finish through the split-K reduce kernel anyway)
#endif
        if constexpr (RES_FRAGS) {
        // the residual in accumulator layout, one channel fragment ahead: 8 B per lane = 16 quarter-used cache lines per wave instruction, five
        // dependent round trips
        __syncthreads();  // all waves done with the LDS: the epilogue reuses it as staging space
        // (split storage: all FC fragment rows at once would need 80 registers next to the 80 accumulators)
        f16x4 rh[2][FP], rl[2][FP];   // residual of channel fragment i + 1 in flight while fragment i is finished
        auto load_res = [&](int i) __attribute__((always_inline)) {
            const int nr = min(n0 + wc * (BC / 2) + i * 16 + lg * 4, p.Cout - 4);
#pragma unroll
            for (int j = 0; j < FP; ++j) { rh[i & 1][j] = *(const f16x4*)(res + mres[j] + nr); rl[i & 1][j] = *(const f16x4*)(res + mres[j] + ldres + nr); }
        };
        if (res_ok) load_res(0);
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            if (res_ok && i + 1 < FC) load_res(i + 1);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                f32x4 v = acc[i][j] * osc + bvs[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (act == RS_ACT_SILU) v[r] = rs_silu(v[r]); else if (act == RS_ACT_GELU) v[r] = rs_gelu(v[r]);
                    if (res_ok) v[r] += rs_join(rh[i & 1][j][r], rl[i & 1][j][r]);
                }
                acc[i][j] = v;
            }
            if (ystats) {   // (the stored pair reproduces v to 2^-23)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int j = 0; j < FP; ++j) { a += acc[i][j][r]; q = fmaf(acc[i][j][r], acc[i][j][r], q); }
                    stats_put(i, r, a, q);
                }
            }
        }
        } else {
        // The residual as ROWS (16 B per lane, the store loop's addressing: a pixel's BC / 2 channels are contiguous - a quarter of the
        // cache-line look-ups of the accumulator-layout fetch, as in the fp16 epilogue below), one plane at a time through the wave-private
        // staging tile: the hi rows are requested in FRONT of the barrier that ends the K loop, the lo rows travel while the hi cells are
        // added.  v + hi, then + lo 2^-11 by fma: one fp32 rounding more than v + join(hi, lo) (the stored pair reproduces the sum to
        // 2^-23 either way).
        uint4 rq[CPR];
        auto load_rows = [&](int plane) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                const int idx = lane + 64 * k, row = idx / CPR, c8 = idx - row * CPR;
                const int n = n0 + wc * (BC / 2) + c8 * 8;
                rq[k] = n < p.Cout ? *(const uint4*)(res + pixel(row) * ldres * 2 + plane * ldres + n) : uint4{0u, 0u, 0u, 0u};
            }
        };
        auto park_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                const int idx = lane + 64 * k, row = idx / CPR, c8 = idx - row * CPR;
                *(uint4*)(stg + row * ROWB + c8 * 16) = rq[k];
            }
            RS_STAGING_SYNC();
        };
        if (res_ok) load_rows(0);
        __syncthreads();  // all waves done with the LDS: the epilogue reuses it as staging space
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                f32x4 v = acc[i][j] * osc + bvs[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) { if (act == RS_ACT_SILU) v[r] = rs_silu(v[r]); else if (act == RS_ACT_GELU) v[r] = rs_gelu(v[r]); }
                acc[i][j] = v;
            }
        if (res_ok) {
            park_rows();
            load_rows(1);          // the lo rows travel while the hi cells are added
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    const f16x4 c = *(const f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)c[r];
                }
            RS_STAGING_SYNC();     // every lane of the wave has read its hi cells
            park_rows();
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    const f16x4 c = *(const f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf((float)c[r], RS_LO_INV, acc[i][j][r]);
                }
            RS_STAGING_SYNC();     // ... and its lo cells: the tile is free for the store passes
        }
        if (ystats) {   // (the stored pair reproduces v to 2^-23)
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int j = 0; j < FP; ++j) { a += acc[i][j][r]; q = fmaf(acc[i][j][r], acc[i][j][r], q); }
                    stats_put(i, r, a, q);
                }
        }
        }
        if (ystats) stats_out();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    f16x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { f16 hh, ll; rs_split(acc[i][j][r], hh, ll); h[r] = half ? ll : hh; }
                    *(f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = h;
                }
            RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
            for (int idx = lane; idx < NITEM; idx += 64) {
                const int row = idx / CPR, c8 = idx - row * CPR;
                const int n = n0 + wc * (BC / 2) + c8 * 8;
                if (n >= p.Cout) continue;
                *(uint4*)(y + pixel(row) * p.ldy * 2 + half * p.ldy + n) = *(const uint4*)(stg + row * ROWB + c8 * 16);
            }
            RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
        }
    } else {
        auto run = [&](auto res_tag) __attribute__((always_inline)) {
        constexpr bool RES = decltype(res_tag)::value;
#ifdef RS_IG4_RES_FRAGS   // (A/B: the residual in accumulator layout, 8 B per lane = 16 quarter-used cache lines per wave instruction)
        f16x4 rv[FC][FP];   // the residual of the whole wave tile, all loads in flight together
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int nr = min(n0 + wc * (BC / 2) + i * 16 + lg * 4, p.Cout - 4);
#pragma unroll
                for (int j = 0; j < FP; ++j) rv[i][j] = *(const f16x4*)(res + mres[j] + nr);
            }
        }
        __syncthreads();  // all waves done with the LDS: the epilogue reuses it as staging space
#else
        // The residual of the wave tile as ROWS: 16 B per lane with the store loop's addressing (a pixel's BC / 2 channels are
        // contiguous), requested in front of the barrier, parked in the wave-private staging tile behind it and added from there.  In
        // accumulator layout (8 B per lane, 16 pixels x 32 B per instruction) the same data cost 4 x the cache-line look-ups: the
        // epilogue was 14 - 16 k cycles with a residual, 6 k without (profiles/r3_igemm4_phases.txt, r3_attn_phases.txt).
        uint4 rq[CPR];
        if constexpr (RES) {
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                const int idx = lane + 64 * k, row = idx / CPR, c8 = idx - row * CPR;
                const int n = n0 + wc * (BC / 2) + c8 * 8;
                rq[k] = n < p.Cout ? *(const uint4*)(res + pixel(row) * p.ldres + n) : uint4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();  // all waves done with the LDS: the epilogue reuses it as staging space
        if constexpr (RES) {
#pragma unroll
            for (int k = 0; k < CPR; ++k) {
                const int idx = lane + 64 * k, row = idx / CPR, c8 = idx - row * CPR;
                *(uint4*)(stg + row * ROWB + c8 * 16) = rq[k];
            }
            RS_STAGING_SYNC();   // wave-private tile: the wave's own LDS order suffices
        }
#endif
        auto finish = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    f32x4 v = acc[i][j] * p.out_scale + bvs[i];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rs_act_t<ACT, true>(v[r]);
                    f16x4* const cell = (f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2);   // read and rewritten by this lane only
                    if constexpr (RES) {
#ifdef RS_IG4_RES_FRAGS
                        const f16x4 rr = rv[i][j];
#else
                        const f16x4 rr = *cell;
#endif
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                    }
                    f16x4 h;
                    h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)v[3];
                    *cell = h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float f = (float)h[r]; s1[r] += f; s2[r] = fmaf(f, f, s2[r]); }   // of the STORED value
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ystats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) stats_put(i, r, s1[r], s2[r]);
                }
            }
        };
        if (p.act == RS_ACT_GELU) finish(std::integral_constant<int, RS_ACT_GELU>{});
        else if (p.act == RS_ACT_SILU) finish(std::integral_constant<int, RS_ACT_SILU>{});
        else finish(std::integral_constant<int, RS_ACT_NONE>{});
        };
        if (res_ok) run(std::true_type{}); else run(std::false_type{});
        if (ystats) stats_out();
        RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
        for (int idx = lane; idx < NITEM; idx += 64) {
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int n = n0 + wc * (BC / 2) + c8 * 8;
            if (n >= p.Cout) continue;
            *(uint4*)(y + pixel(row) * p.ldy + n) = *(const uint4*)(stg + row * ROWB + c8 * 16);
        }
    }
    if (tail_on && ystats) {   // (kernel-uniform) every wave is done with its staging tile behind this barrier: the image's last workgroup turns to the coefficients
        __syncthreads();
        if (*tail_flag) rs_gn_tail_finish<NT>(p.tail, b, (float*)smem);
    }
#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
    if (tid == 0 && blockIdx.x < 8192) g_ig4_clk[4 * blockIdx.x + 3] = clock64();
#endif
}

#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
}  // namespace
// ablate builds: mean cycles of the three kernel phases over the first `nwg` workgroups of the last igemm4 launch
extern "C" int rs_igemm4_phase_cycles(int nwg, double* out3) {
    static long long h[4 * 8192];
    if (nwg < 1 || nwg > 8192) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ig4_clk), sizeof(long long) * 4 * nwg) != hipSuccess) return -1;
    out3[0] = out3[1] = out3[2] = 0.0;
    for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 3; ++k) out3[k] += (double)(h[4 * i + k + 1] - h[4 * i + k]) / nwg;
    return 0;
}
namespace {
#endif

template <int TW, int BC, bool SPLIT, int SEG = 0, int NWV = 8>
hipError_t launch4_cfg(IGemmParams p, hipStream_t st) {
    constexpr int TH = 32 * NWV / TW, NXB = NWV == 8 ? 2 : 1;
    constexpr size_t cap = NWV == 8 ? 160 * 1024 : 80 * 1024;
    constexpr size_t xbuf = (size_t)((TH + 2) * ((TW + 2 + 7) / 8 * 8)) * 128;
    constexpr size_t lds = NXB * xbuf + ((NXB * xbuf + 3 * BC * 128 <= cap) ? 3 : 2) * (size_t)BC * 128;   // (NSLOT of the kernel)
    const int tiles = (SEG == 8 ? p.B / 4 : SEG == 16 ? p.B : p.B * (p.Ho / TH) * (p.Wo / TW)) * ((p.Cout + BC - 1) / BC);
    const int sk = p.splitk > 1 ? p.splitk : 1;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)igemm4_kernel<TW, BC, SPLIT, SEG, 0, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const size_t esz = SPLIT ? 4 : 2;
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * esz, wb = (size_t)p.Cout * p.Ktot * esz;
    if (xb >= 0xF0000000ull || wb >= 0xF0000000ull) return hipErrorInvalidValue;  // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    if (p.sx) {   // folded 1x1 shortcut: only where the kernel carries it (see SKIPOK), whole 32-channel chunks, 16-byte rows
        const size_t sxb = (size_t)p.B * p.Hs * p.Ws * p.sld * esz, swb = (size_t)p.Cout * p.sC * esz;
        if (SEG != 0 || NWV != 8 || sk > 1 || !p.sw || p.sC < 32 || (p.sC % 32) || (p.sld % 8) || p.sld < p.sC || sxb >= 0xF0000000ull ||
            ((uintptr_t)p.sx & 15) || ((uintptr_t)p.sw & 15))
            return hipErrorInvalidValue;
        p.sx_bytes = (unsigned)sxb;
        p.sw_bytes = (unsigned)swb;
    } else { p.sC = 0; p.sx_bytes = p.sw_bytes = 0; }
    if (p.tail.coef) {   // GroupNorm tail: this launch's statistics are segment 0; every (pixel tile, channel tile) workgroup of an image arrives once
        if (!p.ystats || sk > 1 || SEG == 8) return hipErrorInvalidValue;   // (split-K slices / four-image tiles: the reduce kernel carries the tail)
        if (p.tail.C > 2048 || p.tail.groups < 1 || p.tail.groups > 64 || (p.tail.C % p.tail.groups)) return hipErrorInvalidValue;   // (the finish's LDS scratch: 2 C + 2 groups floats)
        const int per_image = SEG == 16 ? 1 : (p.Ho / TH) * (p.Wo / TW);
        p.tail.expected = per_image * ((p.Cout + BC - 1) / BC);
        p.tail.st0 = p.ystats; p.tail.S0 = per_image; p.tail.ld0 = p.ystats_ld; p.tail.n0 = p.Cout;
    }
#if defined(RS_SPLIT_ABLATE) && defined(RS_IGEMM4_MAIN_TU)
    if constexpr (!SPLIT && BC == 160 && SEG == 0 && NWV == 8) {
        static const int abl = []() { const char* v = getenv("RS_IGEMM4_ABL"); return v ? atoi(v) : 0; }();
#define RS_ABL4_CASE(A)                                                                                                              \
    if (abl == A) {                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)igemm4_kernel<TW, BC, SPLIT, 0, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((igemm4_kernel<TW, BC, SPLIT, 0, A>), dim3(tiles), dim3(512), lds, st, p);                                    \
        return hipGetLastError();                                                                                                      \
    }
        RS_ABL4_CASE(1) RS_ABL4_CASE(2) RS_ABL4_CASE(3) RS_ABL4_CASE(4) RS_ABL4_CASE(7)
#undef RS_ABL4_CASE
    }
#endif
    if (SEG != 0) hipLaunchKernelGGL((igemm4_kernel<TW, BC, SPLIT, SEG, 0, NWV>), dim3(tiles * sk), dim3(64 * NWV), lds, st, p);   // (id = tile * sk + slice: see the kernel)
    else hipLaunchKernelGGL((igemm4_kernel<TW, BC, SPLIT, SEG, 0, NWV>), dim3(tiles, 1, sk), dim3(64 * NWV), lds, st, p);
    return hipGetLastError();
}

}  // namespace
