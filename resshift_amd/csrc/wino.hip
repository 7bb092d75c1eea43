// Winograd F(2x2, 3x3) for the 3x3 / stride-1 / pad-1 convolutions in split storage (RS_F16S), with the halo kernel's fusions: GroupNorm
// affine (+FiLM) + SiLU on the input while the halo tile sits in LDS, residual, per-channel statistics of the stored output and the
// GroupNorm tail (gn_tail.h).  Replaces igemm4_kernel<.., SPLIT = true> on the big planes: ResBlock.in_layers / out_layers convs
// (models/unet.py:128-147,173,186-206) and ResnetBlock.conv1 / conv2 (ldm/modules/diffusionmodules/model.py:100-149) - 2.25 x fewer
// multiply-adds, and none of the direct kernel's structure: no weight ring shared by the workgroup, no barrier per (chunk, tap) stage.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; F(2x2,3x3): 16 "positions" per 2 x 2 output tile)
//
// Arithmetic (priced against the parity criterion on the CPU in oracle/study_winograd.py): V = B^T d B on the JOINED fp32
// activation, re-split into an (hi, lo) fp16 pair; U = G g G^T formed in double when the weights are packed, stored as a pair; the sixteen
// GEMMs M_p = U_p V_p over the input channels with split storage's three-term product acc = (2^11 Uh) Vh + Uh Vl + Ul Vh in ONE fp32
// accumulator (igemm4's form; exact scaling for |U| < 32, checked at pack time); Y = A^T M A in fp32.
//
// Structure - position per wave.  The sixteen positions have independent accumulators, so they split over the 8 waves with NOTHING shared
// but the input halo: wave w owns positions (i, 2 jp) and (i, 2 jp + 1), i = w >> 1, jp = w & 1, of the workgroup's 32 Winograd tiles (8 x 16
// output pixels of one image) x BC = 16 CF output channels - CF = 10 (Cout = 160, 320: ALL of a 160-channel layer's outputs in one
// workgroup), 8 (128, 256, 512) or 4.  What the measurements of the first forms of this kernel said (profiles/r6_wino_*): per input value
// the transform + split costs ~7 VALU operations per position and the conversion ~9 per pixel, whatever BC is, so BC decides whether the
// matrix pipe or the VALU is the limit (64 tiles x 64 channels: 11 VALU instructions per MFMA, matrix pipe 15 % busy); the input halo
// arrives by LDS-DMA in 64-byte pieces at ~10 B / clk / CU, so every channel block that re-reads the halo costs as much as computing on
// it; and the L2 delivers the wave-private weight stream at 20 - 30 TB/s (scripts/probe/l2_stream_probe.hip), so a smaller pixel tile -
// twice the weight bytes per multiply-add - is the cheap side of the trade.
//   * halo chunk (10 x 18 pixels x 32 channels) -> LDS by LDS-DMA, three buffers, pixel rows of 128 B; every wave converts the rows IT
//     fetched from (hi, lo) pairs to fp32 IN PLACE (a pair and an fp32 are both 4 bytes), applying the GroupNorm affine + SiLU on the way,
//     while chunk c computes: chunk c + 2 lands, chunk c + 1 (landed behind the previous barrier) is converted.  ONE workgroup barrier per
//     32-channel chunk.  (LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt wait FOLLOWED BY A BARRIER - a read right
//     behind the wait saw stale bytes in one launch out of a few, cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait
//     that retires it" - hence three buffers, and hence no LDS-DMA for the weights.)
//   * every row of B^T has exactly two non-zeros, so V_p of a tile is a signed sum of FOUR pixels: a wave reads them straight from the
//     fp32 halo (8 ds_read_b128 per fragment pair), 3 FMAs per value with the signs as scalars (sigma folded into U at pack time), splits,
//     and has its MFMA B operand - no transformed tensor in LDS, no second barrier.  The halo is stored column-deinterleaved with a
//     20-row pitch and a 32-byte swizzle keyed on (row >> 1) & 3, odd k-groups reading their two halves in the opposite order: the
//     stride-2 pixel reads of 16 tiles x 4 k-groups are bank-conflict-free (brute-forced over the ds_read_b128 lane groups).
//   * the transformed weights are PRIVATE to a wave (its two positions): streamed from L2 straight into registers two steps ahead, in
//     fragment-major order (1 KB hi + 1 KB lo per 16 channels x 32 k: one contiguous 16 bytes per lane and instruction).
//   * epilogue: the sixteen positions of a (tile, channel) meet through LDS (rounds of 16 tiles x half the channels), output transform,
//     x 2^-11, bias, residual, split, stores; statistics in a fixed order; tail as in igemm4.
#include "igemm_common.h"
#include "gn_tail.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// NB: keep the LDS-DMA builtin inside a plain __device__ function (see igemm2.hip)
__device__ __forceinline__ void wdma16s(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int W_TH = 8, W_TW = 16;          // output pixels of a workgroup: 4 x 8 Winograd tiles = two MFMA column fragments of 16 tiles
constexpr int W_PITCH = 20;                 // halo row pitch in LDS rows (18 columns deinterleaved: evens at 0..8, odds at 10..18; 9 and 19 stay zero)
constexpr int W_HROWS = (W_TH + 2) * W_PITCH;   // 200 LDS rows of 128 B
constexpr int W_UPW = 2;                    // 16-row units (2 KB: two DMA pieces = one in-place conversion instruction) per wave: wave w owns units w, w + 8
constexpr int W_NUNIT = 8 * W_UPW;          // 16 units = 256 rows: rows 200 .. 255 exist only so that all waves run the same code (zeros, never read)
constexpr int W_HBUF = W_NUNIT * 2048;      // 32 768 B per halo buffer
constexpr int W_NBUF = 3;                   // halo buffers: chunk c computes, chunk c + 1 is converted, chunk c + 2 lands
constexpr int W_COEF = W_NBUF * W_HBUF;     // GroupNorm coefficients [2][Cin] fp32
constexpr int W_MAXCIN = 640;               // coefficient table: 5 120 B
constexpr int W_LDS = 160 * 1024;
static_assert(W_COEF + W_MAXCIN * 8 <= W_LDS && W_COEF + 8 * 1024 <= W_LDS, "LDS");   // (the table arrives as 8 x 1 KB LDS-DMA pieces)
constexpr unsigned W_INV = 0xF0000000u;
// weight fragments are requested W_DEPTH steps (6 MFMAs each) ahead: as many register slots as the accumulators leave room for
__host__ __device__ constexpr int w_depth_of(int CF) { return CF <= 8 ? 4 : 2; }

// position (i, j) of B^T d B as sigma (first + tau second): rows / columns {first, second} of the 4 x 4 input patch
__host__ __device__ constexpr int w_first(int i) { return i == 0 ? 0 : 1; }
__host__ __device__ constexpr int w_second(int i) { return i == 3 ? 3 : 2; }
__host__ __device__ constexpr float w_tau(int i) { return i == 1 ? 1.f : -1.f; }
__host__ __device__ constexpr float w_sigma(int i) { return i == 2 ? -1.f : 1.f; }
// channel block of a layer: all 160 / 320 channels in blocks of 160, powers of two in blocks of 128, anything else in blocks of 64
__host__ __device__ constexpr int w_cf_of(int Cout) { return (Cout % 160) == 0 ? 10 : ((Cout % 128) == 0 ? 8 : 4); }

// -DRS_WINO_PHASES: wave 0 of every workgroup accumulates the cycles it spends per phase (s_memtime) into IGemmParams::partial
// [workgroup][8]: 0 prologue, 1 wait + barrier, 2 halo issue, 3 B operand, 4 MFMA steps, 5 conversion, 6 epilogue, 7 total
// ... and IGemmParams::dbg switches work off (timing ablations, results wrong): 1 no MFMAs, 2 no B operand formation, 4 no conversion,
// 8 no weight loads behind the prologue, 16 no epilogue
#ifdef RS_WINO_PHASES
#define WSTAMP(i) do { const long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tprev; tprev = t_; } while (0)
#define WABL(bit) (p.dbg & (bit))
#else
#define WSTAMP(i) do {} while (0)
#define WABL(bit) false
#endif

template <int CF>
__global__ __launch_bounds__(512, 2) void wino_kernel(IGemmParams p) {
    constexpr int W_DEPTH = w_depth_of(CF);
    static_assert(CF % W_DEPTH == 0, "register slots for the weight fragments: the slot of a step is cf % W_DEPTH");
    constexpr int BC = 16 * CF, CFH = CF / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef RS_WINO_PHASES
    long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (8 .. 13: epilogue segments - drain, first barrier, exchange writes, second barrier, transform + stores, statistics)
    long long tprev = __builtin_readcyclecounter();
    const long long tstart = tprev;
#endif
    // ---- tile decode: the channel blocks of one pixel tile are adjacent (they share the halo in L2)
    const int nby = p.Cout / BC;
    const int txb_n = p.Wo / W_TW, tyb_n = p.Ho / W_TH;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = tile % nby; tile /= nby;
    const int txb = tile % txb_n; tile /= txb_n;
    const int tyb = tile % tyb_n;
    const int b = tile / tyb_n;
    const int n0 = nb * BC, y0 = tyb * W_TH, x0 = txb * W_TW;
    const int Cin = p.C0, ld0 = p.ld0, Hs = p.Hs, Ws = p.Ws;
    const int nch = Cin / 32;
    // Every workgroup walks the 32-channel chunks in its own rotation: iteration c works on chunk (c + c0) mod nch.  All workgroups read the SAME
    // weight fragments; started in step they hit the same L2 lines at the same time (the weight stream ran at half the rate of the probe's, whose
    // streams start at scattered phases).  The sum over the chunks is order-independent up to fp32 rounding; the order is fixed per workgroup.
    // (the rotations in flight at one time span R chunks of the weight matrix - kept under ~2 MB so that the window stays L2-resident: rotating a
    // 13 MB matrix over all its chunks turned the 640 -> 320 layer's L2 hits into misses, 274 -> 371 us)
    const int R = max(1, min(nch, 1024 / p.Cout));
    const int c0 = (int)((blockIdx.x >> 3) % (unsigned)R);
    auto chunk_of = [&](int c) -> int { const int t = c + c0; return t >= nch ? t - nch : t; };   // (c < nch)
    const float* const xcoef = p.xcoef;
    const int xact = p.xact;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.x_bytes, 0x00020000);

    // ---- this wave's positions
    const int pi = wave >> 1, jp = wave & 1;
    const float tau_a = w_tau(pi);
    // (which of the wave's two positions is worked on first alternates between workgroups - `pflip`: one more way in which workgroups that run in
    // step do not ask the L2 for the same weight lines at the same moment)
    const int pflip = (int)((blockIdx.x >> 3) / (unsigned)max(1, min(p.C0 / 32, 1024 / p.Cout))) & 1;
    const float tau_b[2] = {w_tau(2 * jp + pflip), w_tau(2 * jp + (1 ^ pflip))};
    const int a1 = w_first(pi), a2 = w_second(pi);

    // ---- halo row -> source pixel
    auto row_src = [&](int row, unsigned& pix) -> bool {
        const int hy = row / W_PITCH, sx = row - hy * W_PITCH;
        const int hx = sx < 10 ? 2 * sx : 2 * (sx - 10) + 1;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        pix = (unsigned)((b * Hs + y) * Ws + x);
        return row < W_HROWS && hx < 18 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
    };
    // LDS-DMA of this wave's units: piece (unit, half) = rows 16 unit + 8 half .. + 7; lane -> (row, 16-byte slot); slot = 2 g' + part,
    // g' = g ^ key(row): the source channel group g of physical position g' (swizzle on the source side), part 0 = hi plane, 1 = lo plane
    unsigned xv[W_UPW][2];
#pragma unroll
    for (int k = 0; k < W_UPW; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int unit = wave + 8 * k, row = 16 * unit + 8 * h + (lane >> 3), slot = lane & 7;
            const int g = (slot >> 1) ^ ((row >> 1) & 3);
            unsigned pix;
            const bool ok = row_src(row, pix);
            xv[k][h] = ok ? pix * (unsigned)ld0 * 4u + (unsigned)(slot & 1) * (unsigned)ld0 * 2u + (unsigned)g * 16u : W_INV;
        }
    auto issue_halo = [&](int c, int buf) __attribute__((always_inline)) {   // chunk c -> halo buffer `buf` = c % 3 (c >= nch: zeros; keeps the counted wait static)
        char* hb = smem + buf * W_HBUF;
        const bool live = c < nch;
#pragma unroll
        for (int k = 0; k < W_UPW; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) wdma16s(rx, hb + (wave + 8 * k) * 2048 + h * 1024, live ? xv[k][h] : W_INV, (unsigned)(live ? chunk_of(c) : 0) * 64u);
    };
    // in-place conversion of this wave's units of chunk c: (hi, lo) pair -> GroupNorm affine (+FiLM) -> SiLU -> fp32.  Lane = (row, g'): the
    // 32 bytes [hi x 8 | lo x 8] of channel group g become [v0..3 | v4..7].  Rows outside the image stay exact zeros.
    unsigned in_mask = 0;
#pragma unroll
    for (int k = 0; k < W_UPW; ++k) {
        unsigned pix;
        if (row_src(16 * (wave + 8 * k) + (lane >> 2), pix)) in_mask |= 1u << k;
    }
    const int cv_g = (lane & 3) ^ (((lane >> 2) >> 1) & 3);   // channel group of this lane's cells (the same in all its units: 16 unit rows leave the key alone)
    const float* const coefs = (const float*)(smem + W_COEF);
    auto convert = [&](int c, int buf, auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        f32x4 ca0 = {1.f, 1.f, 1.f, 1.f}, ca1 = ca0, cd0 = {0.f, 0.f, 0.f, 0.f}, cd1 = cd0;
        if (xcoef) {
            const float* sc = coefs + chunk_of(c) * 32 + cv_g * 8;
            ca0 = *(const f32x4*)sc; ca1 = *(const f32x4*)(sc + 4); cd0 = *(const f32x4*)(sc + Cin); cd1 = *(const f32x4*)(sc + Cin + 4);
        }
        char* hb = smem + buf * W_HBUF + lane * 32;
#pragma unroll
        for (int k = 0; k < W_UPW; ++k) {
            // (no branch on the cell's position: a row outside the image holds zeros and gets zeros back)
            const bool in = (in_mask >> k) & 1;
            char* cell = hb + (wave + 8 * k) * 2048;
            const f16x8 vh = *(const f16x8*)cell, vl = *(const f16x8*)(cell + 16);
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = fmaf(rs_join(vh[e], vl[e]), e < 4 ? ca0[e & 3] : ca1[e & 3], e < 4 ? cd0[e & 3] : cd1[e & 3]);
                float u = ACT == RS_ACT_SILU ? t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) : t;
                u = in ? u : 0.f;
                if (e < 4) o0[e & 3] = u; else o1[e & 3] = u;
            }
            *(f32x4*)cell = o0; *(f32x4*)(cell + 16) = o1;
        }
    };
    auto convert_chunk = [&](int c, int buf) __attribute__((always_inline)) {
        if (xact == RS_ACT_SILU) convert(c, buf, std::integral_constant<int, RS_ACT_SILU>{});
        else convert(c, buf, std::integral_constant<int, RS_ACT_NONE>{});
    };

    // ---- weight stream of this wave: two positions, [nb][position][chunk][cf][hi 1 KB | lo 1 KB]; a lane's MFMA A fragment is 16 contiguous
    // bytes of each KB, fetched from L2 straight into registers W_DEPTH steps ahead (the compiler tracks these loads' vmcnt)
    const char* const wbase = (const char*)p.ww + ((size_t)nb * 16u + (size_t)(4 * pi + 2 * jp)) * (size_t)nch * (size_t)CF * 2048u;
    const int wpos_step = nch * CF * 2048;
    const int wvl = lane * 16;
    // step s of the stream = (chunk, position pp, fragment cf) in consumption order; `ds` steps ahead of (c, pp, cf)
    auto load_w = [&](int c, int pp, int cf, int ds, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        const int t = pp * CF + cf + ds, q = t / CF;
        const int cc = chunk_of(min(c + (q >> 1), nch - 1));   // (the requests behind the last step re-read the last chunk)
        const char* src = wbase + ((q & 1) ^ pflip) * wpos_step + (cc * CF + t % CF) * 2048 + wvl;
        h = *(const f16x8*)src; l = *(const f16x8*)(src + 1024);
    };

    // ---- fragment addressing into the fp32 halo: lane = (tile column l & 15 of the fragment, k-group g = l >> 4); tile (ty, tx) =
    // (2 tf + (l >> 3 & 1), l & 7); pixel (a, b) of the tile sits in LDS row (2 ty + a) PITCH + tx + (b >> 1) + 10 (b & 1)
    int ax[2][4];   // [position][pixel (a1,b1) (a1,b2) (a2,b1) (a2,b2)]: byte address of this lane's FIRST 16 bytes, tile fragment 0, halo buffer 0
    {
        const int tx = lane & 7, tyl = (lane >> 3) & 1, g = lane >> 4;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int j = 2 * jp + (pp ^ pflip), b1 = w_first(j), b2 = w_second(j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = (k & 2) ? a2 : a1, bb = (k & 1) ? b2 : b1;
                const int r = (2 * tyl + a) * W_PITCH + tx + (bb >> 1) + 10 * (bb & 1);
                ax[pp][k] = r * 128 + ((g ^ ((r >> 1) & 3)) << 5) + ((g & 1) << 4);
            }
        }
    }
    // B operand of position pp, tile fragment tf: V' = (P11 + tau_b P12) + tau_a (P21 + tau_b P22) per channel, split into (hi, lo)
    auto make_b = [&](int pp, int tf, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        const char* q = smem + tf * (4 * W_PITCH * 128);
        const float tb = tau_b[pp];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 px[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) px[k] = *(const f32x4*)(q + (half ? (ax[pp][k] ^ 16) : ax[pp][k]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaf(fmaf(px[3][e], tb, px[2][e]), tau_a, fmaf(px[1][e], tb, px[0][e]));
                f16 hh, ll;
                rs_split(v, hh, ll); h[4 * half + e] = hh; l[4 * half + e] = ll;
            }
        }
    };

    f32x4 acc[2][2][CF];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int tf = 0; tf < 2; ++tf)
#pragma unroll
            for (int i = 0; i < CF; ++i) acc[pp][tf][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: coefficients -> LDS, chunks 0 and 1 on their way, chunk 0 converted, the first steps' weights
    if (xcoef) {
        // one LDS-DMA instruction per wave (1 KB each; the descriptor's bound zero-fills what lies behind the image's 2 Cin floats), in front of the
        // halo requests: it lands behind the same wait + barrier as chunk 0.  (The copy loop it replaces - load, wait, ds_write, up to three
        // dependent round trips - stood in front of the first halo request of every workgroup.)
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(xcoef + (long long)b * 2 * Cin), 0, 2 * Cin * 4, 0x00020000);
        wdma16s(rc, smem + W_COEF + wave * 1024, (unsigned)(wave * 1024 + lane * 16), 0u);
    }
    issue_halo(0, 0);
    issue_halo(1, 1);
    f16x8 wh[W_DEPTH], wl[W_DEPTH];   // fragments of the steps ahead: slot = step % W_DEPTH
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < W_DEPTH; ++s) load_w(0, 0, 0, s, wh[s], wl[s]);
    __builtin_amdgcn_sched_barrier(0);
    // chunk 0 (and the coefficients in front of it) have landed: everything issued behind them - chunk 1's four pieces, the fragment loads - flies on
    wait_vm<2 * W_UPW + 2 * W_DEPTH>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the coefficients; chunk 0's LDS-DMA data behind a wait AND a barrier
    asm volatile("" ::: "memory");
    convert_chunk(0, 0);
    WSTAMP(0);

    int buf = 0;               // halo buffer of chunk c
    for (int c = 0; c < nch; ++c) {
        // chunk c + 1 has landed (everything this wave issued except the fragment loads of the W_DEPTH steps ahead) ...
        wait_vm<2 * W_DEPTH>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // ... and is behind a barrier: convertible.  Chunk c is fp32 for everybody; nobody reads chunk c - 1's buffer any more
        asm volatile("" ::: "memory");
        WSTAMP(1);
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        issue_halo(c + 2, buf2);
        WSTAMP(2);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            f16x8 bh[2], bl[2];
            if (WABL(2)) { bh[0] = bh[1] = bl[0] = bl[1] = wh[0]; }
            else {
#pragma unroll
                for (int tf = 0; tf < 2; ++tf) { make_b(pp, tf, bh[tf], bl[tf]); __builtin_amdgcn_sched_barrier(0); }
            }
            WSTAMP(3);
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                const f16x8 ah = wh[cf % W_DEPTH], al = wl[cf % W_DEPTH];
                const f16x8 as = ah * (f16)RS_LO_SCALE;   // v_pk_mul_f16: exact (|U| < 32, checked when the weights are packed)
                if (WABL(1)) { acc[pp][0][cf][0] += (float)(as[0] * bh[0][0] + al[1] * bl[1][1]); }
                else {
#pragma unroll
                    for (int tf = 0; tf < 2; ++tf) acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh[tf], acc[pp][tf][cf], 0, 0, 0);
#pragma unroll
                    for (int tf = 0; tf < 2; ++tf) acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[tf], acc[pp][tf][cf], 0, 0, 0);
#pragma unroll
                    for (int tf = 0; tf < 2; ++tf) acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[tf], acc[pp][tf][cf], 0, 0, 0);
                }
                // the fragments W_DEPTH steps ahead take this step's registers
                __builtin_amdgcn_sched_barrier(0);
                if (!WABL(8)) load_w(c, pp, cf, W_DEPTH, wh[cf % W_DEPTH], wl[cf % W_DEPTH]);
                __builtin_amdgcn_sched_barrier(0);
            }
            WSTAMP(4);
        }
        // this wave's rows of the next chunk - landed before this chunk's barrier - become fp32 while the other waves compute
        if (c + 1 < nch && !WABL(4)) convert_chunk(c + 1, buf1);
        WSTAMP(5);
        {   // the fragment addresses move over to the next halo buffer
            const int flip = buf == 2 ? -2 * W_HBUF : W_HBUF;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int k = 0; k < 4; ++k) ax[pp][k] += flip;
        }
        buf = buf1;
    }

    // ---------------------------------------------------------------- epilogue
    // The sixteen positions of a (tile, channel) sit in eight waves: they meet through LDS in four rounds of (tile fragment, channel half) -
    // 16 positions x 16 tiles x 8 CFH channels - and a thread = (tile, 8 channels, output row yy) applies A^T . A, x 2^-11, bias, residual.
    constexpr int TS = CFH * 64 + 16;           // bytes per (position, tile) row of the exchange buffer: CFH x 16 channels fp32 + 16 (bank spread)
    constexpr int NO = 2 * CFH;                 // channel octets per round
    constexpr int NITEM = 16 * NO * 2;          // threads with work per round
    static_assert(16 * 16 * TS <= W_LDS && NITEM <= 512, "exchange");
    const float osc = p.out_scale * RS_LO_INV;   // the accumulator carries 2^11 x the sum
    float* const ystats = p.ystats;
    const bool tail_on = p.tail.coef != nullptr;
    const f16* res = (const f16*)p.res;
    f16* y = (f16*)p.y;
    const int ldres = p.ldres, ldy = p.ldy;
    const bool item = tid < NITEM;
    const int oct = tid % NO, tl = (tid / NO) & 15, yy = item ? tid / (16 * NO) : 0;
    const float sy = yy ? -1.f : 1.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last halo requests - zeros past the end - must not land in the exchange buffer)
#ifdef RS_WINO_PHASES
    { const long long t_ = __builtin_readcyclecounter(); ph[8] += t_ - tprev; }
    long long tq = __builtin_readcyclecounter();
#define WQ(i) do { const long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tq; tq = t_; } while (0)
#else
#define WQ(i) do {} while (0)
#endif
    if (!WABL(16)) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int nc = n0 + half * CFH * 16 + oct * 8;   // first of this thread's 8 channels
        f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
        if (p.bias && item) { bv0 = *(const f32x4*)(p.bias + nc); bv1 = *(const f32x4*)(p.bias + nc + 4); }
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int tf = 0; tf < 2; ++tf) {
            // (raw barriers: __syncthreads() would also drain the previous round's global stores - vmcnt(0) - at every round)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // all waves done with the LDS (K loop / the previous round's reads)
            asm volatile("" ::: "memory");
            WQ(9);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int cq = 0; cq < CFH; ++cq)
                    *(f32x4*)(smem + ((4 * pi + 2 * jp + (pp ^ pflip)) * 16 + (lane & 15)) * TS + (cq * 16 + 4 * (lane >> 4)) * 4) = acc[pp][tf][half * CFH + cq];
            // residual rows of this thread's two pixels, requested in front of the barrier
            const int ty = 2 * tf + ((tl >> 3) & 1), tx = tl & 7;
            const long long m0 = ((long long)b * p.Ho + y0 + 2 * ty + yy) * p.Wo + x0 + 2 * tx;
            uint4 rh[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}}, rl[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}};
            if (res && item) {
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const f16* rp = res + (m0 + x) * ldres * 2 + nc;
                    rh[x] = *(const uint4*)rp; rl[x] = *(const uint4*)(rp + ldres);
                }
            }
            WQ(10);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WQ(11);
            if (item) {
                // R_i[x] over the four columns, rows i0 .. i0 + 2 with signs (+, sy, sy)
                f32x4 Y0a = {0.f, 0.f, 0.f, 0.f}, Y0b = Y0a, Y1a = Y0a, Y1b = Y0a;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const char* base = smem + (((yy + r) * 4) * 16 + tl) * TS + oct * 32;
                    f32x4 ma[4], mb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ma[j] = *(const f32x4*)(base + j * 16 * TS); mb[j] = *(const f32x4*)(base + j * 16 * TS + 16); }
                    const f32x4 r0a = ma[0] + ma[1] + ma[2], r0b = mb[0] + mb[1] + mb[2];
                    const f32x4 r1a = ma[1] - ma[2] - ma[3], r1b = mb[1] - mb[2] - mb[3];
                    const float sg = r == 0 ? 1.f : sy;
                    Y0a += r0a * sg; Y0b += r0b * sg; Y1a += r1a * sg; Y1b += r1b * sg;
                }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const f32x4 va = (x ? Y1a : Y0a) * osc + bv0, vb = (x ? Y1b : Y0b) * osc + bv1;
                    float v[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    if (res) {
                        const f16x8 h8 = __builtin_bit_cast(f16x8, rh[x]), l8 = __builtin_bit_cast(f16x8, rl[x]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rs_join(h8[e], l8[e]);
                    }
                    f16x8 oh, ol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f16 hh, ll;
                        rs_split(v[e], hh, ll); oh[e] = hh; ol[e] = ll;
                        s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]);   // (the stored pair reproduces v to 2^-23)
                    }
                    f16* yp = y + (m0 + x) * ldy * 2 + nc;
                    *(f16x8*)yp = oh; *(f16x8*)(yp + ldy) = ol;
                }
            }
            WQ(12);
        }
        if (ystats) {
            // per-channel sums of the tile in a fixed order: every thread parks its 8 + 8 partials, then the 32 (tile, row) contributions of a
            // channel are added in index order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            float* sb = (float*)smem;
            if (item) {
                *(f32x4*)(sb + tid * 16) = f32x4{s1[0], s1[1], s1[2], s1[3]}; *(f32x4*)(sb + tid * 16 + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
                *(f32x4*)(sb + tid * 16 + 8) = f32x4{s2[0], s2[1], s2[2], s2[3]}; *(f32x4*)(sb + tid * 16 + 12) = f32x4{s2[4], s2[5], s2[6], s2[7]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // (two levels: four threads per channel add eight contributions each, then one adds the four - a fixed order; the one-level form was a
            // serial chain of 32 LDS round trips on 80 threads while 432 waited: 2.9 k of the epilogue's 28 k cycles, twice)
            float* sb2 = sb + 512 * 16;
            if (tid < NO * 8 * 4) {
                const int part = tid / (NO * 8), ch = tid - part * (NO * 8);
                const int o = ch >> 3, e = ch & 7;
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int k = 8 * part; k < 8 * part + 8; ++k) { a += sb[(o + NO * k) * 16 + e]; q += sb[(o + NO * k) * 16 + 8 + e]; }
                sb2[(part * NO * 8 + ch) * 2] = a; sb2[(part * NO * 8 + ch) * 2 + 1] = q;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tid < NO * 8) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int part = 0; part < 4; ++part) { a += sb2[(part * NO * 8 + tid) * 2]; q += sb2[(part * NO * 8 + tid) * 2 + 1]; }
                float* dst = ystats + (((long long)b * (tyb_n * txb_n) + tyb * txb_n + txb) * p.ystats_ld + n0 + half * CFH * 16 + tid) * 2;
                if (tail_on) rs_pub_pair(dst, a, q);
                else { dst[0] = a; dst[1] = q; }
            }
            WQ(13);
        }
    }
    if (ystats && tail_on) {
        // (the statistics were published by the first 8 NO threads in both halves: each of those waves drains its stores, then - behind a
        // barrier - wave 0 draws the ticket)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const tail_flag = (unsigned*)(smem + 65536);   // (behind the statistics scratch and the finish's 2 C + 2 groups floats)
        if (wave == 0) { const bool last = rs_gn_tail_arrive(p.tail, b); if (lane == 0) *tail_flag = last ? 1u : 0u; }
        __syncthreads();
        if (*tail_flag) rs_gn_tail_finish<512>(p.tail, b, (float*)smem);
    }
    }
#ifdef RS_WINO_PHASES
    WSTAMP(6);
    ph[7] = tprev - tstart;
    if (tid == 0 && p.partial) {
#pragma unroll
        for (int i = 0; i < 16; ++i) p.partial[(size_t)blockIdx.x * 16 + i] = (float)ph[i];
    }
#endif
}

template <int CF> hipError_t wino_launch_cf(const IGemmParams& p, int tiles, hipStream_t st) {
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) (void)hipFuncSetAttribute((const void*)wino_kernel<CF>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    hipLaunchKernelGGL((wino_kernel<CF>), dim3(tiles), dim3(512), W_LDS, st, p);
    return hipGetLastError();
}

}  // namespace

// bytes of the packed Winograd weights of a Cin -> Cout 3x3 conv: 16 positions x (hi, lo)
extern "C" size_t rs_wino_weight_bytes(int Cin, int Cout) { return (size_t)Cin * Cout * 64; }

// Pack U = G g G^T (double) as (hi, lo) pairs in the kernel's streaming order:
//   [channel block nb of 16 CF][position p = 4 i + j][32-channel chunk c][16-channel fragment cf][hi 1 KB | lo 1 KB],  CF = w_cf_of(Cout)
// 1 KB = the MFMA A fragment of 64 lanes x 8 halfs: lane (lr = l & 15, g = l >> 4) holds U_p[n0 + 16 cf + lr][32 c + 8 g + perm_g(e)], perm_g(e) =
// e for even g and (e + 4) & 7 for odd g (the order in which the kernel's lanes read their two 16-byte halves of the fp32 halo).  sigma_i sigma_j
// of the (first + tau second) form of B^T's rows is folded in.  `w`: reference layout [Cout][Cin][3][3].  Returns max |U| (the kernel scales the
// hi fragment by 2^11 in fp16: exact below 32).
extern "C" float rs_wino_pack(const float* w, int Cin, int Cout, void* dst_) {
    f16* dst = (f16*)dst_;
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int nch = Cin / 32, CF = w_cf_of(Cout), BC = 16 * CF;
    float mx = 0.f;
    for (int n = 0; n < Cout; ++n) {
        const int nb = n / BC, nl = n - nb * BC, cf = nl / 16, lr = nl & 15;
        const size_t blk = (size_t)nb * 16 * nch * CF * 1024;   // in halfs: 2048 B = 1024 halfs per (position, chunk, fragment)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* g9 = w + ((size_t)n * Cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int k = 0; k < 3; ++k) t[i][k] = G[i][0] * g9[0 * 3 + k] + G[i][1] * g9[1 * 3 + k] + G[i][2] * g9[2 * 3 + k];
            const int c = ci / 32, g = (ci % 32) / 8, e0 = ci % 8;
            const int e = (g & 1) ? ((e0 + 4) & 7) : e0;
            const int lane = g * 16 + lr;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = (t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]) * (double)(w_sigma(i) * w_sigma(j));
                    const float uf = (float)u;
                    mx = std::max(mx, std::fabs(uf));
                    f16 h, l;
                    rs_split(uf, h, l);
                    const size_t o = blk + ((((size_t)(4 * i + j) * nch + c) * CF + cf) * 2) * 512 + (size_t)lane * 8 + e;
                    dst[o] = h; dst[o + 512] = l;
                }
        }
    }
    return mx;
}

// Eligibility (a function of the layout alone - the engine asks in its dry pass, in want_stats and at launch): split storage in and out,
// 3x3 / stride 1 / pad 1, one source, whole 32-channel chunks, Cout in blocks of 160 / 128 / 64, planes that tile by 8 x 16, Winograd weights
// packed for the layer (IGemmParams::ww), no output activation, no folded shortcut, enough tiles to fill the chip.  RS_WINO=0: off.
extern "C" int rs_wino_plan(const IGemmParams* pp, int in_dt, int out_dt, int nz) {
    static const int on = []() { const char* e = getenv("RS_WINO"); return e ? atoi(e) : 1; }();
    static const int min_tiles = []() { const char* e = getenv("RS_WINO_MINTILES"); return e ? atoi(e) : 192; }();
    const IGemmParams& p = *pp;
    if (!on || in_dt != RS_F16S || out_dt != RS_F16S || nz != 1 || !p.ww || p.C1 != 0 || p.no_halo || p.sx || p.act != RS_ACT_NONE) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.up != 1 || p.Ho != p.Hs || p.Wo != p.Ws || p.osc == 2) return 0;
    if ((p.C0 % 32) || p.C0 > W_MAXCIN || (p.ld0 % 8) || (p.Cout % (16 * w_cf_of(p.Cout))) || (p.ldy % 8) || (p.res && (p.ldres % 8))) return 0;
    if ((p.Ho % W_TH) || (p.Wo % W_TW) || p.splitk > 1) return 0;
    const long long tiles = (long long)p.B * (p.Ho / W_TH) * (p.Wo / W_TW) * (p.Cout / (16 * w_cf_of(p.Cout)));
    return (tiles >= min_tiles || (p.dbg & 64)) ? 1 : 0;   // (dbg bit 6: the op-level test entry runs shapes that do not fill the chip)
}

// pixels per statistics slab of a wino launch (one slab per 8 x 16 pixel tile); workgroups of a launch
extern "C" int rs_wino_stats_px() { return W_TH * W_TW; }
extern "C" int rs_wino_tiles(const IGemmParams* p) { return p->B * (p->Ho / W_TH) * (p->Wo / W_TW) * (p->Cout / (16 * w_cf_of(p->Cout))); }

extern "C" int rs_wino_launch(const IGemmParams* pp, hipStream_t st) {
    if (!rs_wino_plan(pp, RS_F16S, RS_F16S, 1)) return -2;
    IGemmParams p = *pp;
    if (((size_t)p.x0 & 15) || ((size_t)p.y & 15) || ((size_t)p.res & 15) || ((size_t)p.ww & 15)) return -2;
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * 4;
    if (xb >= 0xF0000000ull) return -2;   // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb;
    const int CF = w_cf_of(p.Cout), nby = p.Cout / (16 * CF), per_image = (p.Ho / W_TH) * (p.Wo / W_TW);
    if (p.tail.coef) {
        if (!p.ystats) return -2;
        if (p.tail.C > 2048 || p.tail.groups < 1 || p.tail.groups > 64 || (p.tail.C % p.tail.groups)) return -2;
        p.tail.expected = per_image * nby;
        p.tail.st0 = p.ystats; p.tail.S0 = per_image; p.tail.ld0 = p.ystats_ld; p.tail.n0 = p.Cout;
    }
    const int tiles = p.B * per_image * nby;
    const hipError_t e = CF == 10 ? wino_launch_cf<10>(p, tiles, st) : (CF == 8 ? wino_launch_cf<8>(p, tiles, st) : wino_launch_cf<4>(p, tiles, st));
    return e == hipSuccess ? 0 : -1;
}
