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
// but the input halo: wave w owns positions (i, 2 jp) and (i, 2 jp + 1), i = w >> 1, jp = w & 1, of all 64 Winograd tiles (16 x 16 output
// pixels of one image) x BC = 16 CF output channels (CF = 4; 2 for the 32-channel remainder of Cout = 160).
//   * halo chunk (18 x 18 pixels x 32 channels) -> LDS by LDS-DMA, three buffers, pixel rows of 128 B; every wave converts the rows IT
//     fetched from (hi, lo) pairs to fp32 IN PLACE (a pair and an fp32 are both 4 bytes), applying the GroupNorm affine + SiLU on the way,
//     while chunk c computes: chunk c + 2 lands, chunk c + 1 (landed behind the previous barrier) is converted.  ONE workgroup barrier per
//     32-channel chunk.
//   * every row of B^T has exactly two non-zeros, so V_p of a tile is a signed sum of FOUR pixels: a wave reads them straight from the
//     fp32 halo (8 ds_read_b128 per fragment pair), 3 FMAs per value with the signs as scalars (sigma folded into U at pack time), splits,
//     and has its MFMA B operand - no transformed tensor in LDS, no second barrier.  The halo is stored column-deinterleaved with a
//     20-row pitch and a 32-byte swizzle keyed on (row >> 1) & 3, odd k-groups reading their two halves in the opposite order: the
//     stride-2 pixel reads of 16 tiles x 4 k-groups are bank-conflict-free (brute-forced over the ds_read_b128 lane groups).
//   * the transformed weights are PRIVATE to a wave (its two positions): streamed from L2 straight into registers one step ahead, in
//     fragment-major order (1 KB hi + 1 KB lo per 16 channels x 32 k: one contiguous 16 bytes per lane and instruction).
//   * epilogue: the sixteen positions of a (tile, channel) meet through LDS (two rounds of 32 tiles), output transform, x 2^-11, bias,
//     residual, split, whole-line stores; statistics in a fixed order; tail as in igemm4.
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

constexpr int W_PITCH = 20;                 // halo row pitch in LDS rows (18 columns deinterleaved: evens at 0..8, odds at 10..18; 9 and 19 stay zero)
constexpr int W_HROWS = 18 * W_PITCH;       // 360 LDS rows of 128 B = 45 LDS-DMA pieces of 1 KB
constexpr int W_UPW = 3;                    // 16-row units (2 KB: two DMA pieces = one in-place conversion instruction) per wave: wave w owns units w, w + 8, w + 16
constexpr int W_NUNIT = 8 * W_UPW;          // 24 units = 384 rows: rows 360 .. 383 exist only so that all waves run the same code (zeros, never read)
constexpr int W_HBUF = W_NUNIT * 2048;      // 49 152 B per halo buffer
constexpr int W_NBUF = 3;                   // halo buffers: chunk c computes, chunk c + 1 is converted, chunk c + 2 lands
constexpr int W_COEF = W_NBUF * W_HBUF;     // GroupNorm coefficients [2][Cin] fp32
constexpr int W_LDS = 160 * 1024;
constexpr int W_MAXCIN = 640;               // coefficient table: 5 120 B
static_assert(W_COEF + W_MAXCIN * 8 <= W_LDS, "LDS");
constexpr int W_TS = 272;                   // epilogue exchange: bytes per (position, tile) row = 64 channels fp32 + 16 (bank spread)
constexpr unsigned W_INV = 0xF0000000u;

// position (i, j) of B^T d B as sigma (first + tau second): rows / columns {first, second} of the 4 x 4 input patch
__host__ __device__ constexpr int w_first(int i) { return i == 0 ? 0 : 1; }
__host__ __device__ constexpr int w_second(int i) { return i == 3 ? 3 : 2; }
__host__ __device__ constexpr float w_tau(int i) { return i == 1 ? 1.f : -1.f; }
__host__ __device__ constexpr float w_sigma(int i) { return i == 2 ? -1.f : 1.f; }

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
__device__ __forceinline__ void wino_body(const IGemmParams& p, char* smem, int nb, int txb, int tyb, int b) {
    static_assert(CF == 4 || CF == 2, "channel block of 64 or 32");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int txb_n = p.Wo / 16, tyb_n = p.Ho / 16;
    const int n0 = nb * 64, y0 = tyb * 16, x0 = txb * 16;
#ifdef RS_WINO_PHASES
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    const long long tstart = tprev;
#endif
    const int Cin = p.C0, ld0 = p.ld0, Hs = p.Hs, Ws = p.Ws;
    const int nch = Cin / 32;
    const float* const xcoef = p.xcoef;
    const int xact = p.xact;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.ww, 0, p.w_bytes, 0x00020000);

    // ---- this wave's positions
    const int pi = wave >> 1, jp = wave & 1;
    const float tau_a = w_tau(pi);
    const float tau_b0 = w_tau(2 * jp), tau_b1 = w_tau(2 * jp + 1);
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
            for (int h = 0; h < 2; ++h) {
                wdma16s(rx, hb + (wave + 8 * k) * 2048 + h * 1024, live ? xv[k][h] : W_INV, (unsigned)c * 64u);
            }
    };
    // in-place conversion of this wave's units of chunk c: (hi, lo) pair -> GroupNorm affine (+FiLM) -> SiLU -> fp32.  Lane = (row, g'): the
    // 32 bytes [hi x 8 | lo x 8] of channel group g become [v0..3 | v4..7].  Rows outside the image stay exact zeros.
    unsigned in_mask = 0;
#pragma unroll
    for (int k = 0; k < W_UPW; ++k) {
        unsigned pix;
        const int unit = wave + 8 * k;
        if (row_src(16 * unit + (lane >> 2), pix)) in_mask |= 1u << k;
    }
    const int cv_g = (lane & 3) ^ (((lane >> 2) >> 1) & 3);   // channel group of this lane's cells (the same in all its units: 16 unit rows leave the key alone)
    const float* const coefs = (const float*)(smem + W_COEF);
    auto convert = [&](int c, int buf, auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        f32x4 ca0 = {1.f, 1.f, 1.f, 1.f}, ca1 = ca0, cd0 = {0.f, 0.f, 0.f, 0.f}, cd1 = cd0;
        if (xcoef) {
            const float* sc = coefs + c * 32 + cv_g * 8;
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
    // bytes of each KB, fetched from L2 straight into registers one step ahead (the compiler tracks these loads' vmcnt).  (Through a
    // wave-private LDS ring by LDS-DMA - the first form of this kernel - a ds_read right behind the wave's own counted vmcnt wait read
    // stale fragments in one launch out of a few: LDS-DMA data is ordered for a ds_read only by the vmcnt wait FOLLOWED BY A BARRIER,
    // cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it".)
    const char* const wbase = (const char*)p.ww + (size_t)nb * 16u * (size_t)nch * 4u * 2048u   // (all blocks in front of this one are 4 fragments wide)
                              + (size_t)(4 * pi + 2 * jp) * (size_t)nch * (size_t)CF * 2048u;
    const int wpos_step = nch * CF * 2048;
    const int wvl = lane * 16;
    auto load_w = [&](int c, int pp, int cf, f16x8& h, f16x8& l) __attribute__((always_inline)) {
        const int cc = min(c, nch - 1);   // (the reload behind the last phase re-reads the last chunk)
        const char* q = wbase + pp * wpos_step + (cc * CF + cf) * 2048 + wvl;
        h = *(const f16x8*)q; l = *(const f16x8*)(q + 1024);
    };

    // ---- fragment addressing into the fp32 halo: lane = (tile column l & 15 of the fragment, k-group g = l >> 4); tile (ty, tx) =
    // (2 tf + (l >> 3 & 1), l & 7); pixel (a, b) of the tile sits in LDS row (2 ty + a) PITCH + tx + (b >> 1) + 10 (b & 1)
    int ax[2][4];   // [position][pixel (a1,b1) (a1,b2) (a2,b1) (a2,b2)]: byte address of this lane's FIRST 16 bytes, tile fragment 0, halo buffer 0
    {
        const int tx = lane & 7, tyl = (lane >> 3) & 1, g = lane >> 4;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int j = 2 * jp + pp, b1 = w_first(j), b2 = w_second(j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = (k & 2) ? a2 : a1, bb = (k & 1) ? b2 : b1;
                const int r = (2 * tyl + a) * W_PITCH + tx + (bb >> 1) + 10 * (bb & 1);
                ax[pp][k] = r * 128 + ((g ^ ((r >> 1) & 3)) << 5) + ((g & 1) << 4);
            }
        }
    }

    f32x4 acc[2][4][CF];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int tf = 0; tf < 4; ++tf)
#pragma unroll
            for (int i = 0; i < CF; ++i) acc[pp][tf][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: coefficients -> LDS, chunks 0 and 1 on their way, chunk 0 converted, the first position's weights
    if (xcoef) {
        const float* src = xcoef + (long long)b * 2 * Cin;
        for (int i = tid; i < 2 * Cin; i += 512) ((float*)(smem + W_COEF))[i] = src[i];
    }
    issue_halo(0, 0);
    issue_halo(1, 1);
    f16x8 wh[CF], wl[CF];   // the fragments of the position phase at hand (reloaded, fragment by fragment, behind their last use)
#pragma unroll
    for (int cf = 0; cf < CF; ++cf) load_w(0, 0, cf, wh[cf], wl[cf]);
    wait_vm<0>();
    __syncthreads();           // the coefficients; chunk 0's LDS-DMA data behind a wait AND a barrier
    convert_chunk(0, 0);
    WSTAMP(0);

    const float tau_b[2] = {tau_b0, tau_b1};
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
    int buf = 0;               // halo buffer of chunk c
    for (int c = 0; c < nch; ++c) {
        // chunk c + 1 has landed (everything this wave issued except the 2 CF fragment loads of the phase ahead) ...
        wait_vm<2 * CF>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // ... and is behind a barrier: convertible.  Chunk c is fp32 for everybody; nobody reads chunk c - 1's buffer any more
        asm volatile("" ::: "memory");
        WSTAMP(1);
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        issue_halo(c + 2, buf2);
        WSTAMP(2);
        // eight (position, tile fragment) iterations; the B operand of iteration it + 1 is formed (LDS reads + VALU) next to the MFMAs of
        // iteration it - inside the wave, whatever the other wave of the SIMD is doing
        f16x8 bh, bl, bhn, bln;
        if (WABL(2)) { bh = bl = bhn = bln = wh[0]; }
        if (!WABL(2)) make_b(0, 0, bh, bl);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pp = it >> 2, tf = it & 3;
            if (it < 7 && !WABL(2)) make_b((it + 1) >> 2, (it + 1) & 3, bhn, bln);
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                const f16x8 as = wh[cf] * (f16)RS_LO_SCALE;   // v_pk_mul_f16: exact (|U| < 32, checked when the weights are packed)
                if (WABL(1)) { acc[pp][tf][cf][0] += (float)(as[0] * bh[0] + wl[cf][1] * bl[1]); }
                else {
                acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh, acc[pp][tf][cf], 0, 0, 0);
                acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cf], bl, acc[pp][tf][cf], 0, 0, 0);
                acc[pp][tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cf], bh, acc[pp][tf][cf], 0, 0, 0);
                }
                if (tf == 3 && !WABL(8)) load_w(pp ? c + 1 : c, pp ^ 1, cf, wh[cf], wl[cf]);   // last use in this phase: the next phase's fragment
            }
            bh = bhn; bl = bln;
        }
        WSTAMP(4);
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
    const float osc = p.out_scale * RS_LO_INV;   // the accumulator carries 2^11 x the sum
    float* const ystats = p.ystats;
    const bool tail_on = p.tail.coef != nullptr;
    const f16* res = (const f16*)p.res;
    f16* y = (f16*)p.y;
    const int ldres = p.ldres, ldy = p.ldy;
    // output-transform thread: (oct = 8 channels, tile of the round, output row yy); yy is wave-uniform (waves 0-3 / 4-7)
    const int oct = tid & 7, tl = (tid >> 3) & 31, yy = tid >> 8;
    const int nch_ok = min(64, p.Cout - n0);   // (CF = 2: 32 channels of this block exist)
    const bool oct_ok = oct * 8 < nch_ok;
    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
    if (p.bias && oct_ok) { bv0 = *(const f32x4*)(p.bias + n0 + oct * 8); bv1 = *(const f32x4*)(p.bias + n0 + oct * 8 + 4); }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const float sy = yy ? -1.f : 1.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last halo requests - zeros past the end - must not land in the exchange buffer)
    if (!WABL(16)) {
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        __syncthreads();   // all waves done with the LDS (K loop / the previous round's reads)
        // this round's tiles: fragments 2 rd, 2 rd + 1
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int tfl = 0; tfl < 2; ++tfl)
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const int pos = 4 * pi + 2 * jp + pp;
                    *(f32x4*)(smem + (pos * 32 + tfl * 16 + (lane & 15)) * W_TS + (cf * 16 + 4 * (lane >> 4)) * 4) = acc[pp][2 * rd + tfl][cf];
                }
        // residual rows of this thread's two pixels, requested in front of the barrier
        const int ty = 2 * (2 * rd + (tl >> 4)) + ((tl >> 3) & 1), tx = tl & 7;
        const long long m0 = ((long long)b * p.Ho + y0 + 2 * ty + yy) * p.Wo + x0 + 2 * tx;
        uint4 rh[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}}, rl[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}};
        if (res && oct_ok) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const f16* rp = res + (m0 + x) * ldres * 2 + n0 + oct * 8;
                rh[x] = *(const uint4*)rp; rl[x] = *(const uint4*)(rp + ldres);
            }
        }
        __syncthreads();
        if (oct_ok) {
            // R_i[x] over the four columns, rows i0 .. i0 + 2 with signs (+, sy, sy)
            f32x4 Y0a = {0.f, 0.f, 0.f, 0.f}, Y0b = Y0a, Y1a = Y0a, Y1b = Y0a;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const char* base = smem + (((yy + r) * 4) * 32 + tl) * W_TS + oct * 32;
                f32x4 ma[4], mb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { ma[j] = *(const f32x4*)(base + j * 32 * W_TS); mb[j] = *(const f32x4*)(base + j * 32 * W_TS + 16); }
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
                f16* yp = y + (m0 + x) * ldy * 2 + n0 + oct * 8;
                *(f16x8*)yp = oh; *(f16x8*)(yp + ldy) = ol;
            }
        }
    }
    if (ystats) {
        // per-channel sums of the tile in a fixed order: every thread parks its 8 + 8 partials, thread ch adds the 64 contributions of its octet
        __syncthreads();
        float* sb = (float*)smem;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sb[tid * 16 + e] = s1[e]; sb[tid * 16 + 8 + e] = s2[e]; }
        __syncthreads();
        unsigned* const tail_flag = (unsigned*)(smem + 512 * 64);
        if (wave == 0) {
            const int ch = lane;
            if (ch < nch_ok) {
                float a = 0.f, q = 0.f;
                for (int k = 0; k < 64; ++k) { a += sb[((ch >> 3) + 8 * k) * 16 + (ch & 7)]; q += sb[((ch >> 3) + 8 * k) * 16 + 8 + (ch & 7)]; }
                float* dst = ystats + (((long long)b * (tyb_n * txb_n) + tyb * txb_n + txb) * p.ystats_ld + n0 + ch) * 2;
                if (tail_on) rs_pub_pair(dst, a, q);
                else { dst[0] = a; dst[1] = q; }
            }
            if (tail_on) { const bool last = rs_gn_tail_arrive(p.tail, b); if (lane == 0) *tail_flag = last ? 1u : 0u; }
        }
        if (tail_on) {
            __syncthreads();
            if (*tail_flag) rs_gn_tail_finish<512>(p.tail, b, (float*)smem);
        }
    }
    }
#ifdef RS_WINO_PHASES
    WSTAMP(6);
    ph[7] = tprev - tstart;
    if (tid == 0 && p.partial) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.partial[(size_t)blockIdx.x * 8 + i] = (float)ph[i];
    }
#endif
}

// one launch: a workgroup = (image, 16 x 16 pixel tile, channel block of 64 - or the 32-channel remainder, on the narrower body)
__global__ __launch_bounds__(512, 2) void wino_kernel(IGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile decode: the channel blocks of one pixel tile are adjacent (they share the halo in L2)
    const int nby = (p.Cout + 63) / 64;
    const int txb_n = p.Wo / 16, tyb_n = p.Ho / 16;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = tile % nby; tile /= nby;
    const int txb = tile % txb_n; tile /= txb_n;
    const int tyb = tile % tyb_n;
    const int b = tile / tyb_n;
    if (p.Cout - nb * 64 >= 64) wino_body<4>(p, smem, nb, txb, tyb, b);
    else wino_body<2>(p, smem, nb, txb, tyb, b);
}

hipError_t wino_launch_k(const IGemmParams& p, int tiles, hipStream_t st) {
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) (void)hipFuncSetAttribute((const void*)wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    hipLaunchKernelGGL(wino_kernel, dim3(tiles), dim3(512), W_LDS, st, p);
    return hipGetLastError();
}

}  // namespace

// bytes of the packed Winograd weights of a Cin -> Cout 3x3 conv: 16 positions x (hi, lo)
extern "C" size_t rs_wino_weight_bytes(int Cin, int Cout) { return (size_t)Cin * Cout * 64; }

// Pack U = G g G^T (double) as (hi, lo) pairs in the kernel's streaming order:
//   [channel block nb of 64 (the last one may be 32)][position p = 4 i + j][32-channel chunk c][16-channel fragment cf][hi 1 KB | lo 1 KB]
// 1 KB = the MFMA A fragment of 64 lanes x 8 halfs: lane (lr = l & 15, g = l >> 4) holds U_p[n0 + 16 cf + lr][32 c + 8 g + perm_g(e)], perm_g(e) =
// e for even g and (e + 4) & 7 for odd g (the order in which the kernel's lanes read their two 16-byte halves of the fp32 halo).  sigma_i sigma_j
// of the (first + tau second) form of B^T's rows is folded in.  `w`: reference layout [Cout][Cin][3][3].  Returns max |U| (the kernel scales the
// hi fragment by 2^11 in fp16: exact below 32).
extern "C" float rs_wino_pack(const float* w, int Cin, int Cout, void* dst_) {
    f16* dst = (f16*)dst_;
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int nch = Cin / 32;
    float mx = 0.f;
    std::vector<double> U((size_t)16);
    for (int n = 0; n < Cout; ++n) {
        const int nb = n / 64, nl = n - nb * 64, cfn = std::min(4, (Cout - nb * 64) / 16), cf = nl / 16, lr = nl & 15;
        const size_t blk = (size_t)nb * 16 * nch * 4 * 1024;   // in halfs: 2048 B = 1024 halfs per (position, chunk, fragment)
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
                    const size_t o = blk + ((((size_t)(4 * i + j) * nch + c) * cfn + cf) * 2) * 512 + (size_t)lane * 8 + e;
                    dst[o] = h; dst[o + 512] = l;
                }
        }
    }
    return mx;
}

// Eligibility (a function of the layout alone - the engine asks in its dry pass, in want_stats and at launch): split storage in and out,
// 3x3 / stride 1 / pad 1, one source, whole 32-channel chunks, Cout in blocks of 64 (+ 32), planes that tile by 16 x 16, Winograd weights
// packed for the layer (IGemmParams::ww), no output activation, no folded shortcut, enough tiles to fill the chip.  RS_WINO=0: off.
extern "C" int rs_wino_plan(const IGemmParams* pp, int in_dt, int out_dt, int nz) {
    static const int on = []() { const char* e = getenv("RS_WINO"); return e ? atoi(e) : 1; }();
    static const int min_tiles = []() { const char* e = getenv("RS_WINO_MINTILES"); return e ? atoi(e) : 192; }();
    const IGemmParams& p = *pp;
    if (!on || in_dt != RS_F16S || out_dt != RS_F16S || nz != 1 || !p.ww || p.C1 != 0 || p.no_halo || p.sx || p.act != RS_ACT_NONE) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.up != 1 || p.Ho != p.Hs || p.Wo != p.Ws || p.osc == 2) return 0;
    if ((p.C0 % 32) || p.C0 > W_MAXCIN || (p.ld0 % 8) || (p.Cout % 32) || (p.ldy % 8) || (p.res && (p.ldres % 8))) return 0;
    if ((p.Ho % 16) || (p.Wo % 16) || p.splitk > 1) return 0;
    const long long tiles = (long long)p.B * (p.Ho / 16) * (p.Wo / 16) * ((p.Cout + 63) / 64);
    return tiles >= min_tiles ? 1 : 0;
}

extern "C" int rs_wino_launch(const IGemmParams* pp, hipStream_t st) {
    if (!rs_wino_plan(pp, RS_F16S, RS_F16S, 1)) return -2;
    IGemmParams p = *pp;
    if (((size_t)p.x0 & 15) || ((size_t)p.y & 15) || ((size_t)p.res & 15) || ((size_t)p.ww & 15)) return -2;
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * 4, wb = rs_wino_weight_bytes(p.C0, p.Cout);
    if (xb >= 0xF0000000ull || wb >= 0xF0000000ull) return -2;   // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    const int nby = (p.Cout + 63) / 64, per_image = (p.Ho / 16) * (p.Wo / 16);
    if (p.tail.coef) {
        if (!p.ystats) return -2;
        if (p.tail.C > 2048 || p.tail.groups < 1 || p.tail.groups > 64 || (p.tail.C % p.tail.groups)) return -2;
        p.tail.expected = per_image * nby;
        p.tail.st0 = p.ystats; p.tail.S0 = per_image; p.tail.ld0 = p.ystats_ld; p.tail.n0 = p.Cout;
    }
    const int tiles = p.B * per_image * nby;
    return wino_launch_k(p, tiles, st) == hipSuccess ? 0 : -1;
}
