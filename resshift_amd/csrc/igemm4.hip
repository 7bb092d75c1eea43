// Halo-tile implicit GEMM (see igemm4_kernel.h): the single-image tile geometries (4 x 64 / 8 x 32 output pixels), eligibility /
// split-K planning and the launch entry points.  The small-plane geometries (SEG > 0) are instantiated in igemm4s.hip.
#define RS_IGEMM4_MAIN_TU 1
#include "igemm4_kernel.h"
#include <algorithm>

namespace {

template <bool SPLIT>
hipError_t launch4_t(const IGemmParams& p, int TW, int BC, hipStream_t st) {
    if constexpr (!SPLIT) {   // (split storage never plans the 192-channel tile - over the register budget, rs_igemm4_plan - so it is not instantiated)
        if (BC == 192) return TW == 64 ? launch4_cfg<64, 192, false>(p, st) : launch4_cfg<32, 192, false>(p, st);
    } else if (BC == 192) return hipErrorInvalidValue;
    if (TW == 64) return BC == 160 ? launch4_cfg<64, 160, SPLIT>(p, st) : launch4_cfg<64, 128, SPLIT>(p, st);
    return BC == 160 ? launch4_cfg<32, 160, SPLIT>(p, st) : launch4_cfg<32, 128, SPLIT>(p, st);
}

}  // namespace

extern "C" int rs_igemm4_seg_launch(const IGemmParams* pp, int in_dt, int SEG, int BC, hipStream_t st);   // igemm4s.hip
extern "C" int rs_splitk_reduce_launch(const IGemmParams* p, int out_dt, hipStream_t st);                  // igemm.hip

namespace {

// Finish of a split-K halo conv whose consumer is a GroupNorm: y = act(scale * sum_z partial[z] + bias) + res like splitk_reduce_kernel,
// plus the per-(image, slab, channel) sum / sum of squares of the STORED output (IGemmParams::ystats, [B][S][ystats_ld][2], slab = 256
// consecutive pixels of an image, or the whole image when it is smaller) so that the GroupNorm needs no statistics pass.  Grid (image
// slabs, Cout / 64); thread (pl, q) owns channel quad q of the block's 64 channels for pixels pl, pl + 16, ... of the slab.
// Deterministic (fixed summation order, no atomics).
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(IGemmParams p, int HW, int SP) {
    constexpr int PM = Store<TO>::PM;
    // (one LDS array: the reduction scratch [16][16][8], reused by the GroupNorm tail - 2 C + 2 groups floats + the flag word)
    __shared__ float lds[2 * 1280 + 2 * 32 + 4 > 2048 ? 2 * 1280 + 2 * 32 + 4 : 2048];
    float (*red)[16][8] = (float (*)[16][8])lds;
    const int tid = threadIdx.x, q = tid & 15, pl = tid >> 4;
    const int S = (HW + SP - 1) / SP;
    const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
    const int n = blockIdx.y * 64 + q * 4;
    const long long slab = (long long)p.M * p.Cout;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.Cout) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4*)(p.bias + n);
        const int px1 = min(HW, (sl + 1) * SP);
        for (int px = sl * SP + pl; px < px1; px += 16) {
            const long long m = (long long)b * HW + px;
            f32x4 s = *(const f32x4*)(p.partial + m * p.Cout + n);
            for (int zz = 1; zz < p.splitk; ++zz) {
                const f32x4 t = *(const f32x4*)(p.partial + zz * slab + m * p.Cout + n);
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = epi_act<TO>(s[r] * p.out_scale + bv[r], p.act);
            if (p.res) {
                float rv[4];
                Out4<TO>::load((const TO*)p.res + m * p.ldres * PM + n, rv, p.ldres);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            }
            TO* yp = (TO*)p.y + m * p.ldy * PM + n;
            Out4<TO>::store(yp, v, p.ldy);
            // statistics of the STORED value: the fp16 rounding of v; a (hi, lo) pair reproduces v to 2^-23 (as in the tile epilogue)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float w = sizeof(TO) == 2 && PM == 1 ? (float)(f16)v[r] : v[r];
                s1[r] += w; s2[r] = fmaf(w, w, s2[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { red[pl][q][r] = s1[r]; red[pl][q][4 + r] = s2[r]; }
    __syncthreads();
    if (tid < 64) {   // thread = channel of the block: fixed-order sum over the 16 pixel lanes
        const int qq = tid >> 2, r = tid & 3;
        float a = 0.f, c2 = 0.f;
        for (int k = 0; k < 16; ++k) { a += red[k][qq][r]; c2 += red[k][qq][4 + r]; }
        const int nn = blockIdx.y * 64 + tid;
        if (nn < p.Cout) {
            float* dst = p.ystats + (((long long)b * S + sl) * p.ystats_ld + nn) * 2;
            if (p.tail.coef) rs_pub_pair(dst, a, c2);
            else { dst[0] = a; dst[1] = c2; }
        }
    }
    // GroupNorm tail (gn_tail.h): the last (slab, channel block) workgroup of image b writes the consuming GroupNorm's coefficients - wave 0
    // stored the partials and draws the ticket
    if (p.tail.coef) {
        unsigned* const flag = (unsigned*)&lds[2 * 1280 + 2 * 32];
        if (tid < 64) { const bool last = rs_gn_tail_arrive(p.tail, b); if (tid == 0) *flag = last ? 1u : 0u; }
        __syncthreads();
        if (*flag) rs_gn_tail_finish<256>(p.tail, b, lds);
    }
}

int reduce_stats_launch(const IGemmParams& p_in, int out_dt, hipStream_t st) {
    IGemmParams p = p_in;
    const int HW = p.Ho * p.Wo, SP = std::min(HW, 256), S = (HW + SP - 1) / SP;
    if ((p.Cout & 3) || (p.ldy & 3) || (p.res && (p.ldres & 3))) return -2;
    const dim3 grid(p.B * S, (p.Cout + 63) / 64);
    if (p.tail.coef) {
        if (p.tail.C > 1280 || p.tail.groups > 32) return -2;   // (the kernel's LDS scratch)
        p.tail.expected = S * (int)grid.y;
        p.tail.st0 = p.ystats; p.tail.S0 = S; p.tail.ld0 = p.ystats_ld; p.tail.n0 = p.Cout;
    }
    if (out_dt == RS_F16) hipLaunchKernelGGL((splitk_reduce_stats_kernel<f16>), grid, dim3(256), 0, st, p, HW, SP);
    else if (out_dt == RS_F16S) hipLaunchKernelGGL((splitk_reduce_stats_kernel<h2s>), grid, dim3(256), 0, st, p, HW, SP);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// the same finish for the split-K launches of the generic kernels (igemm_split.hip)
extern "C" int rs_splitk_reduce_stats_launch(const IGemmParams* p, int out_dt, hipStream_t st) { return reduce_stats_launch(*p, out_dt, st); }

// Which launches take the halo kernel, and how: fp16 or split storage in / out (the same one), one source, 3x3 stride 1 pad 1 without
// the folded upsample, input channels in multiples of 32, output channels in multiples of 8 with a channel tile of at most 192, no
// batching.  Geometry: planes that tile by 4 x 64 or 8 x 32 output pixels with enough tiles to fill the chip (SEG = 0, no split-K), or
// the small planes of the 16 x 16 / 8 x 8 UNet levels (SEG = 16: one 16 x 16 image per tile; SEG = 8: four 8 x 8 images per tile, batch
// a multiple of 4) with split-K over stages chosen so that tiles x slices fill the chip once (SK).  RS_IGEMM_V4=0 disables the kernel
// (1: fp16 only, 2: split only); RS_IGEMM_V4_SEG=0 keeps the small planes on the generic kernels (A/B runs).
// *SEG carries the tile variant: 0 = 256-pixel tiles on 8 waves, 8 / 16 = the small-plane geometries.
extern "C" int rs_igemm4_plan(const IGemmParams* pp, int in_dt, int out_dt, int nz, int* TW, int* BC, int* SEG, int* SK) {
    static const int on = []() { const char* e = getenv("RS_IGEMM_V4"); return e ? atoi(e) : 3; }();
    // bit 0: fp16, bit 1: split storage, bit 2: the 16 x 16 planes as well.  Default 2 = the 8 x 8 planes in split storage - measured
    // (profiles/r3_small_planes.txt, us per launch at batch 32, generic kernel -> halo kernel): split 8 x 8, 640 -> 640: 96 -> 62, 1280 ->
    // 640: 179 -> 105; split 16 x 16, 320 -> 320: 56 -> 70; fp16 8 x 8: 33 -> 39; fp16 16 x 16: 37 -> 41 (fp16 slices are 11 stages short:
    // prologue, partial slab and reduce dominate and igemm2's 64-pixel tiles win; the 16 x 16 planes already give the generic split
    // kernel 256 tiles)
    static const int seg_on = []() { const char* e = getenv("RS_IGEMM_V4_SEG"); return e ? atoi(e) : 2; }();
    static const int min_tiles = []() { const char* e = getenv("RS_IGEMM_V4_MINTILES"); return e ? atoi(e) : 192; }();
    static const int sk_target = []() { const char* e = getenv("RS_IGEMM_V4_SKTARGET"); return e ? atoi(e) : 256; }();
    static const int sk_minst = []() { const char* e = getenv("RS_IGEMM_V4_SKMINSTAGES"); return e ? atoi(e) : 6; }();
    const IGemmParams& p = *pp;
    if (in_dt != out_dt || nz != 1 || p.C1 != 0 || p.no_halo) return 0;
    if (!((in_dt == RS_F16 && (on & 1)) || (in_dt == RS_F16S && (on & 2)))) return 0;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.up != 1 || p.Ho != p.Hs || p.Wo != p.Ws) return 0;
    // (fp16 storage fetches the residual as 16-byte row pieces: 8-channel alignment of its stride; the BASE alignment is checked at launch -
    // eligibility must be a function of the layout alone, never of a pointer value: the engine asks this planner in its dry sizing pass, in
    // want_stats and again at launch, and the three answers have to agree)
    if ((p.C0 % 32) || (p.ld0 % 8) || (p.Cout % 8) || (p.ldy % 8) || (p.res && (p.ldres % 8))) return 0;
    auto waste = [&](int bc) { return ((p.Cout + bc - 1) / bc) * bc - p.Cout; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    int seg = 0;
    if (p.Ho == 16 && p.Wo == 16 && (seg_on & 4)) seg = 16;
    else if (p.Ho == 8 && p.Wo == 8 && p.B % 4 == 0) seg = 8;
    if (seg && !((in_dt == RS_F16 && (seg_on & 1)) || (in_dt == RS_F16S && (seg_on & 2)))) seg = 0;
    if (seg) {
        // small planes: 8 x 32 geometry in segments, channel tile 128 or 160, split-K over the 9 nch stages
        if (p.Cout < 96) return 0;
        const int KC = in_dt == RS_F16S ? 32 : 64;
        const int nst = 9 * ((p.C0 + KC - 1) / KC);
        const long long tiles = (long long)(seg == 8 ? p.B / 4 : p.B) * ((p.Cout + best - 1) / best);
        int sk = 1;
        if (tiles < min_tiles) {
            sk = (int)std::max<long long>(1, std::min<long long>(sk_target / tiles, nst / std::max(1, sk_minst)));
            sk = std::min(sk, 32);
            const int per = (nst + sk - 1) / sk;
            sk = (nst + per - 1) / per;   // no empty slice
        }
        *TW = 32; *BC = best; *SEG = seg; *SK = sk;
        return 1;
    }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    // 4 x 64 tiles on planes that allow them, except for the 160-channel tile: 8 x 32 leaves room for the third weight slot
    int tw = (p.Wo % 64 == 0) ? 64 : 32;
    if (best == 160 && (p.Wo % 32 == 0) && (p.Ho % 8 == 0)) tw = 32;
    // (round 5: 8 x 32 tiles - halo overlap 1.33 x instead of 1.55 x - for the 128-channel tile as well: 65.6 -> 64.8 ms / 27.3 -> 27.0 ms on the
    // autoencoder's conv mix in the microbenchmark, 241.3 - 241.9 vs 240.9 - 241.3 ms in the pass: within noise, not taken; profiles/r5_negative_results.txt)
    const int th = 256 / tw;
    if ((p.Wo % tw) || (p.Ho % th)) return 0;
    if (p.Cout < 96 || (in_dt == RS_F16S && best == 192)) return 0;   // (split, BC = 192: over the register budget; no 3x3 conv of the models needs it)
    const long long tiles = (long long)p.B * (p.Ho / th) * (p.Wo / tw) * ((p.Cout + best - 1) / best);
    if (tiles < min_tiles) return 0;
    *TW = tw; *BC = best; *SEG = 0; *SK = 1;
    return 1;
}

// launch-time view of the plan: the parameter block must carry the plan's split-K factor (the engine asks rs_igemm4_plan first and
// sizes IGemmParams::partial for it); a block with another split-K factor is not a halo-kernel launch
extern "C" int rs_igemm4_pick(const IGemmParams* pp, int in_dt, int out_dt, int nz, int* TW, int* BC) {
    int seg = 0, sk = 1;
    if (!rs_igemm4_plan(pp, in_dt, out_dt, nz, TW, BC, &seg, &sk)) return 0;
    return (pp->splitk > 1 ? pp->splitk : 1) == sk ? 1 : 0;
}

extern "C" int rs_igemm4_launch(const IGemmParams* pp, int in_dt, int TW, int BC, hipStream_t st) {
    int tw = 0, bc = 0, seg = 0, sk = 1;
    if (!rs_igemm4_plan(pp, in_dt, in_dt, 1, &tw, &bc, &seg, &sk) || tw != TW || bc != BC || (pp->splitk > 1 ? pp->splitk : 1) != sk) return -2;
    if (sk > 1 && !pp->partial) return -2;
    if (((size_t)pp->x0 & 15) || ((size_t)pp->y & 15) || ((size_t)pp->res & 15)) return -2;   // 16-byte row pieces: a view whose channel offset is not a multiple of 8
    IGemmParams p = *pp;
    float* const want_stats = p.ystats;
    const GNTail want_tail = p.tail;
    if (want_tail.coef && !want_stats) return -2;   // a tail finishes statistics this launch produces
    if (sk > 1) { p.ystats = nullptr; p.tail.coef = nullptr; }   // slices cannot see the final values: the reduce kernel produces the statistics (and carries the tail)
    if (seg == 8 && want_stats && sk == 1) return -2;   // four images per tile: no per-image statistics from the tile epilogue
    int rc;
    if (seg) rc = rs_igemm4_seg_launch(&p, in_dt, seg, BC, st);
    else rc = (in_dt == RS_F16S ? launch4_t<true>(p, TW, BC, st) : launch4_t<false>(p, TW, BC, st)) == hipSuccess ? 0 : -1;
    if (rc != 0 || sk == 1) return rc;
    p.ystats = want_stats; p.tail = want_tail;
    return want_stats ? reduce_stats_launch(p, in_dt, st) : rs_splitk_reduce_launch(&p, in_dt, st);
}

// pixels per statistics slab of a halo-kernel launch that is asked for IGemmParams::ystats (0: this launch cannot produce them): the tile
// of the kernel variant, or - for split-K launches, whose reduce kernel produces them - 256 consecutive pixels / the whole small image
extern "C" int rs_igemm4_stats_px(const IGemmParams* pp, int in_dt) {
    int tw = 0, bc = 0, seg = 0, sk = 1;
    if (!rs_igemm4_plan(pp, in_dt, in_dt, 1, &tw, &bc, &seg, &sk)) return 0;
    const int HW = pp->Ho * pp->Wo;
    if (sk > 1) return (HW > 256 && (HW % 256)) ? 0 : std::min(HW, 256);
    if (seg == 8) return 0;
    return 256;
}
