// Direct (non-MFMA) convolutions for the bandwidth-trivial ends of the networks:
//   * few input channels  (UNet input conv 6->160 models/unet.py:707; AE conv_in 3->128 / 3->512
//     ldm/modules/diffusionmodules/model.py:471,579; feature_extractor convs models/unet.py:697)
//   * few output channels (UNet out head 160->3 models/unet.py:862; AE conv_out 512->3 / 128->3
//     model.py:516,621; quant_conv / post_quant_conv ldm/models/autoencoder.py:25-26)
// fp32 weights [K][Cout] (Cout fastest), fp32 accumulation in tap-major / channel-minor order.
#include "common.h"

namespace {


// generic: one thread per (pixel, cout); adjacent threads = adjacent couts of the same pixel, so the
// input reads broadcast and the weight reads / output writes coalesce.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void direct_conv_kernel(DirectConvParams p) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Ho * p.Wo * p.Cout;
    if (gid >= total) return;
    const int co = (int)(gid % p.Cout);
    const long long m = gid / p.Cout;
    const int HoWo = p.Ho * p.Wo;
    const int b = (int)(m / HoWo);
    const int rem = (int)(m - (long long)b * HoWo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int Ctot = p.C0 + p.C1;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const TI* x0 = (const TI*)p.x0;
    const TI* x1 = (const TI*)p.x1;
    float acc = p.bias ? p.bias[co] : 0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
        const int iy = oy * p.stride - p.pad_t + ky;
        if ((unsigned)iy >= (unsigned)Hv) continue;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int ix = ox * p.stride - p.pad_l + kx;
            if ((unsigned)ix >= (unsigned)Wv) continue;
            const long long pix = ((long long)b * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush);
            const float* wk = p.w + (long long)((ky * p.KW + kx) * Ctot) * p.Cout + co;
            const TI* s0 = x0 + pix * p.ld0 * Store<TI>::PM;
            for (int c = 0; c < p.C0; ++c) acc = fmaf(rs_ld<TI>(s0 + c, p.ld0), wk[(long long)c * p.Cout], acc);
            if (p.C1) {
                const TI* s1 = x1 + pix * p.ld1 * Store<TI>::PM;
                wk += (long long)p.C0 * p.Cout;
                for (int c = 0; c < p.C1; ++c) acc = fmaf(rs_ld<TI>(s1 + c, p.ld1), wk[(long long)c * p.Cout], acc);
            }
        }
    }
    rs_st<TO>((TO*)p.y + m * p.ldy * Store<TO>::PM + co, p.ldy, rs_apply_act(acc, p.act));
}

// few output channels (<= NCO): one thread per pixel, 16-byte vector loads along C, weights are
// wave-uniform (scalar loads).  Single source only (C1 == 0), C0 % 8 == 0.
template <typename TI, typename TO, int NCO>
__global__ __launch_bounds__(256) void smallcout_conv_kernel(DirectConvParams p) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long M = (long long)p.B * p.Ho * p.Wo;
    if (m >= M) return;
    const int HoWo = p.Ho * p.Wo;
    const int b = (int)(m / HoWo);
    const int rem = (int)(m - (long long)b * HoWo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const TI* x0 = (const TI*)p.x0;
    float acc[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) acc[o] = (p.bias && o < p.Cout) ? p.bias[o] : 0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
        const int iy = oy * p.stride - p.pad_t + ky;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int ix = ox * p.stride - p.pad_l + kx;
            const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            const long long pix = ok ? ((long long)b * p.Hs + (iy >> ush)) * p.Ws + (ix >> ush) : 0;
            const TI* s0 = x0 + pix * p.ld0 * Store<TI>::PM;
            const float* wk = p.w + (long long)((ky * p.KW + kx) * p.C0) * p.Cout;
            for (int c = 0; c < p.C0; c += 8) {
                Vec8<TI> v;
                v.load(s0 + c, p.ld0);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xv = ok ? v.get(e) : 0.f;
#pragma unroll
                    for (int o = 0; o < NCO; ++o)
                        if (o < p.Cout) acc[o] = fmaf(xv, wk[(long long)(c + e) * p.Cout + o], acc[o]);
                }
            }
        }
    }
    TO* y = (TO*)p.y + m * p.ldy * Store<TO>::PM;
#pragma unroll
    for (int o = 0; o < NCO; ++o)
        if (o < p.Cout) rs_st<TO>(y + o, p.ldy, rs_apply_act(acc[o], p.act));
}

// The three output heads - UNet `out` (models/unet.py:855-862: GroupNorm32 -> SiLU -> conv3x3 160 -> 3), Encoder.norm_out / conv_out and
// Decoder.norm_out / conv_out (ldm/modules/diffusionmodules/model.py:510-516, 615-621, 540-546, 650-659) - in ONE pass over the tensor:
// the GroupNorm arrives as per-(image, channel) affine coefficients [B][2][C] (GNParams::coef: from a tail or the coefficient kernel),
// the normalised + SiLU'd tensor never exists in HBM, and the 3-channel conv does not occupy a 64-channel MFMA tile (the implicit-GEMM
// kernels spent 83 us on the UNet's head at batch 32 - 13 TFLOP/s - behind a 45 us normalisation pass; the head is HBM-bound: 84 MB to read).
// One workgroup = 8 x 32 output pixels of one image, one thread per pixel, NCO <= 4 accumulators; 32 input channels at a time: the 10 x 34
// halo tile is fetched with 16-byte loads, normalised / activated once per element and parked channel-major in LDS as fp32 ([32][349]: a
// thread's nine taps of a channel are then conflict-free reads), the 9 x 32 x NCO weights of the chunk sit in LDS too (broadcast reads).
// fp32 accumulation in the order (chunk, channel, tap).  Output fp32 NHWC.
struct HeadConvParams {
    const void* x; const float* coef; const float* w; const float* bias; float* y;   // w: [9][C][Cout] fp32 (ConvW::wd)
    int B, H, W, C, ldx, Cout, ldy;
};

template <typename TI, int NCO>
__global__ __launch_bounds__(256) void gn_silu_head_conv_kernel(HeadConvParams p) {
    constexpr int TH = 8, TW = 32, HH = TH + 2, HW_ = TW + 2, NPX = HH * HW_, CK = 32;
    constexpr int PITCH = 349;   // = 1 mod 4: the loader's half-wave (8 pixels x 4 channel groups, rows 8 PITCH apart) hits 32 different banks
    __shared__ float xs[CK][PITCH];
    __shared__ float ws[9 * CK * NCO];
    const int tid = threadIdx.x;
    const int txb_n = (p.W + TW - 1) / TW, tyb_n = (p.H + TH - 1) / TH;
    int tile = blockIdx.x;
    const int txb = tile % txb_n; tile /= txb_n;
    const int tyb = tile % tyb_n;
    const int b = tile / tyb_n;
    const int y0 = tyb * TH, x0 = txb * TW;
    const int ty = tid >> 5, tx = tid & 31;
    float acc[NCO];
#pragma unroll
    for (int o = 0; o < NCO; ++o) acc[o] = (p.bias && o < p.Cout) ? p.bias[o] : 0.f;
    const float* cf = p.coef + (long long)b * 2 * p.C;
    const long long rx = (long long)p.ldx * Store<TI>::PM;
    const TI* xb = (const TI*)p.x + (long long)b * p.H * p.W * rx;
    for (int c0 = 0; c0 < p.C; c0 += CK) {
        __syncthreads();   // everybody is done with the previous chunk
        // halo tile: NPX pixels x 4 groups of 8 channels
        for (int it = tid; it < NPX * (CK / 8); it += 256) {
            const int px = it >> 2, cg = it & 3;
            const int hy = px / HW_, hx = px - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx, c = c0 + cg * 8;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && c < p.C;
            float v[8];
            if (ok) {
                Vec8<TI> t;
                t.load(xb + ((long long)y * p.W + x) * rx + c, p.ldx);
                const f32x4 a0 = *(const f32x4*)(cf + c), a1 = *(const f32x4*)(cf + c + 4);
                const f32x4 d0 = *(const f32x4*)(cf + p.C + c), d1 = *(const f32x4*)(cf + p.C + c + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = fmaf(t.get(e), e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? d0[e & 3] : d1[e & 3]);
                    v[e] = rs_silu(u);          // (exact form: the heads feed the latent / the image directly)
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;   // zero padding of the NORMALISED tensor (and channels beyond C)
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[cg * 8 + e][px] = v[e];
        }
        for (int it = tid; it < 9 * CK * NCO; it += 256) {
            const int o = it % NCO, r = it / NCO, c = r % CK, tap = r / CK;
            ws[it] = (o < p.Cout && c0 + c < p.C) ? p.w[((long long)tap * p.C + c0 + c) * p.Cout + o] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < CK; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float xv = xs[c][(ty + tap / 3) * HW_ + tx + tap % 3];
#pragma unroll
                for (int o = 0; o < NCO; ++o) acc[o] = fmaf(xv, ws[(tap * CK + c) * NCO + o], acc[o]);
            }
        }
    }
    const int y = y0 + ty, x = x0 + tx;
    if (y < p.H && x < p.W) {
        float* yp = p.y + (((long long)b * p.H + y) * p.W + x) * p.ldy;
#pragma unroll
        for (int o = 0; o < NCO; ++o)
            if (o < p.Cout) yp[o] = acc[o];
    }
}

template <typename TI, typename TO>
int launch_direct(const DirectConvParams& p, hipStream_t st) {
    const long long M = (long long)p.B * p.Ho * p.Wo;
    if (p.Cout <= 8 && p.C1 == 0 && (p.C0 % 8) == 0 && (p.ld0 % 8) == 0) {
        const unsigned blocks = (unsigned)((M + 255) / 256);
        if (p.Cout <= 4) hipLaunchKernelGGL((smallcout_conv_kernel<TI, TO, 4>), dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((smallcout_conv_kernel<TI, TO, 8>), dim3(blocks), dim3(256), 0, st, p);
    } else {
        const long long total = M * p.Cout;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL((direct_conv_kernel<TI, TO>), dim3(blocks), dim3(256), 0, st, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// x: NHWC tensor of storage type in_dt (pixel stride ldx, C % 8 == 0), coef_dev [B][2][C], w_dev [9][C][Cout] fp32, y fp32 NHWC (pixel stride ldy)
extern "C" int rs_head_conv_launch(const void* x, int in_dt, const float* coef_dev, const float* w_dev, const float* bias_dev, float* y, int B, int H, int W,
                                   int C, int ldx, int Cout, int ldy, hipStream_t st) {
    if (Cout < 1 || Cout > 4 || (C % 8) || (ldx % 8) || !coef_dev || !w_dev) return -2;
    HeadConvParams p{};
    p.x = x; p.coef = coef_dev; p.w = w_dev; p.bias = bias_dev; p.y = y; p.B = B; p.H = H; p.W = W; p.C = C; p.ldx = ldx; p.Cout = Cout; p.ldy = ldy;
    const long long tiles = (long long)B * ((H + 7) / 8) * ((W + 31) / 32);
    if (tiles <= 0 || tiles > 0x7fffffffLL) return -2;
    const dim3 grid((unsigned)tiles);
    if (in_dt == RS_F16) hipLaunchKernelGGL((gn_silu_head_conv_kernel<f16, 4>), grid, dim3(256), 0, st, p);
    else if (in_dt == RS_F16S) hipLaunchKernelGGL((gn_silu_head_conv_kernel<h2s, 4>), grid, dim3(256), 0, st, p);
    else if (in_dt == RS_F32) hipLaunchKernelGGL((gn_silu_head_conv_kernel<float, 4>), grid, dim3(256), 0, st, p);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int rs_direct_conv_launch(const DirectConvParams* pp, int in_dt, int out_dt, hipStream_t st) {
    const DirectConvParams& p = *pp;
    if (p.up != 1 && p.up != 2) return -2;
    if (in_dt == RS_F16 && out_dt == RS_F16) return launch_direct<f16, f16>(p, st);
    if (in_dt == RS_F16 && out_dt == RS_F32) return launch_direct<f16, float>(p, st);
    if (in_dt == RS_F32 && out_dt == RS_F16) return launch_direct<float, f16>(p, st);
    if (in_dt == RS_F32 && out_dt == RS_F32) return launch_direct<float, float>(p, st);
    if (in_dt == RS_F32 && out_dt == RS_F16S) return launch_direct<float, h2s>(p, st);
    if (in_dt == RS_F16S && out_dt == RS_F16S) return launch_direct<h2s, h2s>(p, st);
    if (in_dt == RS_F16S && out_dt == RS_F32) return launch_direct<h2s, float>(p, st);
    return -2;
}
