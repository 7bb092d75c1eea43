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
