// Sampler-side elementwise kernels, layout conversion, bicubic upsampling and the VQ lookup.
//
//   layout:   reference tensors are NCHW fp32 (sampler.py / gaussian_diffusion.py); the engine is NHWC.
//   sampler:  prior_sample gaussian_diffusion.py:517-529; _scale_input :598-609; posterior mean
//             :210-221; p_sample noise add :358-364.
//   bicubic:  F.interpolate(mode='bicubic') gaussian_diffusion.py:503-504 (align_corners=False,
//             A=-0.75, border-clamped taps, no antialias).
//   VQ:       VectorQuantizer2.forward ldm/modules/vqvae/quantize.py:271-312 (expanded-form distance,
//             first-minimum argmin, straight-through expression z + (z_q - z)).
#include "common.h"
#include <algorithm>

namespace {

template <typename TO>
__global__ void nchw_to_nhwc_kernel(const float* in, TO* out, int B, int C, int HW, int ldo, int coff, float scale) {
    const long long n = (long long)B * C * HW;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % C);
        const long long bp = g / C;  // b*HW + pix
        const long long b = bp / HW;
        const long long pix = bp - b * HW;
        rs_st<TO>(out + bp * ldo * Store<TO>::PM + coff + c, ldo, in[(b * C + c) * HW + pix] * scale);
    }
}

template <typename TI>
__global__ void nhwc_to_nchw_kernel(const TI* in, float* out, int B, int C, int HW, int ldi, int coff) {
    const long long n = (long long)B * C * HW;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const long long pix = g % HW;
        const long long bc = g / HW;
        const long long c = bc % C, b = bc / C;
        out[g] = rs_ld<TI>(in + (b * HW + pix) * ldi * Store<TI>::PM + coff + c, ldi);
    }
}

// channel-block copy between NHWC tensors of one storage type (8 channels = one 16-byte chunk per thread; C % 8 == 0)
template <typename T>
__global__ void copy_channels_kernel(const T* src, int lds_, T* dst, int ldd, int C, long long npix) {
    const int nch = C >> 3;
    const long long n = npix * nch;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const long long pix = g / nch;
        const int c = (int)(g - pix * nch) * 8;
        Vec8<T> v;
        v.load(src + pix * lds_ * Store<T>::PM + c, lds_);
        v.store(dst + pix * ldd * Store<T>::PM + c, ldd);
    }
}

// storage conversion [npix][C] -> [npix][C] (dense tensors)
template <typename TI, typename TO>
__global__ void convert_kernel(const TI* src, TO* dst, int C, long long npix) {
    const long long n = npix * C;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const long long pix = g / C;
        const int c = (int)(g - pix * C);
        rs_st<TO>(dst + pix * C * Store<TO>::PM + c, C, rs_ld<TI>(src + pix * C * Store<TI>::PM + c, C));
    }
}

// y = a*x + b*z + c*n   (all fp32, any layout as long as all operands share it)
__global__ void axpbypcz_kernel(const float* x, const float* z, const float* n, float* y, float a, float b, float c, long long cnt) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < cnt; g += (long long)gridDim.x * blockDim.x) {
        float v = a * x[g];
        if (z) v += b * z[g];
        if (n) v += c * n[g];
        y[g] = v;
    }
}

__global__ void clamp_kernel(float* x, float lo, float hi, long long cnt) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < cnt; g += (long long)gridDim.x * blockDim.x)
        x[g] = fminf(fmaxf(x[g], lo), hi);
}

template <typename T>
__global__ void silu_kernel(const T* x, T* y, long long cnt) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < cnt; g += (long long)gridDim.x * blockDim.x)
        y[g] = (T)rs_silu((float)x[g]);
}

// out[r][n] = bias[n] + sum_k act_in(x[r][k]) * w[n][k]   (tiny fp32 linears: time embedding / FiLM tables).
// One wavefront per output: lanes stride over K (coalesced weight reads), fixed-order butterfly reduction.
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, const float* w, const float* bias, float* y, int R, int K, int N,
                                                           int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // output index, one per wave
    if (g >= (long long)R * N) return;
    const int r = (int)(g / N), n = (int)(g - (long long)r * N);
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = x[(long long)r * K + k];
        if (silu_in) xv = xv / (1.0f + expf(-xv));
        acc = fmaf(xv, w[(long long)n * K + k], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        acc += bias ? bias[n] : 0.f;
        if (silu_out) acc = acc / (1.0f + expf(-acc));
        y[g] = acc;
    }
}

// ---- bicubic ----------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// in: NCHW fp32 [B,C,H,W]; out: NHWC TO [B,H*sf,W*sf,ldo] channels [0,C)
template <typename TO>
__global__ void bicubic_up_kernel(const float* in, TO* out, int B, int C, int H, int W, int sf, int ldo) {
    const int Ho = H * sf, Wo = W * sf;
    const long long n = (long long)B * Ho * Wo * C;
    const float A = -0.75f;
    const float rs = 1.0f / (float)sf;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % C);
        long long t = g / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const float fy = ((float)oy + 0.5f) * rs - 0.5f;
        const float fx = ((float)ox + 0.5f) * rs - 0.5f;
        const float fyf = floorf(fy), fxf = floorf(fx);
        const int iy = (int)fyf, ix = (int)fxf;
        const float ty = fy - fyf, tx = fx - fxf;
        float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        const float* src = in + ((long long)b * C + c) * H * W;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), H - 1);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), W - 1);
                r += src[(long long)yy * W + xx] * wx[j];
            }
            acc += r * wy[i];
        }
        rs_st<TO>(out + (((long long)b * Ho + oy) * Wo + ox) * ldo * Store<TO>::PM + c, ldo, acc);
    }
}

// ---- VQ nearest codebook entry ------------------------------------------------
// z: [N][D] fp32 (NHWC latent, D = embed_dim), codebook E: [NE][D].  One thread per token; the
// codebook and its squared norms live in LDS.  d_j = (|z|^2 + |e_j|^2) - 2 z.e_j evaluated in fp32
// in the reference's order; strict '<' keeps the first minimum like torch.argmin.
template <int D>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* z, const float* cb, float* zq, int* idx_out, long long N, int NE) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* e = (float*)smem;        // [NE][D]
    float* ee = e + (long long)NE * D;  // [NE]
    for (int i = threadIdx.x; i < NE * D; i += blockDim.x) e[i] = cb[i];
    __syncthreads();
    for (int j = threadIdx.x; j < NE; j += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = __fadd_rn(s, __fmul_rn(e[j * D + d], e[j * D + d]));
        ee[j] = s;
    }
    __syncthreads();
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    float zv[D];
    float zz = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { zv[d] = z[t * D + d]; zz = __fadd_rn(zz, __fmul_rn(zv[d], zv[d])); }
    float best = 3.4e38f;
    int bi = 0;
    for (int j = 0; j < NE; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) dot = fmaf(zv[d], e[j * D + d], dot);
        const float dist = __fsub_rn(__fadd_rn(zz, ee[j]), __fmul_rn(2.0f, dot));
        if (dist < best) { best = dist; bi = j; }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float q = e[bi * D + d];
        zq[t * D + d] = __fadd_rn(zv[d], __fsub_rn(q, zv[d]));  // z + (z_q - z), quantize.py:298
    }
    if (idx_out) idx_out[t] = bi;
}

// ---- overlap-average tiling (utils/util_image.py:889-979 ImageSpliterTh.update / gather) -----------------------------
// acc[b,c,h0+y,w0+x] += tile[b,c,y,x];  count[h0+y,w0+x] += 1 (one plane: the reference's per-(b,c) counts are all equal)
__global__ void tile_accumulate_kernel(float* acc, float* count, const float* tile, int B, int C, int H, int W, int h0, int w0, int th,
                                       int tw) {
    const long long n = (long long)B * C * th * tw;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(g % tw);
        long long t = g / tw;
        const int y = (int)(t % th);
        const long long bc = t / th;
        acc[(bc * H + h0 + y) * W + w0 + x] += tile[g];
        if (bc == 0) count[(long long)(h0 + y) * W + w0 + x] += 1.0f;
    }
}
__global__ void tile_finalize_kernel(float* acc, const float* count, long long BC, long long HW) {
    const long long n = BC * HW;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x)
        acc[g] = acc[g] / count[g % HW];
}

inline unsigned nblk(long long n, int bs = 256, long long cap = 65536) { return (unsigned)std::min<long long>((n + bs - 1) / bs, cap); }

}  // namespace

// ---------------------------------------------------------------- uint8 pre / post processing on the device
// pre:  datapipe/datasets.py:59-63 (ToTensor + Normalize(0.5, 0.5)): interleaved uint8 HWC -> planar fp32 in [-1, 1].
// post: sampler.py:218-222 + utils/util_image.py:245-269 (tensor2img): x*0.5+0.5, optional inpainting blend
//       sr*m + lq*(1-m), clamp to [0,1], *255, round half to even, optional RGB->BGR, planar fp32 -> interleaved uint8.
// The arithmetic is spelled with the non-contracting intrinsics so that it rounds exactly like the reference's
// separate torch / numpy ops (no fused multiply-add).
__global__ void u8_to_input_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long npix, int HW, int C) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {   // pixel of B*H*W
        const long long b = i / HW, hw = i - b * HW;
        for (int c = 0; c < C; ++c) {
            const float v = __fdiv_rn((float)src[i * C + c], 255.0f);
            dst[(b * C + c) * HW + hw] = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
        }
    }
}

__global__ void output_to_u8_kernel(const float* __restrict__ sr, const float* __restrict__ lq, const float* __restrict__ mask,
                                    unsigned char* __restrict__ dst, long long npix, int HW, int C, int bgr) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW, hw = i - b * HW;
        float m = 1.0f;
        if (mask) m = __fadd_rn(__fmul_rn(mask[b * HW + hw], 0.5f), 0.5f);
        for (int c = 0; c < C; ++c) {
            float v = __fadd_rn(__fmul_rn(sr[(b * C + c) * HW + hw], 0.5f), 0.5f);
            if (mask) {
                const float l = __fadd_rn(__fmul_rn(lq[(b * C + c) * HW + hw], 0.5f), 0.5f);
                v = __fadd_rn(__fmul_rn(v, m), __fmul_rn(l, __fsub_rn(1.0f, m)));
            }
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            const int oc = (bgr && C == 3) ? 2 - c : c;
            dst[i * C + oc] = (unsigned char)rintf(__fmul_rn(v, 255.0f));
        }
    }
}

template <typename TI>
static int convert_from(const TI* src, void* dst, int dst_dt, int C, long long npix, hipStream_t st) {
    const long long n = npix * C;
    if (dst_dt == RS_F16) hipLaunchKernelGGL((convert_kernel<TI, f16>), dim3(nblk(n)), dim3(256), 0, st, src, (f16*)dst, C, npix);
    else if (dst_dt == RS_F16S) hipLaunchKernelGGL((convert_kernel<TI, h2s>), dim3(nblk(n)), dim3(256), 0, st, src, (h2s*)dst, C, npix);
    else if (dst_dt == RS_F32) hipLaunchKernelGGL((convert_kernel<TI, float>), dim3(nblk(n)), dim3(256), 0, st, src, (float*)dst, C, npix);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// out[pl][i][j] = scale * in[pl][refl(h0 + i, H)][refl(w0 + j, W)] on fp32 planes, refl(i, n) = i < n ? i : 2 (n - 1) - i.
// One kernel for the data movement the host mirror needs around the networks: bottom / right reflect padding of the LQ batch
// (sampler.py:130-138, F.pad mode 'reflect'), the tile crop of the tiled path (util_image.py:946-952) and the latent scaling
// of encode_first_stage (gaussian_diffusion.py:514).
__global__ void window_copy_kernel(const float* in, float* out, long long planes, int H, int W, int h0, int w0, int Ho, int Wo, float scale) {
    const long long n = planes * Ho * Wo;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(g % Wo);
        const long long t = g / Wo;
        const int i = (int)(t % Ho);
        const long long pl = t / Ho;
        int y = h0 + i, x = w0 + j;
        y = y < H ? y : 2 * (H - 1) - y;
        x = x < W ? x : 2 * (W - 1) - x;
        out[g] = scale * in[(pl * H + y) * W + x];
    }
}

extern "C" {

int rs_nchw_to_nhwc_launch(const float* in, void* out, int out_dt, int B, int C, int HW, int ldo, int coff, float scale, hipStream_t st) {
    const long long n = (long long)B * C * HW;
    if (out_dt == RS_F16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16>), dim3(nblk(n)), dim3(256), 0, st, in, (f16*)out, B, C, HW, ldo, coff, scale);
    else if (out_dt == RS_F16S) hipLaunchKernelGGL((nchw_to_nhwc_kernel<h2s>), dim3(nblk(n)), dim3(256), 0, st, in, (h2s*)out, B, C, HW, ldo, coff, scale);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(nblk(n)), dim3(256), 0, st, in, (float*)out, B, C, HW, ldo, coff, scale);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_nhwc_to_nchw_launch(const void* in, int in_dt, float* out, int B, int C, int HW, int ldi, int coff, hipStream_t st) {
    const long long n = (long long)B * C * HW;
    if (in_dt == RS_F16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16>), dim3(nblk(n)), dim3(256), 0, st, (const f16*)in, out, B, C, HW, ldi, coff);
    else if (in_dt == RS_F16S) hipLaunchKernelGGL((nhwc_to_nchw_kernel<h2s>), dim3(nblk(n)), dim3(256), 0, st, (const h2s*)in, out, B, C, HW, ldi, coff);
    else hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), dim3(nblk(n)), dim3(256), 0, st, (const float*)in, out, B, C, HW, ldi, coff);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_copy_channels_launch(const void* src, int lds_, void* dst, int ldd, int C, long long npix, int dt, hipStream_t st) {
    if ((C % 8) || (lds_ % 8) || (ldd % 8)) return -2;
    const long long n = npix * (C / 8);
    if (dt == RS_F16) hipLaunchKernelGGL((copy_channels_kernel<f16>), dim3(nblk(n)), dim3(256), 0, st, (const f16*)src, lds_, (f16*)dst, ldd, C, npix);
    else if (dt == RS_F16S) hipLaunchKernelGGL((copy_channels_kernel<h2s>), dim3(nblk(n)), dim3(256), 0, st, (const h2s*)src, lds_, (h2s*)dst, ldd, C, npix);
    else hipLaunchKernelGGL((copy_channels_kernel<float>), dim3(nblk(n)), dim3(256), 0, st, (const float*)src, lds_, (float*)dst, ldd, C, npix);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_convert_launch(const void* src, int src_dt, void* dst, int dst_dt, int C, long long npix, hipStream_t st) {
    if (src_dt == RS_F16) return convert_from((const f16*)src, dst, dst_dt, C, npix, st);
    if (src_dt == RS_F16S) return convert_from((const h2s*)src, dst, dst_dt, C, npix, st);
    if (src_dt == RS_F32) return convert_from((const float*)src, dst, dst_dt, C, npix, st);
    return -2;
}

int rs_axpbypcz_launch(const float* x, const float* z, const float* n, float* y, float a, float b, float c, long long cnt, hipStream_t st) {
    hipLaunchKernelGGL(axpbypcz_kernel, dim3(nblk(cnt)), dim3(256), 0, st, x, z, n, y, a, b, c, cnt);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_clamp_launch(float* x, float lo, float hi, long long cnt, hipStream_t st) {
    hipLaunchKernelGGL(clamp_kernel, dim3(nblk(cnt)), dim3(256), 0, st, x, lo, hi, cnt);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_silu_launch(const void* x, void* y, int dt, long long cnt, hipStream_t st) {
    if (dt == RS_F16) hipLaunchKernelGGL((silu_kernel<f16>), dim3(nblk(cnt)), dim3(256), 0, st, (const f16*)x, (f16*)y, cnt);
    else hipLaunchKernelGGL((silu_kernel<float>), dim3(nblk(cnt)), dim3(256), 0, st, (const float*)x, (float*)y, cnt);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_small_linear_launch(const float* x, const float* w, const float* bias, float* y, int R, int K, int N, int silu_in, int silu_out,
                           hipStream_t st) {
    hipLaunchKernelGGL(small_linear_kernel, dim3((unsigned)(((long long)R * N + 3) / 4)), dim3(256), 0, st, x, w, bias, y, R, K, N, silu_in, silu_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_bicubic_launch(const float* in, void* out, int out_dt, int B, int C, int H, int W, int sf, int ldo, hipStream_t st) {
    const long long n = (long long)B * H * sf * W * sf * C;
    if (out_dt == RS_F16) hipLaunchKernelGGL((bicubic_up_kernel<f16>), dim3(nblk(n)), dim3(256), 0, st, in, (f16*)out, B, C, H, W, sf, ldo);
    else if (out_dt == RS_F16S) hipLaunchKernelGGL((bicubic_up_kernel<h2s>), dim3(nblk(n)), dim3(256), 0, st, in, (h2s*)out, B, C, H, W, sf, ldo);
    else hipLaunchKernelGGL((bicubic_up_kernel<float>), dim3(nblk(n)), dim3(256), 0, st, in, (float*)out, B, C, H, W, sf, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_tile_accumulate(float* acc, float* count, const float* tile, int B, int C, int H, int W, int h0, int w0, int th, int tw,
                       void* stream) {
    if (h0 < 0 || w0 < 0 || h0 + th > H || w0 + tw > W) return -2;
    const long long n = (long long)B * C * th * tw;
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, acc, count, tile, B, C, H, W, h0, w0, th, tw);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_window_copy(const float* in, float* out, long long planes, int H, int W, int h0, int w0, int Ho, int Wo, float scale, void* stream) {
    // (one reflection at most, as torch: the window may overhang the plane by less than its size)
    if (planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || h0 < 0 || w0 < 0 || h0 + Ho > 2 * H - 1 || w0 + Wo > 2 * W - 1 || h0 >= H || w0 >= W)
        return -2;
    const long long n = planes * Ho * Wo;
    hipLaunchKernelGGL(window_copy_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in, out, planes, H, W, h0, w0, Ho, Wo, scale);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_tile_finalize(float* acc, const float* count, int B, int C, int H, int W, void* stream) {
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(tile_finalize_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, acc, count, (long long)B * C, (long long)H * W);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_u8_to_input(const void* src_u8_nhwc, float* dst_f32_nchw, int B, int H, int W, int C, void* stream) {
    const long long npix = (long long)B * H * W;
    hipLaunchKernelGGL(u8_to_input_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8_nhwc, dst_f32_nchw,
                       npix, H * W, C);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_output_to_u8(const float* sr_f32_nchw, const float* lq_f32_nchw, const float* mask_f32_n1hw, void* dst_u8_nhwc, int B, int H, int W,
                    int C, int bgr, void* stream) {
    if ((mask_f32_n1hw != nullptr) != (lq_f32_nchw != nullptr)) return -2;
    const long long npix = (long long)B * H * W;
    hipLaunchKernelGGL(output_to_u8_kernel, dim3(nblk(npix)), dim3(256), 0, (hipStream_t)stream, sr_f32_nchw, lq_f32_nchw, mask_f32_n1hw,
                       (unsigned char*)dst_u8_nhwc, npix, H * W, C, bgr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int rs_vq_launch(const float* z, const float* codebook, float* zq, int* idx, long long N, int NE, int D, hipStream_t st) {
    const size_t lds = (size_t)NE * (D + 1) * sizeof(float);
    if (lds > 160 * 1024) return -2;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (D == 3) {
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((vq_nearest_kernel<3>), dim3(blocks), dim3(256), lds, st, z, codebook, zq, idx, N, NE);
    } else if (D == 4) {
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((vq_nearest_kernel<4>), dim3(blocks), dim3(256), lds, st, z, codebook, zq, idx, N, NE);
    } else if (D == 8) {
        (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((vq_nearest_kernel<8>), dim3(blocks), dim3(256), lds, st, z, codebook, zq, idx, N, NE);
    } else {
        return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
