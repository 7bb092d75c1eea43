// Implicit-GEMM convolution / linear / batched-GEMM kernel for gfx950 (CDNA4).
//
// Replaces every torch conv2d / linear / bmm call on the ResShift hot path:
//   models/unet.py:147,173,69,99,184,707,862 (ResBlock / Up / Downsample / skip / in / out convs)
//   models/swin_transformer.py:22,24,105,107,480,515 (MLP 1x1 convs, qkv/proj linears, patch (un)embed)
//   ldm/modules/diffusionmodules/model.py:55,80,100,110,123,158-177,191,199 (AE convs, q/k/v/proj, QK^T, PV)
//
// Formulation: D[n][m] = sum_k W[n][k] * X[m][k]   (n = output channel, m = output pixel)
//   * X is never materialised: each 16-byte chunk of a K-row is gathered straight from the
//     NHWC source(s) with tap / channel arithmetic (zero fill outside the image, optional
//     nearest-x2 upsample folded into the address, optional second source = channel concat).
//   * The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so each
//     lane ends up holding 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
//   * Tile: BP pixels x BC channels x 128 bytes of K per stage, 256 threads = 4 waves (2x2),
//     LDS double buffered, one barrier per K stage, XOR-swizzled 16-byte chunks so that the
//     ds_read_b128 fragment reads are (at most 2-way) conflict free.
//   * f16 storage -> v_mfma_f32_16x16x32_f16; f32 storage -> v_mfma_f32_16x16x4_f32 (exact fp32).
#include "common.h"

namespace {

template <typename T> struct MfmaOps;

template <> struct MfmaOps<f16> {
    static constexpr int CH = 8;  // elements per 16-byte chunk
    // one K stage = 8 chunks = 64 halfs = 2 MFMA k-steps of 32
    template <int FC, int FP>
    static __device__ __forceinline__ void stage(const char* ws, const char* xs, int wrow0, int xrow0, int lr, int lg,
                                                 f32x4 (&acc)[FC][FP]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + lg;
            f16x8 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int r = wrow0 + i * 16 + lr;
                a[i] = *(const f16x8*)(ws + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int r = xrow0 + j * 16 + lr;
                b[j] = *(const f16x8*)(xs + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <> struct MfmaOps<float> {
    static constexpr int CH = 4;
    // one K stage = 8 chunks = 32 floats.  Lane group lg reads chunk ss*4+lg (4 floats) and feeds
    // element s to MFMA step s: the k permutation is identical for both operands, so the sum is exact.
    template <int FC, int FP>
    static __device__ __forceinline__ void stage(const char* ws, const char* xs, int wrow0, int xrow0, int lr, int lg,
                                                 f32x4 (&acc)[FC][FP]) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int chunk = ss * 4 + lg;
            f32x4 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int r = wrow0 + i * 16 + lr;
                a[i] = *(const f32x4*)(ws + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int r = xrow0 + j * 16 + lr;
                b[j] = *(const f32x4*)(xs + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename TO> struct Out4;
template <> struct Out4<f16> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[4]) {
        f16x4 t = *(const f16x4*)p;
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[4]) {
        f16x4 t; t[0] = (f16)v[0]; t[1] = (f16)v[1]; t[2] = (f16)v[2]; t[3] = (f16)v[3];
        *(f16x4*)p = t;
    }
};
template <> struct Out4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        *(f32x4*)p = t;
    }
};

template <typename TI, typename TO, int BP, int BC>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmParams p) {
    constexpr int CH = MfmaOps<TI>::CH;
    constexpr int BK = 8 * CH;         // elements of K per stage (128 bytes)
    constexpr int XR = BP / 32;        // pixel rows staged per thread
    constexpr int WR = BC / 32;        // weight rows staged per thread
    constexpr int FP = BP / 32;        // 16-wide pixel fragments per wave (wave covers BP/2 pixels)
    constexpr int FC = BC / 32;        // 16-wide channel fragments per wave
    static_assert(BP % 32 == 0 && BC % 32 == 0, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs_base = smem;                       // [2][BP][128]
    char* ws_base = smem + 2 * BP * 128;        // [2][BC][128]

    const int tid = threadIdx.x;
    const int kc = tid & 7;
    const int r0 = tid >> 3;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 1, wc = wave >> 1;

    const int m0 = blockIdx.x * BP;
    const int n0 = blockIdx.y * BC;
    const long long z = blockIdx.z;

    const TI* x0 = (const TI*)p.x0 + z * p.bs_x0;
    const TI* x1 = (const TI*)p.x1;
    const TI* w = (const TI*)p.w + z * p.bs_w;

    const int Ctot = p.C0 + p.C1;
    const int ntaps = p.KH * p.KW;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const int HoWo = p.Ho * p.Wo;

    // per-thread pixel-row bookkeeping
    int pixbase[XR], iy0[XR], ix0[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            pixbase[i] = b * p.Hs * p.Ws;
            iy0[i] = oy * p.stride - p.pad_t;
            ix0[i] = ox * p.stride - p.pad_l;
        } else {
            pixbase[i] = -1; iy0[i] = 0; ix0[i] = 0;
        }
    }
    // K position of this thread's chunk
    int kk = kc * CH;              // absolute k of the chunk
    int tap = kk / Ctot;
    int cc = kk - tap * Ctot;

    uint4 xreg[XR], wreg[WR];

    auto gload = [&]() {
        const bool kvalid = tap < ntaps;
        const int ky = tap / p.KW;
        const int kx = tap - ky * p.KW;
        const TI* src; int ld, c;
        if (cc < p.C0) { src = x0; ld = p.ld0; c = cc; } else { src = x1; ld = p.ld1; c = cc - p.C0; }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kvalid && pixbase[i] >= 0 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const long long pix = (long long)pixbase[i] + (long long)(iy >> ush) * p.Ws + (ix >> ush);
                v = *(const uint4*)(src + pix * ld + c);
            }
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int n = n0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n < p.Cout && kk < p.Ktot) v = *(const uint4*)(w + (long long)n * p.Ktot + kk);
            wreg[i] = v;
        }
    };
    auto advance = [&]() {
        kk += BK;
        cc += BK;
        while (cc >= Ctot) { cc -= Ctot; ++tap; }
    };
    auto lds_write = [&](int buf) {
        char* xs = xs_base + buf * BP * 128;
        char* ws = ws_base + buf * BC * 128;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int r = r0 + 32 * i;
            *(uint4*)(xs + r * 128 + ((kc ^ (r & 7)) << 4)) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int r = r0 + 32 * i;
            *(uint4*)(ws + r * 128 + ((kc ^ (r & 7)) << 4)) = wreg[i];
        }
    };

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.Ktot + BK - 1) / BK;
    gload();
    lds_write(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { advance(); gload(); }
        MfmaOps<TI>::template stage<FC, FP>(ws_base + cur * BC * 128, xs_base + cur * BP * 128, wc * (BC / 2),
                                            wp * (BP / 2), lr, lg, acc);
        if (kt + 1 < nk) lds_write(cur ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds channels n..n+3 of pixel m for each fragment
    TO* y = (TO*)p.y + z * p.bs_y;
    const TO* res = p.res ? (const TO*)p.res + z * p.bs_res : nullptr;
    const bool vec_ok = ((p.ldy & 3) == 0) && (!res || (p.ldres & 3) == 0);
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int m = m0 + wp * (BP / 2) + j * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
            if (n >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[i][j][r] * p.out_scale;
                if (p.bias && n + r < p.Cout) t += p.bias[n + r];
                v[r] = rs_apply_act(t, p.act);
            }
            TO* yp = y + (long long)m * p.ldy + n;
            if (n + 3 < p.Cout && vec_ok) {
                if (res) {
                    float rv[4];
                    Out4<TO>::load(res + (long long)m * p.ldres + n, rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                }
                Out4<TO>::store(yp, v);
            } else {
                for (int r = 0; r < 4 && n + r < p.Cout; ++r) {
                    float t = v[r];
                    if (res) t += (float)res[(long long)m * p.ldres + n + r];
                    yp[r] = (TO)t;
                }
            }
        }
    }
}

template <typename TI, typename TO, int BP, int BC>
hipError_t launch_cfg(const IGemmParams& p, int nz, hipStream_t st) {
    dim3 grid((p.M + BP - 1) / BP, (p.Cout + BC - 1) / BC, nz);
    const size_t lds = 2 * (BP + BC) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<TI, TO, BP, BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_kernel<TI, TO, BP, BC>), grid, dim3(256), lds, st, p);
    return hipGetLastError();
}

template <typename TI, typename TO>
hipError_t launch_t(const IGemmParams& p, int nz, hipStream_t st) {
    // channel-tile selection: the model's Cout values are multiples of 160, 192 or 128.
    const int n = p.Cout;
    auto waste = [&](int bc) { return ((n + bc - 1) / bc) * bc - n; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    if (n <= 64) best = 64;
    const bool small_m = p.M <= 64 * 48;  // few pixels: use the 64-pixel tile to get more workgroups
    switch (best) {
        case 64: return small_m ? launch_cfg<TI, TO, 64, 64>(p, nz, st) : launch_cfg<TI, TO, 128, 64>(p, nz, st);
        case 160: return small_m ? launch_cfg<TI, TO, 64, 160>(p, nz, st) : launch_cfg<TI, TO, 128, 160>(p, nz, st);
        case 192: return small_m ? launch_cfg<TI, TO, 64, 192>(p, nz, st) : launch_cfg<TI, TO, 128, 192>(p, nz, st);
        default: return small_m ? launch_cfg<TI, TO, 64, 128>(p, nz, st) : launch_cfg<TI, TO, 128, 128>(p, nz, st);
    }
}

}  // namespace

// in_dt: storage type of x/w; out_dt: storage type of y/res.  Supported: (F16,F16) (F16,F32) (F32,F32).
// Requirements: (C0+C1) and C0 multiples of the 16-byte chunk (8 halfs / 4 floats); ld0/ld1 likewise;
// source base pointers 16-byte aligned.
extern "C" int rs_igemm_launch(const IGemmParams* pp, int in_dt, int out_dt, int nz, hipStream_t st) {
    const IGemmParams& p = *pp;
    const int ch = in_dt == RS_F16 ? 8 : 4;
    if ((p.C0 % ch) || (p.C1 % ch) || (p.ld0 % ch) || (p.C1 && (p.ld1 % ch)) || p.M <= 0 || p.Cout <= 0) return -2;
    if (p.up != 1 && p.up != 2) return -2;
    hipError_t e;
    if (in_dt == RS_F16 && out_dt == RS_F16) e = launch_t<f16, f16>(p, nz, st);
    else if (in_dt == RS_F16 && out_dt == RS_F32) e = launch_t<f16, float>(p, nz, st);
    else if (in_dt == RS_F32 && out_dt == RS_F32) e = launch_t<float, float>(p, nz, st);
    else return -2;
    return e == hipSuccess ? 0 : -1;
}
