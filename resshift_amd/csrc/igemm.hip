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
//     lane ends up holding 4 consecutive output channels of one pixel.
//   * Tile: BP pixels x BC channels x 128 bytes of K per stage, 256 threads = 4 waves (2x2),
//     LDS double buffered, one barrier per K stage, XOR-swizzled 16-byte chunks so that the
//     ds_read_b128 fragment reads are (at most 2-way) conflict free.
//   * f16 storage -> v_mfma_f32_16x16x32_f16; f32 storage -> v_mfma_f32_16x16x4_f32 (exact fp32).
//   * Workgroup -> tile map is XCD aware: the 8 XCDs (private L2s) each get a contiguous run of
//     tiles, channel tiles of one pixel tile are adjacent, so the gathered X rows are L2 hits.
//   * Epilogue: scale, bias, GELU, residual in fp32 registers; fp16 results are transposed through
//     LDS so that every lane stores 16 contiguous bytes (full 128..192-byte pixel rows per wave).
//   * Split-K (grid.z) for launches with too few tiles to fill 256 CUs: fp32 partial slabs + a
//     deterministic reduce kernel that applies the same epilogue.
#include "igemm_common.h"
#include <algorithm>

namespace {

using namespace igemm_detail;

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

    const int nby = (p.Cout + BC - 1) / BC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nby) * BP;
    const int n0 = (tile % nby) * BC;
    const long long z = blockIdx.z;
    const bool split = p.splitk > 1;

    const TI* x0 = (const TI*)p.x0 + (split ? 0 : z * p.bs_x0);
    const TI* x1 = (const TI*)p.x1;
    const TI* w = (const TI*)p.w + (split ? 0 : z * p.bs_w);

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
    // K range of this workgroup (split-K: grid.z slices the K stages)
    const int nk_total = (p.Ktot + BK - 1) / BK;
    int kt0 = 0, kt1 = nk_total;
    if (split) {
        const int per = (nk_total + p.splitk - 1) / p.splitk;
        kt0 = min(nk_total, (int)z * per);
        kt1 = min(nk_total, kt0 + per);
    }
    // K position of this thread's chunk
    int kk = kt0 * BK + kc * CH;   // absolute k of the chunk
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

    if (kt0 < kt1) {
        gload();
        lds_write(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) { advance(); gload(); }
        MfmaOps<TI>::template stage<FC, FP>(ws_base + cur * BC * 128, xs_base + cur * BP * 128, wc * (BC / 2),
                                            wp * (BP / 2), lr, lg, acc);
        if (kt + 1 < kt1) lds_write(cur ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    // lane holds channels n..n+3 of pixel m for each fragment
    if (split) {
        // raw fp32 partial sums; scale / bias / activation / residual are applied by splitk_reduce_kernel
        float* part = p.partial + z * (long long)p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / 2) + j * 16 + lr;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
                if (n >= p.Cout) continue;
                float* pp = part + (long long)m * p.Cout + n;
                if (n + 3 < p.Cout && (p.Cout & 3) == 0) {
                    *(f32x4*)pp = acc[i][j];
                } else {
                    for (int r = 0; r < 4 && n + r < p.Cout; ++r) pp[r] = acc[i][j][r];
                }
            }
        }
        return;
    }
    TO* y = (TO*)p.y + z * p.bs_y;
    const TO* res = p.res ? (const TO*)p.res + z * p.bs_res : nullptr;
    const bool res_vec = res && (p.ldres & 3) == 0;
    if constexpr (sizeof(TO) == 2) {
        // fp16: finish the math in registers, transpose the wave's (BP/2 x BC/2) tile through LDS, store 16 B per lane
        constexpr int ROWB = (BC / 2) * 2 + 16;  // padded row pitch in bytes
        char* stg = smem + wave * (BP / 2) * ROWB;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / 2) + j * 16 + lr;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int nl = i * 16 + lg * 4;
                const int n = n0 + wc * (BC / 2) + nl;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] * p.out_scale;
                    if (p.bias && n + r < p.Cout) t += p.bias[n + r];
                    v[r] = epi_act<TO>(t, p.act);
                }
                if (res && m < p.M && n < p.Cout) {
                    if (res_vec && n + 3 < p.Cout) {
                        float rv[4];
                        Out4<TO>::load(res + (long long)m * p.ldres + n, rv);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    } else {
                        for (int r = 0; r < 4 && n + r < p.Cout; ++r) v[r] += (float)res[(long long)m * p.ldres + n + r];
                    }
                }
                f16x4 h; h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)v[3];
                *(f16x4*)(stg + (j * 16 + lr) * ROWB + nl * 2) = h;
            }
        }
        __syncthreads();
        constexpr int CPR = (BC / 2) / 8;           // 16-byte chunks per row of the wave tile
        constexpr int NITEM = (BP / 2) * CPR;
        const bool vec_ok = (p.ldy & 7) == 0;
        for (int idx = lane; idx < NITEM; idx += 64) {
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int m = m0 + wp * (BP / 2) + row;
            const int n = n0 + wc * (BC / 2) + c8 * 8;
            if (m >= p.M || n >= p.Cout) continue;
            const uint4 v = *(const uint4*)(stg + row * ROWB + c8 * 16);
            TO* yp = y + rs_out_m(p, m) * p.ldy + n;
            if (vec_ok && n + 7 < p.Cout) {
                *(uint4*)yp = v;
            } else {
                const f16x8 hv = __builtin_bit_cast(f16x8, v);   // (no address-of: a pointer into `v` would park it in scratch memory)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (n + r < p.Cout) yp[r] = (TO)hv[r];
            }
        }
    } else {
        const bool vec_ok = ((p.ldy & 3) == 0);
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
                    v[r] = epi_act<TO>(t, p.act);
                }
                TO* yp = y + rs_out_m(p, m) * p.ldy + n;
                if (n + 3 < p.Cout && vec_ok && (!res || res_vec)) {
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
}

// y[m][n] = act(scale * sum_z partial[z][m][n] + bias[n]) + res[m][n]; fixed summation order -> deterministic
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(IGemmParams p) {
    constexpr int PM = Store<TO>::PM;   // split storage: pixel record = 2 * ld halfs
    const long long nq = (long long)p.M * (p.Cout >> 2);  // quads of 4 channels (Cout % 4 == 0 is guaranteed by the planner)
    const long long slab = (long long)p.M * p.Cout;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
        const long long m = q / (p.Cout >> 2);
        const int n = (int)(q - m * (p.Cout >> 2)) * 4;
        f32x4 s = *(const f32x4*)(p.partial + m * p.Cout + n);
        for (int zz = 1; zz < p.splitk; ++zz) {
            const f32x4 t = *(const f32x4*)(p.partial + zz * slab + m * p.Cout + n);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = s[r] * p.out_scale;
            if (p.bias) t += p.bias[n + r];
            v[r] = epi_act<TO>(t, p.act);
        }
        if (p.res) {
            const TO* rp = (const TO*)p.res + m * p.ldres * PM + n;
            if ((p.ldres & 3) == 0) {
                float rv[4];
                Out4<TO>::load(rp, rv, p.ldres);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rv[r];
            } else {
                for (int r = 0; r < 4; ++r) v[r] += rs_ld<TO>(rp + r, p.ldres);
            }
        }
        TO* yp = (TO*)p.y + m * p.ldy * PM + n;
        if ((p.ldy & 3) == 0) Out4<TO>::store(yp, v, p.ldy);
        else for (int r = 0; r < 4; ++r) rs_st<TO>(yp + r, p.ldy, v[r]);
    }
}

void pick_tile(int M, int Cout, int& BP, int& BC) {
    // channel-tile selection: the model's Cout values are multiples of 160, 192 or 128.
    auto waste = [&](int bc) { return ((Cout + bc - 1) / bc) * bc - Cout; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    if (Cout <= 64) best = 64;
    BC = best;
    BP = (M <= 64 * 48) ? 64 : 128;  // few pixels: use the 64-pixel tile to get more workgroups
}

template <typename TI, typename TO, int BP, int BC>
hipError_t launch_cfg(const IGemmParams& p, int nz, hipStream_t st) {
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    dim3 grid(tiles, 1, p.splitk > 1 ? p.splitk : nz);
    const size_t lds = 2 * (BP + BC) * 128;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<TI, TO, BP, BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((igemm_kernel<TI, TO, BP, BC>), grid, dim3(256), lds, st, p);
    if (p.splitk > 1) {
        const long long nq = (long long)p.M * (p.Cout >> 2);
        const unsigned blocks = (unsigned)std::min<long long>((nq + 255) / 256, 4096);
        hipLaunchKernelGGL((splitk_reduce_kernel<TO>), dim3(blocks), dim3(256), 0, st, p);
    }
    return hipGetLastError();
}

template <typename TI, typename TO>
hipError_t launch_t(const IGemmParams& p, int nz, hipStream_t st) {
    int BP, BC;
    pick_tile(p.M, p.Cout, BP, BC);
    const bool small_m = BP == 64;
    switch (BC) {
        case 64: return small_m ? launch_cfg<TI, TO, 64, 64>(p, nz, st) : launch_cfg<TI, TO, 128, 64>(p, nz, st);
        case 160: return small_m ? launch_cfg<TI, TO, 64, 160>(p, nz, st) : launch_cfg<TI, TO, 128, 160>(p, nz, st);
        case 192: return small_m ? launch_cfg<TI, TO, 64, 192>(p, nz, st) : launch_cfg<TI, TO, 128, 192>(p, nz, st);
        default: return small_m ? launch_cfg<TI, TO, 64, 128>(p, nz, st) : launch_cfg<TI, TO, 128, 128>(p, nz, st);
    }
}

}  // namespace

extern "C" int rs_igemm2_pick(int M, int Cout, int Kbytes, int nz, int* BP, int* BC);
extern "C" int rs_igemm2_launch(const IGemmParams* pp, int in_dt, int out_dt, int BP, int BC, int nz, hipStream_t st);
extern "C" int rs_igemm3_pick(int M, int Cout, int Ktot, int in_dt, int nz, int splitk, int* BC);
extern "C" int rs_igemm3_launch(const IGemmParams* pp, int out_dt, int BC, hipStream_t st);
extern "C" int rs_igemm4_pick(const IGemmParams* pp, int in_dt, int out_dt, int nz, int* TW, int* BC);
extern "C" int rs_igemm4_launch(const IGemmParams* pp, int in_dt, int TW, int BC, hipStream_t st);
extern "C" int rs_wino_plan(const IGemmParams* pp, int in_dt, int out_dt, int nz);
extern "C" int rs_wino_launch(const IGemmParams* pp, hipStream_t st);
extern "C" void rs_igemm_split_pick(int M, int Cout, int nz, int* BP, int* BC);
extern "C" int rs_igemm_split_launch(const IGemmParams* pp, int out_dt, int nz, hipStream_t st);

extern "C" int rs_splitk_reduce_launch(const IGemmParams* pp, int out_dt, hipStream_t st) {
    const IGemmParams& p = *pp;
    const long long nq = (long long)p.M * (p.Cout >> 2);
    const unsigned blocks = (unsigned)std::min<long long>((nq + 255) / 256, 4096);
    if (out_dt == RS_F16) hipLaunchKernelGGL((splitk_reduce_kernel<f16>), dim3(blocks), dim3(256), 0, st, p);
    else if (out_dt == RS_F16S) hipLaunchKernelGGL((splitk_reduce_kernel<h2s>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(blocks), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Split-K planner: returns the number of K slices (1 = no split) for a single (non-batched) launch.  The caller owns the
// fp32 workspace of splitk * M * Cout floats (IGemmParams::partial).
extern "C" int rs_igemm_splitk_plan(int M, int Cout, int Ktot, int in_dt) {
    int BP, BC;
    if (in_dt == RS_F16S) {   // split storage (igemm_split.hip): one workgroup per CU, so aim at one round of workgroups
        rs_igemm_split_pick(M, Cout, 1, &BP, &BC);
        const int tiles = ((M + BP - 1) / BP) * ((Cout + BC - 1) / BC);
        const int nk = (Ktot + 63) / 64;
        if (tiles >= 200 || nk < 16 || (Cout & 3)) return 1;
        int s = std::min((256 + tiles - 1) / tiles, 16);
        s = std::min(s, nk / 8);
        return std::max(s, 1);
    }
    pick_tile(M, Cout, BP, BC);
    if (Cout > 64) {          // igemm2 takes every single-source launch with more than 64 channels: ask it for its tile
        int bp2 = 0, bc2 = 0;
        if (rs_igemm2_pick(M, Cout, Ktot * (in_dt == RS_F16 ? 2 : 4), 1, &bp2, &bc2)) { BP = bp2 == 133 ? 64 : (bp2 == 256 || bp2 == 131) ? 256 : 128; BC = bc2; }
    }
    const int tiles = ((M + BP - 1) / BP) * ((Cout + BC - 1) / BC);
    const int bk = in_dt == RS_F16 ? 64 : 32;
    const int nk = (Ktot + bk - 1) / bk;
    if (tiles >= (BP == 64 ? 400 : 200) || nk < 16 || (Cout & 3)) return 1;
    // aim at ~3 workgroups per CU, keep >= 6 K stages per slice (RS_SPLITK_TARGET / RS_SPLITK_MINSTAGES override for tuning)
    static const int target = []() { const char* e = getenv("RS_SPLITK_TARGET"); return e ? atoi(e) : 512; }();
    static const int minst = []() { const char* e = getenv("RS_SPLITK_MINSTAGES"); return e ? atoi(e) : 8; }();
    int s = (target + tiles - 1) / tiles;
    s = std::min(s, 16);
    s = std::min(s, nk / minst);
    return std::max(s, 1);
}

// in_dt: storage type of x/w; out_dt: storage type of y/res.  Supported: (F16,F16) (F16,F32) (F32,F32) (F16S,F16S) (F16S,F32).
// Requirements: (C0+C1) and C0 multiples of the 16-byte chunk (8 halfs / 4 floats); ld0/ld1 likewise;
// source base pointers 16-byte aligned.  p.splitk > 1 requires nz == 1 and p.partial.
extern "C" int rs_igemm_launch(const IGemmParams* pp, int in_dt, int out_dt, int nz, hipStream_t st) {
    IGemmParams p = *pp;
    const int ch = in_dt == RS_F32 ? 4 : 8;
    if ((p.C0 % ch) || (p.C1 % ch) || (p.ld0 % ch) || (p.C1 && (p.ld1 % ch)) || p.M <= 0 || p.Cout <= 0) return -2;
    if (p.up != 1 && p.up != 2) return -2;
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > 1 && (nz != 1 || !p.partial || (p.Cout & 3))) return -2;
    // scattered rows (sub-pixel form of an upsampling conv, IGemmParams::osc): plain single launches only - the residual, the output
    // statistics and the split-K slabs are all indexed by the GEMM row
    // (output statistics: the split-storage kernel only - its four launches fill one slab array and share the GroupNorm tail's ticket)
    if (p.osc != 0 && p.osc != 1 && (p.osc != 2 || p.res || (p.ystats && in_dt != RS_F16S) || p.splitk > 1 || nz != 1 || p.up != 1 || p.stride != 1 || p.C1 != 0)) return -2;
    // Winograd F(2x2,3x3) kernel (split storage, big planes; carries the halo kernel's input transform / statistics / tail): wino.hip
    if (rs_wino_plan(&p, in_dt, out_dt, nz)) return rs_wino_launch(&p, st);
    {   // halo-tile kernel (3x3 stride-1 convs, optional fused GroupNorm affine + SiLU on the input): igemm4.hip
        int tw4 = 0, bc4 = 0;
        if (rs_igemm4_pick(&p, in_dt, out_dt, nz, &tw4, &bc4)) return rs_igemm4_launch(&p, in_dt, tw4, bc4, st);
        if (p.xcoef || p.sx) return -2;   // only the halo kernel applies an input transform / carries a folded shortcut: the caller must ask rs_igemm4_pick first
    }
    if (in_dt == RS_F16S) return rs_igemm_split_launch(&p, out_dt, nz, st);   // split storage: igemm_split.hip (single source)
    // second-generation kernel (LDS-DMA ring) for everything that fills the chip; RS_IGEMM_V2=0 forces the first one
    static const bool use_v2 = []() { const char* e = getenv("RS_IGEMM_V2"); return !(e && e[0] == '0'); }();
    int bp2 = 0, bc2 = 0;
    // third generation (256-pixel tiles, one workgroup per CU, register-pipelined fragments) for the long-K fp16 layers that fill the chip
    int bc3 = 0;
    if (use_v2 && p.C1 == 0 && (out_dt == RS_F16 || out_dt == RS_F32) && rs_igemm3_pick(p.M, p.Cout, p.Ktot, in_dt, nz, p.splitk, &bc3)) return rs_igemm3_launch(&p, out_dt, bc3, st);
    if (use_v2 && p.C1 == 0 && rs_igemm2_pick(p.M, p.Cout, p.Ktot * (in_dt == RS_F16 ? 2 : 4), nz * p.splitk, &bp2, &bc2)) return rs_igemm2_launch(&p, in_dt, out_dt, bp2, bc2, nz, st);
    hipError_t e;
    if (in_dt == RS_F16 && out_dt == RS_F16) e = launch_t<f16, f16>(p, nz, st);
    else if (in_dt == RS_F16 && out_dt == RS_F32) e = launch_t<f16, float>(p, nz, st);
    else if (in_dt == RS_F32 && out_dt == RS_F32) e = launch_t<float, float>(p, nz, st);
    else return -2;
    return e == hipSuccess ? 0 : -1;
}
