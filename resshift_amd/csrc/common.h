// Shared definitions for the ResShift gfx950 kernels.
//
// Activation layout everywhere inside the engine: NHWC ("pixel-major"): tensor
// [B, H, W, C] with C contiguous.  A "pixel stride" (ld) may exceed C so that a
// kernel can read or write a channel slice of a wider tensor (free concat).
//
// Three storage types:
//   RS_F16:  _Float16 storage, fp32 accumulation (v_mfma_f32_16x16x32_f16)
//   RS_F32:  float storage, exact fp32 MFMA (v_mfma_f32_16x16x4_f32)
//   RS_F16S: "split" storage - every value x is kept as TWO fp16 numbers, hi = fp16(x) and lo = fp16((x - hi) * 2^11), so
//            that x = hi + lo * 2^-11 to a relative error <= 2^-23 (fp32 has 2^-24).  A product of two such numbers is
//            hi*hi + 2^-11 (hi*lo + lo*hi) up to 2^-24: three fp16 MFMAs with fp32 accumulation (main + cross
//            accumulator) give fp32-class GEMMs at 1/3 of the fp16 matrix rate instead of the 1/16 of the f32 MFMA
//            (igemm_split.hip).  NHWC pixel record of a tensor with pixel stride ld: [ld halfs hi | ld halfs lo], i.e.
//            4*ld bytes like fp32 storage; a channel slice at c0 starts c0 halfs into the record and keeps `ld`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum RsDType { RS_F16 = 0, RS_F32 = 1, RS_F16S = 2 };
enum RsAct { RS_ACT_NONE = 0, RS_ACT_GELU = 1, RS_ACT_SILU = 2 };

// bytes of one element of a tensor (allocation size); the channel offset of a slice is rs_dtype_chan_bytes
static inline size_t rs_dtype_size(int dt) { return dt == RS_F16 ? 2 : 4; }
static inline size_t rs_dtype_chan_bytes(int dt) { return dt == RS_F32 ? 4 : 2; }

// pointer type of split storage: pointer arithmetic counts halfs, the lo half of element i of a pixel record sits `ld`
// halfs after its hi half
struct h2s { unsigned short bits; };
#define RS_LO_SCALE 2048.0f
#define RS_LO_INV 4.8828125e-4f
__host__ __device__ inline void rs_split(float x, f16& hi, f16& lo) {
    hi = (f16)x;
    lo = (f16)((x - (float)hi) * RS_LO_SCALE);   // x - hi is exact in fp32; the scaling keeps lo a NORMAL fp16 number
}
__device__ __forceinline__ float rs_join(f16 hi, f16 lo) { return fmaf((float)lo, RS_LO_INV, (float)hi); }
// per storage type: pixel-record length in units of `ld` elements, and whether the cheap v_rcp / v_exp activations may
// be used (only where the result is rounded to fp16 anyway)
template <typename T> struct Store { static constexpr int PM = 1; static constexpr bool FAST = false; };
template <> struct Store<f16> { static constexpr int PM = 1; static constexpr bool FAST = true; };
template <> struct Store<h2s> { static constexpr int PM = 2; static constexpr bool FAST = false; };
// scalar element access through a pointer to the hi part (`lo` = distance to the lo part in halfs, ignored otherwise)
template <typename T> __device__ __forceinline__ float rs_ld(const T* p, int lo) { return (float)*p; }
template <> __device__ __forceinline__ float rs_ld<h2s>(const h2s* p, int lo) { return rs_join(((const f16*)p)[0], ((const f16*)p)[lo]); }
template <typename T> __device__ __forceinline__ void rs_st(T* p, int lo, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void rs_st<h2s>(h2s* p, int lo, float v) { f16 h, l; rs_split(v, h, l); ((f16*)p)[0] = h; ((f16*)p)[lo] = l; }

// per-device "function attribute already set" flags of the launchers (hipFuncSetAttribute is per device: a process that drives
// several GPUs must set the dynamic-LDS limit on each of them)
#define RS_MAX_DEVICES 64
// (a device index outside the table, or a failing hipGetDevice, gets the spare slot RS_MAX_DEVICES, whose flag the launchers never keep
// set: such a device pays the hipFuncSetAttribute call on every launch instead of silently inheriting device 0's flag)
static inline int rs_device_slot() { int d = -1; if (hipGetDevice(&d) != hipSuccess) d = -1; return d >= 0 && d < RS_MAX_DEVICES ? d : RS_MAX_DEVICES; }
struct RsAttrFlags {   // "hipFuncSetAttribute done on this device?" of one kernel instantiation; true = the caller must set it now
    bool done[RS_MAX_DEVICES] = {};
    bool need() { const int s = rs_device_slot(); if (s >= RS_MAX_DEVICES) return true; if (done[s]) return false; done[s] = true; return true; }
};

// Between a wave's writes to ITS OWN epilogue staging tile and its reads of that tile (and before the tile is overwritten) the wave's own
// LDS instruction order is all that is needed: a workgroup barrier there makes every wave wait for the slowest one four times per
// split-storage epilogue.  -DRS_EPI_FULL_BARRIER restores the barriers (A/B builds).
#ifdef RS_EPI_FULL_BARRIER
#define RS_STAGING_SYNC() __syncthreads()
#else
#define RS_STAGING_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// Sum over the 16 lanes of a row (lanes 16 r .. 16 r + 15 of the wave), result in every lane: four v_add_f32 with a DPP operand (quad
// permutes, then row_half_mirror and row_mirror - once the four lanes of a quad agree, mirroring pairs the same partners as the xor-4 /
// xor-8 steps of a butterfly, so the operands of every addition - and the bits of the result - are those of the __shfl_xor(1, 2, 4, 8)
// butterfly this replaces).  __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip per step - 640 of them per wave in the halo
// kernel's statistics epilogue.
template <int CTRL> __device__ __forceinline__ float rs_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float rs_sum16(float v) {
    v += rs_dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
    v += rs_dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
    v += rs_dpp_f32<0x141>(v);   // row_half_mirror      (the other quad of the 8)
    v += rs_dpp_f32<0x140>(v);   // row_mirror           (the other 8 of the 16)
    return v;
}

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float rs_silu(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (nn.GELU() default; reference models/swin_transformer.py:18)
__device__ __forceinline__ float rs_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float rs_apply_act(float x, int act) {
    if (act == RS_ACT_GELU) return rs_gelu(x);
    if (act == RS_ACT_SILU) return rs_silu(x);
    return x;
}
// Variants for fp16-STORAGE outputs: v_rcp_f32 / v_exp_f32 (1 ulp each) instead of the IEEE division sequence (12 VALU
// instructions) and libm erff; their error is three orders of magnitude below the fp16 rounding of the stored value.
// The fp32 ("exact") kernels never use them.
__device__ __forceinline__ float rs_silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
// GELU(x) = x/2 * (1 + g(x)), g(x) = erf(x / sqrt 2) ~= x P(x^2) / Q(x^2) with cubic P and Q (odd rational minimax fit on
// |x| <= 4.6, fitted offline against scipy's erf with Lawson reweighting: max abs error 1.7e-6; beyond the clamp 1 - erf
// < 4.2e-6).  The result is stored as fp16 (relative rounding 2.4e-4), so this is two orders below the storage error, and it
// costs one transcendental (v_rcp_f32) plus FMAs the compiler pairs into v_pk_fma_f32 - about half the VALU time of the
// exp + rcp Abramowitz-Stegun form, which matters where the GELU epilogue is VALU-bound (fused Swin MLP, fc1).
__device__ __forceinline__ float rs_gelu_fast(float x) {
    const float xc = fminf(fmaxf(x, -4.6f), 4.6f);
    const float w = xc * xc;
    float pn = fmaf(w, 6.639406754e-05f, 7.644910961e-03f);
    pn = fmaf(w, pn, 5.457502881e-02f);
    pn = fmaf(w, pn, 7.978815839e-01f);
    float qd = fmaf(w, 1.163358082e-03f, 2.373116075e-02f);
    qd = fmaf(w, qd, 2.350743908e-01f);
    qd = fmaf(w, qd, 1.0f);
    const float g = (xc * pn) * __builtin_amdgcn_rcpf(qd);
    const float hx = 0.5f * x;
    return fmaf(hx, g, hx);
}
// fp32-class GELU / SiLU without the IEEE division sequence and libm erff, for the split-precision kernels: v_rcp_f32 and
// v_exp_f32 are 1 ulp each; erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute), i.e. the result is good to ~3e-7 of
// |x| + 1.5e-7 - the class of the split pair itself (2^-23) - at a quarter of the VALU cost of rs_gelu
__device__ __forceinline__ float rs_silu_acc(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float rs_gelu_acc(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(t, 1.061405429f, -1.453152027f);
    pl = fmaf(t, pl, 1.421413741f);
    pl = fmaf(t, pl, -0.284496736f);
    pl = fmaf(t, pl, 0.254829592f);
    const float e = 1.0f - pl * t * __expf(-z * z);      // erf(|x| / sqrt 2)
    const float hx = 0.5f * x;
    return fmaf(copysignf(e, x), hx, hx);
}
// two values per lane at a time: the full-rate operations become v_pk_{mul,fma,add}_f32 (one issue slot for two fp32 results),
// which is what keeps this epilogue work under the MFMA time of the fused split kernels
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 rs_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rs_f32x2 rs_gelu_acc2(rs_f32x2 x) {
    const rs_f32x2 one = {1.0f, 1.0f};
    const rs_f32x2 z = rs_f32x2{fabsf(x.x), fabsf(x.y)} * 0.70710678118654752440f;
    const rs_f32x2 d = __builtin_elementwise_fma(z, rs_f32x2{0.3275911f, 0.3275911f}, one);
    const rs_f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    rs_f32x2 pl = __builtin_elementwise_fma(t, rs_f32x2{1.061405429f, 1.061405429f}, rs_f32x2{-1.453152027f, -1.453152027f});
    pl = __builtin_elementwise_fma(t, pl, rs_f32x2{1.421413741f, 1.421413741f});
    pl = __builtin_elementwise_fma(t, pl, rs_f32x2{-0.284496736f, -0.284496736f});
    pl = __builtin_elementwise_fma(t, pl, rs_f32x2{0.254829592f, 0.254829592f});
    const rs_f32x2 q = pl * t;
    const rs_f32x2 a = (x * x) * (-0.5f * 1.44269504088896340736f);       // -z^2 log2(e)
    const rs_f32x2 ex = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const rs_f32x2 e = __builtin_elementwise_fma(-q, ex, one);               // erf(|x| / sqrt 2)
    const rs_f32x2 hx = x * 0.5f;
    const rs_f32x2 es = {copysignf(e.x, x.x), copysignf(e.y, x.y)};
    return __builtin_elementwise_fma(es, hx, hx);
}
__device__ __forceinline__ void rs_split2(rs_f32x2 v, rs_f16x2& h, rs_f16x2& l) {
    h = __builtin_convertvector(v, rs_f16x2);
    l = __builtin_convertvector((v - __builtin_convertvector(h, rs_f32x2)) * RS_LO_SCALE, rs_f16x2);
}
// ACT is a compile-time constant here so that the element loops carry no per-value branches
template <int ACT, bool FAST> __device__ __forceinline__ float rs_act_t(float x) {
    if constexpr (ACT == RS_ACT_GELU) return FAST ? rs_gelu_fast(x) : rs_gelu(x);
    else if constexpr (ACT == RS_ACT_SILU) return FAST ? rs_silu_fast(x) : rs_silu(x);
    else return x;
}

template <typename T> struct Vec8;  // 8 consecutive elements of T (`lo`: see h2s; ignored by the plain types)
template <> struct Vec8<f16> {
    f16x8 v;
    __device__ __forceinline__ void load(const f16* p, int lo = 0) { v = *(const f16x8*)p; }
    __device__ __forceinline__ void store(f16* p, int lo = 0) const { *(f16x8*)p = v; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};
template <> struct Vec8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p, int lo = 0) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ __forceinline__ void store(float* p, int lo = 0) const { *(f32x4*)p = a; *(f32x4*)(p + 4) = b; }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
    __device__ __forceinline__ void set(int i, float x) { if (i < 4) a[i] = x; else b[i - 4] = x; }
};
template <> struct Vec8<h2s> {
    f16x8 h, l;
    __device__ __forceinline__ void load(const h2s* p, int lo) { h = *(const f16x8*)p; l = *(const f16x8*)((const f16*)p + lo); }
    __device__ __forceinline__ void store(h2s* p, int lo) const { *(f16x8*)p = h; *(f16x8*)((f16*)p + lo) = l; }
    __device__ __forceinline__ float get(int i) const { return rs_join(h[i], l[i]); }
    __device__ __forceinline__ void set(int i, float x) { f16 a, b; rs_split(x, a, b); h[i] = a; l[i] = b; }
};

// ---- launch parameter blocks (plain C structs, passed by value) ------------

// One GroupNorm whose coefficients are produced by a tail.  `coef == nullptr`: no tail on this launch.
struct GNTail {
    const float* gamma; const float* beta;
    const float* film;        // optional [2 * C] FiLM row (scale then shift) of this timestep, or null
    float* coef;              // out: [B][2][C] (scale row, shift row) - GNParams::coef / IGemmParams::xcoef / WinAttnParams::xcoef
    unsigned* ticket;         // [B] arrival counters of THIS tail, zero when the launch starts
    int expected;             // contributing workgroups per image
    int C, groups, HW;        // channels of the normalised tensor (both segments together), groups (32), pixels per image
    float eps;
    // per-channel partial sums [B][S][ld][2] (sum, sum of squares): segment 0 covers channels [0, n0) of the normalised tensor (columns
    // 0 .. n0 - 1 of st0's rows), segment 1 - the other half of a channel concatenation (models/unet.py:891), complete before this launch
    // starts - channels [n0, C) (columns 0 .. C - n0 - 1 of st1's rows).  n0 == C: one segment.
    const float* st0; int S0, ld0, n0;
    const float* st1; int S1, ld1;
    // alternatively per-GROUP partial sums [B][Sg][groups][2] of a statistics pass over the whole tensor (gn_stats_kernel): stg != null
    const float* stg; int Sg;
};


// Implicit-GEMM convolution / GEMM (igemm.hip).  y[m][n] = sum_k X[m][k] W[n][k]
// with X gathered on the fly from NHWC sources (im2col never materialised).
struct IGemmParams {
    const void* x0;      // source 0, NHWC [B,Hs,Ws,(ld0)]
    const void* x1;      // optional source 1 (channel-concatenated after x0), may be null
    const void* w;       // weights [Cout][KH*KW*(C0+C1)], k = (ky*KW+kx)*Ctot + c
    const float* bias;   // [Cout] fp32 or null
    const void* res;     // residual (output dtype) [M][ldres] or null
    void* y;             // output [M][ldy]
    int C0, C1, ld0, ld1;
    int B, Hs, Ws, up;   // source spatial dims; `up`=2 folds a nearest x2 upsample into addressing
    int Ho, Wo, KH, KW, stride, pad_t, pad_l;
    int Cout, ldy, ldres;
    int M, Ktot;
    int act;
    float out_scale;     // applied to the accumulator before bias
    long long bs_x0, bs_w, bs_y, bs_res;  // blockIdx.z batch strides in elements (batched GEMM mode)
    int splitk;          // >1: grid.z slices K; fp32 partial slabs in `partial`, finished by splitk_reduce_kernel
    float* partial;      // [splitk][M][Cout] fp32 workspace (caller owned)
    unsigned x_bytes, w_bytes;  // extents of the x0 / w buffers for the igemm2 buffer descriptors (filled in by its launcher)
    int sh_howo, sh_wo;         // log2(Ho*Wo), log2(Wo) when both are powers of two, else -1 (filled in by the igemm2 launcher)
    int dbg;                    // timing ablations (RS_IGEMM_DBG): 1 = no operand loads after the prologue, 2 = no ds_read/MFMA, 4 = no barriers
    int no_halo;                // 1: never the halo kernel (split storage: it scales the hi weight fragment by 2^11 - a layer with |w| >= 30 takes igemm_split)
    // fused input transform (halo kernel igemm4.hip only): x is the RAW tensor; act_in(x * xcoef[b][0][c] + xcoef[b][1][c]),
    // rounded to the storage type, is what the convolution sees (GroupNorm affine [B][2][C0] of GNParams::coef + SiLU)
    const float* xcoef;
    int xact;
    // fused output statistics (halo kernel only): per-(image, pixel tile, channel) partial sum / sum of squares of the STORED
    // output, [B][tiles per image][ystats_ld][2] floats, for the GroupNorm that consumes y (gn_apply_kernel, GNParams::cpartial):
    // that GroupNorm then needs no statistics pass over the tensor.  Deterministic (fixed summation order, no atomics).
    float* ystats;
    int ystats_ld;
    // GroupNorm tail (gn_tail.h; halo kernel, split-K reduce-with-statistics kernel, igemm_split): tail.coef != null - the launch that
    // completes the statistics of y also writes the consuming GroupNorm's coefficients; the engine fills the GroupNorm's parameters, the
    // coefficient / ticket pointers and the other segment of a concatenation, the launcher the arrival count and segment 0 (= ystats)
    GNTail tail;
    // folded 1x1 shortcut (halo kernel, split storage, 8-wave big-plane tiles only; models/unet.py:178-183,205-206 skip_connection,
    // ldm/modules/diffusionmodules/model.py:121-127,148-149 nin_shortcut): y = conv3x3(act(gn(x0))) + bias + W_s sx + sbias.  The
    // shortcut is sC more K columns of the SAME accumulator - centre-tap stages behind the nine taps' stages, fed from the RAW block
    // input `sx` [B,Hs,Ws,(sld)] (no GroupNorm on those chunks) and the shortcut's split weights `sw` [Cout][sC hi | sC lo]: its
    // GEMM launch, its output tensor and the residual read of that tensor disappear.  sx == null: none.
    const void* sx; const void* sw; const float* sbias;
    int sC, sld;
    unsigned sx_bytes, sw_bytes;   // (filled in by the launcher)
    // output scatter of the sub-pixel form of "nearest x2 upsample + conv3x3" (engine.hip upfold; models/unet.py:53-81,
    // ldm/modules/diffusionmodules/model.py:50-65): osc == 2 - this launch is one of four 2x2 convs over the LOW-resolution grid Ho x Wo and
    // GEMM row (b, oy, ox) is pixel (2 oy + ooy, 2 ox + oox) of the [B][2 Ho][2 Wo] output tensor.  Generic kernels only (igemm / igemm2 /
    // igemm3 / igemm_split), no residual, no split-K; output statistics from igemm_split only (the four launches fill ONE slab array -
    // an image's slabs are class 0's tiles, then class 1's, ... - and share the GroupNorm tail's ticket).  osc == 0 / 1: none.
    int osc, ooy, oox;
    // Winograd F(2x2,3x3) form of this layer's weights (wino.hip: rs_wino_pack order; split storage only), or null: with it a 3x3 / stride-1
    // conv on a plane that tiles by 16 x 16 runs on wino_kernel instead of the halo kernel (rs_wino_plan)
    const void* ww;
};

#if defined(__HIPCC__)
__device__ __forceinline__ long long rs_out_m(const IGemmParams& p, long long m) {
    if (p.osc != 2) return m;
    const int hw = p.Ho * p.Wo;
    const int b = (int)(m / hw), r = (int)(m - (long long)b * hw);
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    return ((long long)b * (2 * p.Ho) + 2 * oy + p.ooy) * (2 * p.Wo) + 2 * ox + p.oox;
}
#endif

struct DirectConvParams {
    const void* x0; const void* x1;
    const float* w;      // fp32, layout [KH*KW*Ctot][Cout] (cout fastest)
    const float* bias;
    void* y;
    int C0, C1, ld0, ld1;
    int B, Hs, Ws, up, Ho, Wo, KH, KW, stride, pad_t, pad_l;
    int Cout, ldy, act;
};

struct GNParams {
    const void* x; void* y;
    const float* gamma; const float* beta;
    const float* film;   // optional [2*C] (scale then shift) for this (timestep); null if none
    float* partial;      // [B][S][32][2] partial sums
    int B, HW, C, ldx, ldy, S, groups;
    float eps; int act;
    float* coef;         // non-null: do not normalise; write the per-(image, channel) affine [B][2][C] (scale row, then shift row)
                         // so that a consumer kernel can apply y = x * scale + shift while it loads x (fused Swin kernels)
    const float* cpartial;   // non-null: per-channel partial sums [B][S][cp_ld][2] written by the producing conv's epilogue
    int cp_ld;               // (IGemmParams::ystats) replace `partial`; no statistics kernel runs
    const float* cpartial2;  // non-null: the tensor is a channel concatenation (models/unet.py:891) whose halves come from two producers -
    int cp2_ld, cp2_S, cp_n0;   // `cpartial` (S sets) covers channels [0, cp_n0), `cpartial2` ([B][cp2_S][cp2_ld][2]) channels [cp_n0, C)
    unsigned* ticket;        // non-null (with `coef`, without `cpartial`): [B] zeroed arrival counters - the statistics kernel's last
                             // workgroup per image writes the coefficients itself (gn_tail.h), no second launch
};

struct WinAttnParams {
    const void* qkv;      // [B,H,W,ldq]: feature f = which*E + head*32 + d (swin_transformer.py:121)
    void* out;            // [B,H,W,ldo]: feature head*32 + d
    const float* bias_t;  // [heads][64 (key j)][64 (query i)] relative position bias, transposed (VALU kernel)
    const float* bias_n;  // [heads][64 (query i)][64 (key j)] (MFMA kernel); may be null -> VALU kernel is used
    const float* bias_c;  // [heads][256]: the 225 distinct values per head (index (dy + 7) * 15 + dx + 7, swin_transformer.py:93-102) x log2 e, zero
                          // padded - the split fused kernel copies it to LDS with one LDS-DMA instruction per wave (win_attn_split.hip)
    int B, H, W, heads, shift, ldq, ldo;
    float scale;
    // fused qkv projection (win_attn_qkv_kernel): normalised tokens instead of a qkv tensor
    const void* x;        // [B,H,W,ldx] fp16, features 0..E-1
    const void* wqkv;     // [3E][E] weight of the qkv Linear (swin_transformer.py:85) in FRAGMENT-MAJOR order (engine.hip ConvW::wh_frag / ws_frag)
    const float* bqkv;    // [3E]
    int ldx;
    // optional fused output projection + residual (swin_transformer.py:141-143,277): out = res + proj(attention)
    const void* wproj;    // [E][E] in fragment-major order, null -> `out` receives the attention result itself
    const float* bproj;   // [E]
    const void* res;      // [B,H,W,ldres] fp16 shortcut, read at the same (un-shifted) pixels the result is written to
    int ldres;
    const float* xcoef;   // optional GroupNorm affine [B][2][E] (GNParams::coef): x is the raw tensor, normalised on the fly
    // optional statistics of the stored output for the GroupNorm that consumes it (norm2, swin_transformer.py:279; GNParams::cpartial):
    // [B][windows per image][ystats_ld][2] floats (sum, sum of squares over the window's 64 tokens), written by the fused kernels'
    // projection epilogue (each wave owns 32 output features of ALL tokens of the window: a wave-local reduction, no atomics)
    float* ystats;
    int ystats_ld;
    // GroupNorm tail (gn_tail.h; the split-storage fused kernel): the wave that publishes the image's last statistics also writes norm2's
    // coefficients - every wave (head) of every window arrives once
    GNTail tail;
};
