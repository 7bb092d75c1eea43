// Fused Swin MLP for fp16 storage (models/swin_transformer.py:17-33 Mlp + the residual of :279):
//     y = res + fc2( GELU( fc1(x) ) ),   x = GroupNorm'd tokens [M][E], fc1: E -> HD, fc2: HD -> E  (E = 192, HD = 768)
// The two 1x1-conv GEMMs of this block are HBM-bound when run separately (profiles/r1_igemm_ablation.txt: the fc1 launch
// spends 70 % of its time writing the [M][HD] hidden tensor, fc2 starts by reading it back: 2 x 201 MB at batch 32 on
// the 64x64 level).  Here one workgroup owns 128 tokens: the token tile stays in LDS, the hidden activations exist only
// as 64-wide chunks (MFMA accumulators -> bias + GELU -> fp16 -> LDS), and the fc2 accumulators stay in registers:
//   for hc in 0 .. HD/64:   S  = X W1[hc]^T         (128 x 64, K = E)      24 MFMAs / wave
//                           P  = fp16(GELU(S + b1))  -> LDS (16 KB)
//                           O += P W2[:, hc]^T       (128 x E, K = 64)      24 MFMAs / wave
// LDS (160 KB): W1 chunk ring 3 x 24 KB | W2 chunk ring 3 x 24 KB | P 16 KB; weights arrive by LDS-DMA two chunks ahead
// (they are L2-resident: every workgroup streams the same 590 KB); the token fragments stay in registers.  Same LDS image / swizzle / MFMA operand
// roles as igemm2.hip: weights are the A operand, tokens the B operand, so a lane ends with 4 consecutive output
// channels of one token.
// (A variant with wave-private token rows that keeps the hidden chunk in registers - GEMM1's accumulators reused as GEMM2's
// B operand through a consistent k-permutation, no P round trip, one barrier per chunk - was measured at 180 us against
// 146 us for this kernel at 131072 tokens: every wave then reads the whole W1 and W2 chunk from LDS, 72 LDS instructions
// per chunk per wave.  profiles/r1_igemm_ablation.txt.)
#include "igemm_common.h"
#include "gn_tail.h"
#include <type_traits>
#include <cstdlib>

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

// per-channel sum / sum of squares of a wave tile (32 tokens x 16 FC2 channels, lane (lr, lg) holds 4 channels of 2 tokens per channel
// fragment) -> ystats[image][slab of 32 tokens][channel][2]; every token of the tile belongs to one image (HW % 32 == 0)
template <int FC2>
__device__ __forceinline__ void mlp_stats(const f32x4 (&o)[FC2][2], float* ystats, int ld, int mw, int HW, int c0, int lr, int lg) {
    const int b = mw / HW, slab = (mw - b * HW) >> 5;
    float* dst = ystats + (((long long)b * (HW >> 5) + slab) * ld + c0) * 2;
#pragma unroll
    for (int i = 0; i < FC2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = rs_sum16(o[i][0][r] + o[i][1][r]), q = rs_sum16(fmaf(o[i][0][r], o[i][0][r], o[i][1][r] * o[i][1][r]));   // (DPP adds)
            if (lr == 0) { dst[(i * 16 + lg * 4 + r) * 2] = a; dst[(i * 16 + lg * 4 + r) * 2 + 1] = q; }
        }
}

#ifdef RS_SPLIT_ABLATE
__device__ long long g_mlp_clk[8 * 4096];   // phase stamps of wave 0 of the first 4096 workgroups (fp16 and split kernels share it)
#define RS_MLP_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0 && blockIdx.x < 4096) g_mlp_clk[8 * blockIdx.x + (k)] = clock64(); } while (0)
#else
#define RS_MLP_STAMP(k)
#endif

struct MlpParams {
    const f16* x; const f16* w1; const float* b1; const f16* w2; const float* b2; const f16* res; f16* y;
    int M, ldx, ldres, ldy;
    const float* xcoef;   // optional GroupNorm affine [B][2][E] (GNParams::coef): x is the raw tensor, normalised while it is loaded
    int HW;               // tokens per image (a multiple of the 128-token tile whenever xcoef is set)
    // optional statistics of the stored output for the GroupNorm that consumes it (the next block's norm1; GNParams::cpartial):
    // [B][HW / 32][ystats_ld][2] floats = sum / sum of squares over the 32 tokens of one wave tile (wave-local, no atomics); needs
    // HW % 32 == 0 and M % 32 == 0
    float* ystats;
    int ystats_ld;
};

template <int E, int HD>
__global__ __launch_bounds__(512, 2) void swin_mlp_kernel(MlpParams p) {
    constexpr int BP = 128, HC = 64, NHC = HD / HC, KS1 = E / 64, NSL = 3;
    // LDS: W1 chunk ring | W2 chunk ring (NSL slots each, chunks travel two iterations ahead of their use) | P.  The token tile
    // is NOT staged: its MFMA fragments are loop-invariant over the hidden chunks and live in registers (48 VGPRs).
    constexpr int W1R = 0, W1_STAGE = HC * 128, W1_SLOT = KS1 * W1_STAGE;
    constexpr int W2R = W1R + NSL * W1_SLOT, W2_SLOT = E * 128;
    constexpr int PS = W2R + NSL * W2_SLOT;
    constexpr int FC2 = E / 32;                    // channel fragments per wave in GEMM2 (wave covers E/2 channels)
    constexpr int LW = KS1 + E / 64;               // LDS-DMA instructions per thread per weight chunk
    static_assert(E % 64 == 0 && HD % HC == 0 && NHC >= 3 && PS + BP * 128 <= 160 * 1024, "shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 3, wc = wave >> 2;             // 4 token-waves x 2 channel-waves
    const int rr = 8 * wave + (lane >> 3);               // row inside a 64-row load round
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);      // source K-chunk (swizzle on the source side)
    const int m0 = blockIdx.x * BP;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (unsigned)(HD * E * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)(HD * E * 2), 0x00020000);

    // one hidden chunk of both weight matrices -> ring slot
    auto issue_w = [&](int hc, int slot) {
        char* b1 = smem + W1R + slot * W1_SLOT + (8 * wave) * 128;
        const unsigned n = (unsigned)(hc * HC + rr);
#pragma unroll
        for (int s = 0; s < KS1; ++s) lds_dma16(r1, b1 + s * W1_STAGE, (n * E + (unsigned)(s * 64 + kcp * 8)) * 2u);
        char* b2 = smem + W2R + slot * W2_SLOT + (8 * wave) * 128;
#pragma unroll
        for (int i = 0; i < E / 64; ++i) lds_dma16(r2, b2 + (64 * i) * 128, ((unsigned)(64 * i + rr) * HD + (unsigned)(hc * HC + kcp * 8)) * 2u);
    };
    RS_MLP_STAMP(0);
    issue_w(0, 0);

    // token fragments (MFMA B operand: lane (lr, lg) holds 8 consecutive K values of token row lr): straight from global memory
    f16x8 xf[2 * KS1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = min(m0 + wp * 32 + j * 16 + lr, p.M - 1);   // rows beyond M are clamped: their results are never stored
        const f16* xr = p.x + (long long)m * p.ldx + lg * 8;
#pragma unroll
        for (int ks = 0; ks < 2 * KS1; ++ks) xf[ks][j] = *(const f16x8*)(xr + ks * 32);
    }
    // the shortcut of this wave's output tile (lane: 4 channels of 2 tokens per channel fragment), requested with the tokens and consumed
    // in the epilogue: fetched there it was an exposed burst of twelve half-used-line loads per wave (8 k of a workgroup's 70 k cycles,
    // profiles/r3_attn_phases.txt); 24 registers for the length of the kernel
    const bool res_ok = p.res != nullptr;
    f16x4 rv[FC2][2];
    if (res_ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long mr = (long long)min(m0 + wp * 32 + j * 16 + lr, p.M - 1) * p.ldres;
#pragma unroll
            for (int i = 0; i < FC2; ++i) rv[i][j] = *(const f16x4*)(p.res + mr + wc * (E / 2) + i * 16 + lg * 4);
        }
    }
    if (p.xcoef) {   // GroupNorm (norm2) folded in, rounded to fp16 exactly where the separate apply kernel rounds
        const float* sc = p.xcoef + (long long)(m0 / p.HW) * 2 * E;
#pragma unroll
        for (int ks = 0; ks < 2 * KS1; ++ks) {
            const int c0 = ks * 32 + lg * 8;
            const f32x4 a0 = *(const f32x4*)(sc + c0), a1 = *(const f32x4*)(sc + c0 + 4);
            const f32x4 d0 = *(const f32x4*)(sc + E + c0), d1 = *(const f32x4*)(sc + E + c0 + 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xf[ks][j][e] = (f16)fmaf((float)xf[ks][j][e], a0[e], d0[e]);
                    xf[ks][j][4 + e] = (f16)fmaf((float)xf[ks][j][4 + e], a1[e], d1[e]);
                }
        }
    }
    // fc1 bias of the chunk being processed, fetched one chunk ahead
    f32x4 bcur[2], bnxt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { bcur[i] = *(const f32x4*)(p.b1 + wc * 32 + i * 16 + lg * 4); bnxt[i] = bcur[i]; }
    // chunk 1 is requested LAST: the first iteration's counted wait (all but the youngest LW + 2 operations) then covers chunk 0, the
    // tokens and the shortcut but not chunk 1 - the order every later iteration has (bias pair, then the chunk two ahead)
    __builtin_amdgcn_sched_barrier(0);   // (pins "bias pair, then chunk 1" as the youngest LW + 2 operations the counted wait below leaves in flight)
    issue_w(1, 1);

    const int swz[2] = {(lg ^ (lr & 7)) << 4, ((4 + lg) ^ (lr & 7)) << 4};   // k-step 0 / 1 inside a 128-byte stage
    f32x4 o[FC2][2];
#pragma unroll
    for (int i = 0; i < FC2; ++i) { o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    int slot = 0;
    RS_MLP_STAMP(1);
    for (int hc = 0; hc < NHC; ++hc) {
        // chunk hc has landed once only the younger loads may be outstanding: the bias pair and the LW DMA instructions of
        // chunk hc+1 (issued one iteration ago, in that order); the tail iterations simply drain everything
        if (hc + 1 < NHC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LW + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... everybody's part too; GEMM2 of chunk hc-1 is finished everywhere
        if (hc + 2 < NHC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) bnxt[i] = *(const f32x4*)(p.b1 + (hc + 1) * HC + wc * 32 + i * 16 + lg * 4);
            issue_w(hc + 2, slot >= 1 ? slot - 1 : NSL - 1);   // == (slot + 2) % NSL: the slot chunk hc-1 just released
        } else if (hc + 1 < NHC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) bnxt[i] = *(const f32x4*)(p.b1 + (hc + 1) * HC + wc * 32 + i * 16 + lg * 4);
        }

        // ---- GEMM1: S[h][m] over K = E, wave tile 32 hidden x 32 tokens
        f32x4 s_[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { s_[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s_[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2 * KS1; ++ks) {
            const char* pa = smem + W1R + slot * W1_SLOT + (ks >> 1) * W1_STAGE + (wc * 32 + lr) * 128 + swz[ks & 1];
            f16x8 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *(const f16x8*)(pa + i * 2048);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) s_[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], xf[ks][j], s_[i][j], 0, 0, 0);
        }
        // ---- bias + GELU -> fp16 hidden chunk P[m][h] in LDS (a lane holds hidden h..h+3 of token m)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int h = wc * 32 + i * 16 + lg * 4;                                  // hidden index inside the chunk
            const f32x4 bv = bcur[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = wp * 32 + j * 16 + lr;
                const f32x4 v = s_[i][j] + bv;
                f16x4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (f16)rs_gelu_fast(v[r]);
                *(f16x4*)(smem + PS + m * 128 + (((h >> 3) ^ (m & 7)) << 4) + (h & 7) * 2) = hv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's P stores are in LDS (a raw barrier carries no wait, and
        __builtin_amdgcn_s_barrier();                        // __syncthreads() would also drain the weight prefetch in flight)
        // ---- GEMM2: O[c][m] += W2[c][hc*64 ..] . P[m][..], wave tile (E/2) channels x 32 tokens, K = 64
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* pa = smem + W2R + slot * W2_SLOT + (wc * (E / 2) + lr) * 128 + swz[ks];
            const char* pb = smem + PS + (wp * 32 + lr) * 128 + swz[ks];
            f16x8 a[FC2], b[2];
#pragma unroll
            for (int i = 0; i < FC2; ++i) a[i] = *(const f16x8*)(pa + i * 2048);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const f16x8*)(pb + j * 2048);
#pragma unroll
            for (int i = 0; i < FC2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], o[i][j], 0, 0, 0);
        }
        bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
        slot = slot + 1 == NSL ? 0 : slot + 1;
#ifdef RS_SPLIT_ABLATE
        if (hc == 0) RS_MLP_STAMP(2);
#endif
    }
    RS_MLP_STAMP(3);
    __syncthreads();   // every read of the rings / P has retired: the front of the LDS becomes the output staging area
    RS_MLP_STAMP(4);

    // ---- epilogue: + b2 + residual -> fp16 -> transposition through LDS -> 16-byte NHWC stores
    constexpr int ROWB = (E / 2) * 2 + 16;
    char* stg = smem + wave * 32 * ROWB;
#pragma unroll
    for (int i = 0; i < FC2; ++i) {
        const int n = wc * (E / 2) + i * 16 + lg * 4;
        const f32x4 bv = *(const f32x4*)(p.b2 + n);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 v = o[i][j] + bv;
            if (res_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rv[i][j][r];
            }
            f16x4 hv;
            hv[0] = (f16)v[0]; hv[1] = (f16)v[1]; hv[2] = (f16)v[2]; hv[3] = (f16)v[3];
            *(f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][j][r] = (float)hv[r];   // the stored value, for the statistics below
        }
    }
    if (p.ystats && m0 + wp * 32 < p.M) mlp_stats<FC2>(o, p.ystats, p.ystats_ld, m0 + wp * 32, p.HW, wc * (E / 2), lr, lg);
    RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
    RS_MLP_STAMP(5);
    constexpr int CPR = (E / 2) / 8, NITEM = 32 * CPR;
    for (int idx = lane; idx < NITEM; idx += 64) {
        const int row = idx / CPR, c8 = idx - row * CPR;
        const int m = m0 + wp * 32 + row;
        if (m >= p.M) continue;
        *(uint4*)(p.y + (long long)m * p.ldy + wc * (E / 2) + c8 * 8) = *(const uint4*)(stg + row * ROWB + c8 * 16);
    }
    RS_MLP_STAMP(6);
}

}  // namespace

// Launch conditions (checked by the caller as well): fp16 storage, E = 192, HD = 768, 16-byte aligned rows.
extern "C" int rs_swin_mlp_supported(int E, int HD) { return E == 192 && HD == 768; }

extern "C" int rs_swin_mlp_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y,
                                  int M, int ldx, int ldres, int ldy, int E, int HD, const float* xcoef, int HW, float* ystats, int ystats_ld,
                                  hipStream_t st) {
    if (ystats && (HW <= 0 || (HW & 31) || (M & 31))) return -2;
    if (!rs_swin_mlp_supported(E, HD) || (ldx & 7) || (ldy & 7) || (res && (ldres & 3)) || M <= 0) return -2;
    if (xcoef && (HW <= 0 || HW % 128)) return -2;
    MlpParams p{};
    p.x = (const f16*)x; p.w1 = (const f16*)w1; p.b1 = b1; p.w2 = (const f16*)w2; p.b2 = b2; p.res = (const f16*)res; p.y = (f16*)y;
    p.M = M; p.ldx = ldx; p.ldres = ldres; p.ldy = ldy; p.xcoef = xcoef; p.HW = HW; p.ystats = ystats; p.ystats_ld = ystats_ld;
    constexpr int LDS = 160 * 1024;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)swin_mlp_kernel<192, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    hipLaunchKernelGGL((swin_mlp_kernel<192, 768>), dim3((M + 127) / 128), dim3(512), LDS, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same fusion for SPLIT storage (RS_F16S, common.h): y = res + fc2(GELU(fc1(x))) with every product formed from (hi, lo)
// fp16 pairs - three MFMAs into ONE accumulator, the hi weight fragment scaled by 2^11 (igemm4.hip) - so that the parity-
// qualified policy does not write and re-read the [M][768] hidden tensor (2 x 403 MB per Swin block at batch 32 on the 64 x 64
// level) through two short-K split GEMMs.  One workgroup owns 128 tokens; a step handles 32 hidden units:
//     S = X W1[h..h+32]^T  (K = E: six 32-channel chunks, token fragments (hi, lo) in registers)      36 MFMAs / wave
//     P = split(GELU(S 2^-11 + b1)) -> LDS rows [32 hid hi | 32 hid lo] per token                      (16 KB)
//     O += P W2[:, h..h+32]^T (K = 32)                                                                 36 MFMAs / wave
// LDS rows are 128 bytes = [32 hi | 32 lo] halfs with the (position ^ row & 7) swizzle everywhere: W1 step [6 chunks][32 rows],
// W2 step [192 rows], two slots each (96 KB) + P.  Waves: 4 token-waves x 2 (hidden / channel)-waves.
// Measured at 131 072 tokens: 337 us (229 TFLOP/s algorithmic = 688 TFLOP/s of MFMA work, 29 % MFMA-busy) against 215 + 109 us
// + the hidden tensor's round trip for the two separate split GEMMs.  Three other schedules of the same tile were measured
// and dropped (scripts/mlp_split_time.py): GEMM1 of step t+1 issued ahead of the activation math of step t (W1 one step ahead of
// W2), one barrier per step with a double-buffered P, and that one cut into six MFMA / VALU blocks by scheduling fences:
// 343 - 363 us, all at the 256-register limit.  With 8 waves per CU (one workgroup: 112 KB of LDS) the kernel is bound by the
// latency chain barrier -> LDS fragment reads -> MFMA -> activation -> LDS, not by any single pipe: timing ablations without
// the weight DMA, without the activation math and without the MFMAs each remove only 12 - 28 %.
namespace {

struct MlpSplitParams {
    const f16* x; const f16* w1; const float* b1; const f16* w2; const float* b2; const f16* res; f16* y;
    int M, ldx, ldres, ldy;
    const float* xcoef;   // optional GroupNorm affine [B][2][E]: x is the raw tensor, normalised (joined value) while it is loaded
    int HW;
    float* ystats;        // optional output statistics: [B][HW / 128][ystats_ld][2] floats, one set per workgroup tile (128 tokens of one image:
    int ystats_ld;        // HW % 128 == 0, M % 128 == 0) - sum / sum of squares per channel, for the GroupNorm that consumes y (the next block's norm1)
    GNTail tail;          // ... and with tail.coef that GroupNorm's coefficients too (gn_tail.h)
};

// NO != E: the layer's patch_unembed (models/swin_transformer.py:515,521-528: a 1x1 conv E -> NO behind the last block, no norm in the shipped
// configs) folded into this launch.  y = Wu (x + fc2(h) + b2) + bu = [Wu W2 | Wu] [h ; x] + (Wu b2 + bu): GEMM2 runs on the product matrix
// (packed by the engine: `w2` = [NO][HD + E] columns, hi | lo; `b2` = the merged bias, NO floats) with NO output rows, and the block's
// shortcut becomes KC more K steps whose token operand is the RAW x (re-read into the token-fragment registers once GEMM1 has used them
// for the last time - the same bytes the residual read costs the unfused form).  No `res`; y has NO channels; the statistics are y's.
template <int E, int HD, int NO = E>
__global__ __launch_bounds__(512, 2) void swin_mlp_split_kernel(MlpSplitParams p) {
    constexpr bool FOLD = NO != E;
    constexpr int BP = 128, HS = 32, NST = HD / HS, KC = E / 32;     // tokens per workgroup, hidden units per step, steps, K chunks
    constexpr int NSTT = FOLD ? NST + KC : NST, HDT = FOLD ? HD + E : HD;   // steps / K columns of GEMM2 incl. the folded shortcut
    constexpr int W1_SLOT = KC * HS * 128, W2_SLOT = NO * 128;       // 24 KB / 20 - 24 KB
    constexpr int W1R = 0, W2R = 2 * W1_SLOT, PS = W2R + 2 * W2_SLOT;
    constexpr int FC2 = NO / 32;                                     // channel fragments per wave in GEMM2 (wave covers NO/2 channels)
    static_assert(E % 64 == 0 && NO % 32 == 0 && NO <= E && HD % HS == 0 && PS + BP * 128 <= 160 * 1024, "shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 3, wc = wave >> 2;
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);      // logical 16-byte chunk this lane fetches (source-side swizzle)
    const int m0 = blockIdx.x * BP;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (unsigned)(HD * E * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)(HDT * NO * 4), 0x00020000);
    // weights of one step -> ring slot.  W1 (rows [E hi | E lo]): 6 chunks x 32 hidden rows = 24 one-KB pieces, 3 per wave;
    // W2 (rows [HD hi | HD lo]): 192 channel rows = 24 pieces, 3 per wave
    auto issue_w = [&](int t, int slot) {
        const unsigned plane = (unsigned)(kcp >> 2), sub = (unsigned)(kcp & 3) * 8u;
        if (!FOLD || t < NST) {   // (workgroup-uniform; the folded shortcut's steps have no W1 part)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int piece = wave * 3 + q;                 // 0..23: chunk = piece / 4, row group = piece % 4
                const int c = piece >> 2, rg = piece & 3;
                const unsigned n = (unsigned)(t * HS + rg * 8 + (lane >> 3));
                lds_dma16(r1, smem + W1R + slot * W1_SLOT + piece * 1024, (n * (unsigned)(2 * E) + plane * (unsigned)E + (unsigned)(c * 32) + sub) * 2u);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int piece = wave * 3 + q;                 // rows 8 piece ..
            if (NO < 192 && piece * 8 >= NO) continue;      // (wave-uniform; NO = 160: 20 of the 24 pieces)
            const unsigned n = (unsigned)(piece * 8 + (lane >> 3));
            lds_dma16(r2, smem + W2R + slot * W2_SLOT + piece * 1024, (n * (unsigned)(2 * HDT) + plane * (unsigned)HDT + (unsigned)(t * HS) + sub) * 2u);
        }
    };
    RS_MLP_STAMP(0);
    issue_w(0, 0);

    // token fragments (MFMA B operand), hi and lo, straight from global memory: lane (lr, lg) holds channels 32 c + 8 lg .. of token lr
    f16x8 xh[KC][2], xl[KC][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = min(m0 + wp * 32 + j * 16 + lr, p.M - 1);   // rows beyond M are clamped: their results are never stored
        const f16* xr = p.x + (long long)m * p.ldx * 2 + lg * 8;
#pragma unroll
        for (int c = 0; c < KC; ++c) { xh[c][j] = *(const f16x8*)(xr + c * 32); xl[c][j] = *(const f16x8*)(xr + p.ldx + c * 32); }
    }
    if (p.xcoef) {   // GroupNorm (norm2) folded in: joined value * scale + shift, split again
        const float* sc = p.xcoef + (long long)(m0 / p.HW) * 2 * E;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int c0 = c * 32 + lg * 8;
            const f32x4 a0 = *(const f32x4*)(sc + c0), a1 = *(const f32x4*)(sc + c0 + 4);
            const f32x4 d0 = *(const f32x4*)(sc + E + c0), d1 = *(const f32x4*)(sc + E + c0 + 4);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f16 hh, ll;
                    rs_split(fmaf(rs_join(xh[c][j][e], xl[c][j][e]), e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? d0[e & 3] : d1[e & 3]), hh, ll);
                    xh[c][j][e] = hh; xl[c][j][e] = ll;
                }
        }
    }
    const int swh = (lg ^ (lr & 7)) << 4, swl = ((4 + lg) ^ (lr & 7)) << 4;   // hi / lo position of k-group lg in row lr (16-row aligned tiles)
    f32x4 o[FC2][2];
#pragma unroll
    for (int i = 0; i < FC2; ++i) { o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    RS_MLP_STAMP(1);
    for (int t = 0; t < NST; ++t) {
        const int slot = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of step t have landed
        __builtin_amdgcn_s_barrier();                        // ... everybody's; GEMM2 of step t-1 is finished everywhere (P and the other slot are free)
        if (t + 1 < NSTT) issue_w(t + 1, slot ^ 1);
        const f32x4 bv = *(const f32x4*)(p.b1 + t * HS + wc * 16 + lg * 4);
        // ---- GEMM1: S[hidden 16 of this wave][32 tokens] over K = E (accumulator carries 2^11 x the sum)
        f32x4 s_[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const char* pa = smem + W1R + slot * W1_SLOT + c * (HS * 128) + (wc * 16 + lr) * 128;
            const f16x8 ah = *(const f16x8*)(pa + swh), al = *(const f16x8*)(pa + swl);
            const f16x8 as = ah * (f16)RS_LO_SCALE;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s_[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, xh[c][j], s_[j], 0, 0, 0);
                s_[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[c][j], s_[j], 0, 0, 0);
                s_[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[c][j], s_[j], 0, 0, 0);
            }
        }
        if (FOLD && t == NST - 1) {
            // GEMM1 has read the (normalised) token fragments for the last time: the registers take the RAW tokens, the operand of the folded
            // shortcut's K steps; the loads travel under this step's activation math and GEMM2 (the next step's vmcnt(0) covers them)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = min(m0 + wp * 32 + j * 16 + lr, p.M - 1);
                const f16* xr = p.x + (long long)m * p.ldx * 2 + lg * 8;
#pragma unroll
                for (int c = 0; c < KC; ++c) { xh[c][j] = *(const f16x8*)(xr + c * 32); xl[c][j] = *(const f16x8*)(xr + p.ldx + c * 32); }
            }
        }
        // ---- bias + GELU -> (hi, lo) hidden activations P[token][32 hidden] in LDS (a lane holds hidden h..h+3 of token m)
        {
            const int h = wc * 16 + lg * 4;       // hidden index inside the step: hi position h >> 3, lo position 4 + (h >> 3)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = wp * 32 + j * 16 + lr;
                // (two values per lane at a time: v_pk_fma_f32 / v_pk_mul_f32 / v_cvt_pk_f16_f32)
                rs_f16x2 h0, l0, h1, l1;
                rs_split2(rs_gelu_acc2(__builtin_elementwise_fma(rs_f32x2{s_[j][0], s_[j][1]}, rs_f32x2{RS_LO_INV, RS_LO_INV}, rs_f32x2{bv[0], bv[1]})), h0, l0);
                rs_split2(rs_gelu_acc2(__builtin_elementwise_fma(rs_f32x2{s_[j][2], s_[j][3]}, rs_f32x2{RS_LO_INV, RS_LO_INV}, rs_f32x2{bv[2], bv[3]})), h1, l1);
                char* row = smem + PS + m * 128 + (h & 7) * 2;
                *(f16x4*)(row + (((h >> 3) ^ (m & 7)) << 4)) = f16x4{h0.x, h0.y, h1.x, h1.y};
                *(f16x4*)(row + (((4 + (h >> 3)) ^ (m & 7)) << 4)) = f16x4{l0.x, l0.y, l1.x, l1.y};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- GEMM2: O[c][m] += W2[c][h..h+32] . P[m][..], wave tile (E/2) channels x 32 tokens, K = 32
        {
            const char* pa = smem + W2R + slot * W2_SLOT + (wc * (NO / 2) + lr) * 128;
            const char* pb = smem + PS + (wp * 32 + lr) * 128;
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { bh[j] = *(const f16x8*)(pb + j * 2048 + swh); bl[j] = *(const f16x8*)(pb + j * 2048 + swl); }
#pragma unroll
            for (int i = 0; i < FC2; ++i) {
                const f16x8 ah = *(const f16x8*)(pa + i * 2048 + swh), al = *(const f16x8*)(pa + i * 2048 + swl);
                const f16x8 as = ah * (f16)RS_LO_SCALE;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh[j], o[i][j], 0, 0, 0);
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], o[i][j], 0, 0, 0);
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], o[i][j], 0, 0, 0);
                }
            }
        }
#ifdef RS_SPLIT_ABLATE
        if (t == 0) RS_MLP_STAMP(2);
#endif
    }
    if constexpr (FOLD) {
        // ---- the folded shortcut: KC more K steps of GEMM2, columns [Wu] of the product matrix against the raw tokens (registers)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int t = NST + c, slot = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of step t (and, c = 0, its raw tokens) have landed
            __builtin_amdgcn_s_barrier();                        // ... everybody's; the other slot is free
            if (t + 1 < NSTT) issue_w(t + 1, slot ^ 1);
            const char* pa = smem + W2R + slot * W2_SLOT + (wc * (NO / 2) + lr) * 128;
#pragma unroll
            for (int i = 0; i < FC2; ++i) {
                const f16x8 ah = *(const f16x8*)(pa + i * 2048 + swh), al = *(const f16x8*)(pa + i * 2048 + swl);
                const f16x8 as = ah * (f16)RS_LO_SCALE;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, xh[c][j], o[i][j], 0, 0, 0);
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[c][j], o[i][j], 0, 0, 0);
                    o[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[c][j], o[i][j], 0, 0, 0);
                }
            }
        }
    }
    RS_MLP_STAMP(3);
    __syncthreads();   // every read of the rings / P has retired: the front of the LDS becomes the output staging area
    RS_MLP_STAMP(4);

    // ---- epilogue: 2^-11 O + b2 + residual, then the hi and the lo halves through LDS into 16-byte stores
    constexpr int ROWB = (NO / 2) * 2 + 16;
    char* stg = smem + wave * 32 * ROWB;
    // (the residual of the whole wave tile in flight at once - 48 registers, the token fragments are dead by now: loaded per
    // fragment it was twelve L2 round trips in a row)
    const bool res_ok = !FOLD && p.res != nullptr;
    f16x4 rh[FC2][2], rl[FC2][2];
    if (res_ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long mr = (long long)min(m0 + wp * 32 + j * 16 + lr, p.M - 1) * p.ldres * 2;
#pragma unroll
            for (int i = 0; i < FC2; ++i) {
                const int n = wc * (NO / 2) + i * 16 + lg * 4;
                rh[i][j] = *(const f16x4*)(p.res + mr + n); rl[i][j] = *(const f16x4*)(p.res + mr + p.ldres + n);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FC2; ++i) {
        const int n = wc * (NO / 2) + i * 16 + lg * 4;
        const f32x4 bv = *(const f32x4*)(p.b2 + n);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 v = o[i][j] * RS_LO_INV + bv;
            if (res_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rs_join(rh[i][j][r], rl[i][j][r]);
            }
            o[i][j] = v;
        }
    }
    // Output statistics (one set per 128-token tile): the waves' per-channel sums over their 32 tokens meet in LDS behind the staging tiles,
    // wave 0 adds the four token-waves in a fixed order, publishes the tile's pairs and - GroupNorm tail - draws the image's ticket while the
    // other waves are already storing (gn_tail.h)
    float* const sb = (float*)(smem + 8 * 32 * ROWB);                 // [8 waves][E / 2][2]
    unsigned* const tail_flag = (unsigned*)(sb + 8 * (NO / 2) * 2);
    const bool tail_on = p.ystats != nullptr && p.tail.coef != nullptr;
    const int img = m0 / (p.HW > 0 ? p.HW : 1);
    if (p.ystats) {
#pragma unroll
        for (int i = 0; i < FC2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = rs_sum16(o[i][0][r] + o[i][1][r]), q = rs_sum16(fmaf(o[i][0][r], o[i][0][r], o[i][1][r] * o[i][1][r]));
                if (lr == 0) { sb[(wave * (NO / 2) + i * 16 + lg * 4 + r) * 2] = a; sb[(wave * (NO / 2) + i * 16 + lg * 4 + r) * 2 + 1] = q; }
            }
        __syncthreads();
        if (wave == 0) {
            float* dst = p.ystats + (((long long)img * (p.HW >> 7) + ((m0 - img * p.HW) >> 7)) * p.ystats_ld) * 2;
            for (int c = lane; c < NO; c += 64) {
                const int hw_ = c / (NO / 2), cl = c - hw_ * (NO / 2);   // channel-wave, channel inside its half
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { a += sb[((hw_ * 4 + w4) * (NO / 2) + cl) * 2]; q += sb[((hw_ * 4 + w4) * (NO / 2) + cl) * 2 + 1]; }
                if (tail_on) rs_pub_pair(dst + c * 2, a, q);
                else { dst[c * 2] = a; dst[c * 2 + 1] = q; }
            }
            if (tail_on) { const bool last = rs_gn_tail_arrive(p.tail, img); if (lane == 0) *tail_flag = last ? 1u : 0u; }
        }
    }
    RS_MLP_STAMP(5);
    constexpr int CPR = (NO / 2) / 8, NITEM = 32 * CPR;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < FC2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) { f16 hh, ll; rs_split(o[i][j][r], hh, ll); h[r] = half ? ll : hh; }
                *(f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = h;
            }
        RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
        for (int idx = lane; idx < NITEM; idx += 64) {
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int m = m0 + wp * 32 + row;
            if (m >= p.M) continue;
            *(uint4*)(p.y + (long long)m * p.ldy * 2 + half * p.ldy + wc * (NO / 2) + c8 * 8) = *(const uint4*)(stg + row * ROWB + c8 * 16);
        }
        RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
    }
    RS_MLP_STAMP(6);
    if (tail_on) {   // (kernel-uniform) behind this barrier the LDS is free: the image's last workgroup turns the tile partials into coefficients
        __syncthreads();
        if (*tail_flag) rs_gn_tail_finish<512>(p.tail, img, (float*)smem);
    }
}

}  // namespace

#ifdef RS_SPLIT_ABLATE
// ablate builds: mean cycles between consecutive stamps 0 .. 6 of wave 0 over the first `nwg` workgroups of the last fused-MLP launch
extern "C" int rs_mlp_phase_cycles(int nwg, double* out6) {
    static long long h[8 * 4096];
    if (nwg < 1 || nwg > 4096) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mlp_clk), sizeof(long long) * 8 * nwg) != hipSuccess) return -1;
    for (int k = 0; k < 6; ++k) {
        out6[k] = 0.0;
        for (int i = 0; i < nwg; ++i) out6[k] += (double)(h[8 * i + k + 1] - h[8 * i + k]) / nwg;
    }
    return 0;
}
#endif

// split storage: x / res / y tensors of (hi, lo) pairs, w1 / w2 packed [rows][K hi | K lo]
// patch_unembed folded into the layer's last fused MLP (the kernel's NO != E form): the output widths this build instantiates
extern "C" int rs_swin_mlp_split_unembed_supported(int E, int HD, int NO) { return E == 192 && HD == 768 && NO == 160; }

// `NO` = 0 (or E): the block alone, y = x' + fc2(gelu(fc1(norm2(x)))) with `res` = x'.  NO != E: + patch_unembed - `w2` is the product matrix
// [NO][HD + E] (hi | lo), `b2` the merged bias (NO floats), `res` must be null (the shortcut is the RAW `x`, which therefore has to be the
// block's own input: xcoef set), y has NO channels.
extern "C" int rs_swin_mlp_split_launch_n(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y,
                                          int M, int ldx, int ldres, int ldy, int E, int HD, int NO, const float* xcoef, int HW, float* ystats,
                                          int ystats_ld, const GNTail* tail, hipStream_t st) {
    if (NO <= 0) NO = E;
    if (ystats && (HW <= 0 || (HW & 127) || (M & 127))) return -2;   // one statistics set per 128-token tile of one image
    if (tail && tail->coef && !ystats) return -2;
    if (!rs_swin_mlp_supported(E, HD) || (ldx & 7) || (ldy & 7) || (res && (ldres & 3)) || M <= 0) return -2;
    if (NO != E && (!rs_swin_mlp_split_unembed_supported(E, HD, NO) || res || !xcoef)) return -2;
    if (xcoef && (HW <= 0 || HW % 128)) return -2;
    MlpSplitParams p{};
    p.x = (const f16*)x; p.w1 = (const f16*)w1; p.b1 = b1; p.w2 = (const f16*)w2; p.b2 = b2; p.res = (const f16*)res; p.y = (f16*)y;
    p.M = M; p.ldx = ldx; p.ldres = ldres; p.ldy = ldy; p.xcoef = xcoef; p.HW = HW; p.ystats = ystats; p.ystats_ld = ystats_ld;
    if (tail && tail->coef) {   // GroupNorm tail: this launch's statistics are segment 0; every 128-token tile of an image arrives once
        p.tail = *tail;
        p.tail.expected = HW / 128;
        p.tail.st0 = ystats; p.tail.S0 = HW / 128; p.tail.ld0 = ystats_ld; p.tail.n0 = NO;
    }
    constexpr int LDS = 2 * 6 * 32 * 128 + 2 * 192 * 128 + 128 * 128;   // 114688 (NO = 160: 8 KB less; one size keeps it simple)
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)swin_mlp_split_kernel<192, 768>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)swin_mlp_split_kernel<192, 768, 160>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    if (NO == 160) hipLaunchKernelGGL((swin_mlp_split_kernel<192, 768, 160>), dim3((M + 127) / 128), dim3(512), LDS, st, p);
    else hipLaunchKernelGGL((swin_mlp_split_kernel<192, 768>), dim3((M + 127) / 128), dim3(512), LDS, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int rs_swin_mlp_split_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* res, void* y,
                                        int M, int ldx, int ldres, int ldy, int E, int HD, const float* xcoef, int HW, float* ystats, int ystats_ld,
                                        const GNTail* tail, hipStream_t st) {
    return rs_swin_mlp_split_launch_n(x, w1, b1, w2, b2, res, y, M, ldx, ldres, ldy, E, HD, E, xcoef, HW, ystats, ystats_ld, tail, st);
}
