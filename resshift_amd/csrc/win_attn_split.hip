// Fused Swin window attention for SPLIT storage (RS_F16S, common.h): GroupNorm affine (norm1) + qkv Linear + W-MSA / SW-MSA +
// output projection + shortcut in ONE launch, every product formed from (hi, lo) fp16 pairs with three fp16 MFMAs, i.e. the
// fp32-class counterpart of win_attn_qkv_kernel (norm_attn.hip).  Replaces, per SwinTransformerBlock (models/swin_transformer.py:
// 85-145, 238-277): the GroupNorm apply pass, the qkv implicit GEMM (its [M][3E] split tensor is 302 MB at batch 32 on the 64 x 64
// level: written once, read once), win_attn_split_kernel and the projection GEMM with its residual.
//
// One workgroup per 8 x 8 window, one wave per head (6 heads of 32: every shipped config).
//   1. the window's 64 tokens -> LDS by LDS-DMA, hi plane and lo plane in the row / swizzle format of the implicit-GEMM kernels
//      (roll + window_partition folded into the addressing); with `xcoef` the GroupNorm affine is applied there (join, fma, split);
//   2. wave h projects them with ITS 96 rows of the qkv weight (fragments straight from L2 in MFMA A-operand layout):
//      acc = (2^11 Wh).Xh + Wh.Xl + Wl.Xh in ONE accumulator (the 2^11 scaling of the hi fragment is exact for |w| < 32, checked when
//      the weights are packed - igemm4.hip uses the same form), bias pre-scaled in the accumulator; q and k stay in registers as (hi,
//      lo) fragments - the accumulator layout IS the attention operand layout - and V^T goes through LDS;
//   3. S^T = Kh Qh^T + 2^-11 (Kh Ql^T + Kl Qh^T), + relative position bias + shift mask, softmax in fp32 (expf, IEEE division),
//      O^T = V^T P with P split on the fly (as win_attn_split_kernel);
//   4. the heads' results meet in LDS (the token tile's space, same format) and wave h produces output features 32h .. 32h + 31 of
//      the projection, adds the shortcut and stores the (hi, lo) pair.
#include "common.h"
#include "gn_tail.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
__device__ __forceinline__ void lds_dma16_ws(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr3_t)lds, 16, voff, 0, 0, 0);
}

#ifdef RS_SPLIT_ABLATE
__device__ long long g_attns_clk[16 * 4096];   // phase stamps of wave 0 of the first 4096 workgroups
#define RS_ATTN_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) { const int wg_ = blockIdx.y * gridDim.x + blockIdx.x; if (wg_ < 4096) g_attns_clk[16 * wg_ + (k)] = clock64(); } } while (0)
#else
#define RS_ATTN_STAMP(k)
#endif

__global__ __launch_bounds__(384) void win_attn_qkv_split_kernel(WinAttnParams p, unsigned x_bytes, unsigned res_bytes) {
    constexpr int HD = 32, WS = 8, NT = 64, VP = NT + 8, E = 192, KS = E / 32;
    constexpr int XS_STAGE = NT * 128;          // 64 token rows x 128 B per 64-wide K stage
    constexpr int XS_PLANE = 3 * XS_STAGE;      // hi (or lo) plane of the token tile
    constexpr int VT_HEAD = 2 * HD * VP;        // halfs: [HD][VP] hi, then [HD][VP] lo of one head
    constexpr int BT_BYTES = 6 * 1024 + 256 + 2048;   // bias table [6 heads][256 floats] + the tail flag's slot + the image's GroupNorm coefficients
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int nwx = p.W / WS;
    const int b = blockIdx.y;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x - wy * nwx;
    RS_ATTN_STAMP(0);
    f16* vth = (f16*)(smem + 2 * XS_PLANE) + (size_t)h * VT_HEAD;
    f16* vtl = vth + HD * VP;
    // relative position bias: the 225 distinct values per head (swin_transformer.py:93-102) copied once from the dense [h][i][j] table
    // into LDS; the softmax reads them with constant offsets instead of 96 KB per window through L2 (as win_attn_qkv_kernel)
    float* const btab = (float*)(smem + 2 * XS_PLANE + 6 * VT_HEAD * 2);
    // residual / output tile (fused projection only; hi plane, lo plane in the token tile format): the shortcut's rows arrive by LDS-DMA,
    // the projection adds its result in place and the finished tile leaves as whole 128-byte lines
    char* const rt = smem + 2 * XS_PLANE + 6 * VT_HEAD * 2 + BT_BYTES;
    // (GroupNorm tail: one LDS word in the slack behind the bias table says whether a wave of this workgroup drew the image's last ticket)
    unsigned* const tail_flag = (unsigned*)(btab + 6 * 256);
    const bool tail_on = p.tail.coef != nullptr && p.ystats != nullptr;
    if (tid == 0) *tail_flag = 0u;
    // The table arrives compact and in units of log2 (WinAttnParams::bias_c: the softmax below works in base 2): wave h copies head h's 1 KB with ONE
    // LDS-DMA instruction, requested in front of the token tile - it lands behind the same wait + barrier.  (Until round 6 every thread gathered
    // its 3 - 4 values from the dense [h][i][j] table in a loop of dependent loads BEFORE the token requests went out: four exposed round trips
    // at the head of every window's 30 us.)
    {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias_c, 0, 6 * 1024, 0x00020000);
        lds_dma16_ws(rb, (char*)btab + h * 1024, (unsigned)(h * 1024 + lane * 16));
    }
    // norm1's coefficients of this image ([2][E] floats = 1536 B) the same way: the fold below reads them from LDS instead of waiting for four
    // dependent L2 round trips behind the barrier (waves 0 and 1, 1 KB each; the descriptor's bound zero-fills the last 512 B)
    float* const cf = (float*)((char*)btab + 6 * 1024 + 256);
    if (p.xcoef && h < 2) {
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xcoef + (long long)b * 2 * E), 0, 2 * E * 4, 0x00020000);
        lds_dma16_ws(rc, (char*)cf + h * 1024, (unsigned)(h * 1024 + lane * 16));
    }
    const int shift = p.shift, H = p.H, W = p.W;
    auto pixel = [&](int t) -> long long {
        int sy = wy * WS + (t >> 3) + shift; if (sy >= H) sy -= H;
        int sx = wx * WS + (t & 7) + shift; if (sx >= W) sx -= W;
        return ((long long)b * H + sy) * W + sx;
    };
    // ---- tokens -> LDS: 2 planes x 3 K stages x 8 row groups = 48 LDS-DMA instructions, 8 per wave
    {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
        const int rsub = lane >> 3, kcp = (lane & 7) ^ (rsub & 7);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int it = h * 8 + q, plane = it / 24, r24 = it - plane * 24, st = r24 >> 3, grp = r24 & 7;
            // pixel record [ldx halfs hi | ldx halfs lo]
            const unsigned off = (unsigned)(pixel(grp * 8 + rsub) * p.ldx * 2 + plane * p.ldx + st * 64 + kcp * 8) * 2u;
            lds_dma16_ws(rx, smem + plane * XS_PLANE + st * XS_STAGE + (grp * 8) * 128, off);
        }
    }
    // the hand-counted vmcnt(24) below assumes the bias table's and the 8 token DMAs are OLDER than the 24 q-weight loads: pin that order (ae_attn.hip does the same)
    __builtin_amdgcn_sched_barrier(0);
    const int swz[2] = {(lg ^ (lr & 7)) << 4, ((4 + lg) ^ (lr & 7)) << 4};
    // one projection pass over the token tile: 32 output features starting at weight row n0 (rows [K hi | K lo], K = E) ->
    // acc[2 feature frags][4 token frags], final values (bias added, 2^-11 applied)
    // The 32 weight rows of a pass are fetched from L2 in ONE burst (24 sixteen-byte loads per lane, 96 VGPRs) in front of the MFMAs: one
    // exposed L2 round trip per pass; fetched k-step by k-step next to their MFMAs they were six round trips in a row (27.7 ms per
    // parity pass for this kernel, profiles/r3_*).
    auto load_w = [&](const f16* wsrc, int n0, f16x8 (&wh)[2][KS], f16x8 (&wl)[2][KS]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // fragment-major weights (engine.hip ConvW::ws_frag): per (16-row block, k step) 1 KB of hi then 1 KB of lo, lane-contiguous
            const f16* wr = wsrc + (long long)((n0 >> 4) + f) * KS * 1024 + lane * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wh[f][ks] = *(const f16x8*)(wr + ks * 1024);
                wl[f][ks] = *(const f16x8*)(wr + ks * 1024 + 512);
            }
        }
    };
    // (the pass's bias values are requested by the caller IN FRONT of the pass's weight burst: younger than it, the first MFMA waited for the whole burst)
    auto load_b = [&](const float* bias, int n0, f32x4 (&bv)[2]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) bv[f] = *(const f32x4*)(bias + n0 + 16 * f + 4 * lg);
    };
    auto project = [&](const f16x8 (&wh)[2][KS], const f16x8 (&wl)[2][KS], const f32x4 (&bias)[2], f32x4 (&acc)[2][4]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const f32x4 bv = bias[f] * RS_LO_SCALE;
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[f][fi] = bv;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 xh[4], xl[4];
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const char* xr = smem + (ks >> 1) * XS_STAGE + (16 * fi + lr) * 128 + swz[ks & 1];
                xh[fi] = *(const f16x8*)xr;
                xl[fi] = *(const f16x8*)(xr + XS_PLANE);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const f16x8 ws = wh[f][ks] * (f16)RS_LO_SCALE;   // v_pk_mul_f16: exact (|w| < 32)
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws, xh[fi], acc[f][fi], 0, 0, 0);
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[f][ks], xl[fi], acc[f][fi], 0, 0, 0);
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[f][ks], xh[fi], acc[f][fi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[f][fi] = acc[f][fi] * RS_LO_INV;
    };
    // The same pass with the operands swapped (the fragment registers of the two operands have the same shape): acc[f][fi] holds TOKENS
    // 16 fi + 4 lg + r of feature 16 f + lr - four consecutive tokens of one feature per lane, i.e. eight-byte pieces of a V^T row (the
    // feature-major form needed 64 two-byte LDS stores per lane for V^T, this one 16 eight-byte stores).
    auto project_t = [&](const f16x8 (&wh)[2][KS], const f16x8 (&wl)[2][KS], const float (&bias)[2], f32x4 (&acc)[2][4]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float bv = bias[f] * RS_LO_SCALE;
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[f][fi] = f32x4{bv, bv, bv, bv};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 xh[4], xl[4];
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const char* xr = smem + (ks >> 1) * XS_STAGE + (16 * fi + lr) * 128 + swz[ks & 1];
                xh[fi] = *(const f16x8*)xr;
                xl[fi] = *(const f16x8*)(xr + XS_PLANE);
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const f16x8 ws = wh[f][ks] * (f16)RS_LO_SCALE;
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[fi], ws, acc[f][fi], 0, 0, 0);
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[fi], wh[f][ks], acc[f][fi], 0, 0, 0);
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[fi], wl[f][ks], acc[f][fi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[f][fi] = acc[f][fi] * RS_LO_INV;
    };
    // accumulator pair of a token fragment -> the 8 head-dim values of that token as (hi, lo) MFMA operands
    auto pack = [&](const f32x4 (&acc)[2][4], f16x8 (&oh)[4], f16x8 (&ol)[4]) {
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16 a, c;
                rs_split(acc[0][fi][r], a, c); oh[fi][r] = a; ol[fi][r] = c;
                rs_split(acc[1][fi][r], a, c); oh[fi][4 + r] = a; ol[fi][4 + r] = c;
            }
    };
    const f16* wq = (const f16*)p.wqkv;
    f16x8 wfh[2][KS], wfl[2][KS];
    f32x4 bq[2];
    load_b(p.bqkv, h * HD, bq);
    load_w(wq, h * HD, wfh, wfl);     // q_h weights: requested together with the token tile ...
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // ... but only the tile (8 LDS-DMA instructions, older than the 24 weight loads) is waited for here
    __syncthreads();   // the window's tokens are in LDS
    RS_ATTN_STAMP(1);
    if (p.xcoef) {
        // GroupNorm (norm1) folded in: x * scale[b][c] + shift[b][c] on the joined value, re-split.  64 rows x 24 chunks of 8
        // channels, 4 chunks per thread; LDS position ps of row t holds chunk ps ^ (t & 7).
        const float* sc = cf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int item = tid + 384 * q;              // 0 .. 1535
            const int st = item >> 9, t = (item >> 3) & 63, ps = item & 7;
            const int c0 = st * 64 + ((ps ^ (t & 7)) << 3);
            f16x8* ch_ = (f16x8*)(smem + st * XS_STAGE + t * 128 + ps * 16);
            f16x8* cl_ = (f16x8*)(smem + XS_PLANE + st * XS_STAGE + t * 128 + ps * 16);
            f16x8 vh = *ch_, vl = *cl_;
            const f32x4 a0 = *(const f32x4*)(sc + c0), a1 = *(const f32x4*)(sc + c0 + 4);
            const f32x4 d0 = *(const f32x4*)(sc + E + c0), d1 = *(const f32x4*)(sc + E + c0 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f16 a, c;
                rs_split(fmaf(rs_join(vh[e], vl[e]), e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? d0[e & 3] : d1[e & 3]), a, c);
                vh[e] = a; vl[e] = c;
            }
            *ch_ = vh; *cl_ = vl;
        }
        __syncthreads();
    }
    RS_ATTN_STAMP(2);
    f16x8 kh[4], kl[4], qh[4], ql[4];
    {
        f32x4 acc[2][4];
        project(wfh, wfl, bq, acc);                        // q_h: lane (lr, lg) holds d = {4 lg + r, 16 + 4 lg + r} of token 16 fi + lr
        load_b(p.bqkv, E + h * HD, bq);
        load_w(wq, E + h * HD, wfh, wfl);
        pack(acc, qh, ql);
        __builtin_amdgcn_sched_barrier(0);
        RS_ATTN_STAMP(3);
        project(wfh, wfl, bq, acc);                        // k_h: the same d set per lane -> a consistent contraction order for S^T
        float bvv[2] = {p.bqkv[2 * E + h * HD + lr], p.bqkv[2 * E + h * HD + 16 + lr]};
        load_w(wq, 2 * E + h * HD, wfh, wfl);
        pack(acc, kh, kl);
        __builtin_amdgcn_sched_barrier(0);
        RS_ATTN_STAMP(4);
        project_t(wfh, wfl, bvv, acc);                     // v_h, tokens along the accumulator rows -> V^T[d][token] (hi, lo) in LDS
#ifdef RS_ATTN_SPLIT_EARLYW   // (projection weights in flight during the whole attention: 96 registers the softmax then spills; measured equal)
        if (p.wproj) load_w((const f16*)p.wproj, h * HD, wfh, wfl);
#endif
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                f16x4 hv, lv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f16 a, c;
                    rs_split(acc[f][ft][r], a, c);
                    hv[r] = a; lv[r] = c;
                }
                *(f16x4*)(vth + (16 * f + lr) * VP + 16 * ft + 4 * lg) = hv;
                *(f16x4*)(vtl + (16 * f + lr) * VP + 16 * ft + 4 * lg) = lv;
            }
    }
    RS_ATTN_STAMP(5);
    __syncthreads();  // V^T of every head is in LDS; every wave is done with the token tile (it is overwritten below)
    RS_ATTN_STAMP(6);
    if (p.wproj && p.res) {
        // the shortcut's rows -> the residual / output tile by LDS-DMA, requested HERE: memory operations complete in order, so the next
        // wait (the projection weights, behind the attention) is the first that includes them - they travel during the whole attention
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, res_bytes, 0x00020000);
        const int rsub = lane >> 3, kcp = (lane & 7) ^ (rsub & 7);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int it = h * 8 + q, plane = it / 24, r24 = it - plane * 24, st = r24 >> 3, grp = r24 & 7;
            const unsigned off = (unsigned)(pixel(grp * 8 + rsub) * p.ldres * 2 + plane * p.ldres + st * 64 + kcp * 8) * 2u;
            lds_dma16_ws(rr, rt + plane * XS_PLANE + st * XS_STAGE + (grp * 8) * 128, off);
        }
    }
    f32x4 s[4][4];  // [fj][fi]
#pragma unroll
    for (int fj = 0; fj < 4; ++fj)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[fj], ql[fi], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[fj], qh[fi], c, 0, 0, 0);
            const f32x4 m = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[fj], qh[fi], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[fj][fi][r] = fmaf(c[r], RS_LO_INV, m[r]);
        }
    // softmax over the keys (per query column i = 16 fi + lr) in base 2, fp32: the scores become s * (scale log2 e) + (bias log2 e) (the table
    // is stored pre-multiplied) and the exponential is the bare v_exp_f32 (1 ulp; expf's range reduction bought nothing here: the arguments
    // are <= 0 and the result is split to 22 bits right after).  The row sums come out of the P V matrix product (a row of ones appended to
    // V^T, below).  Only windows of the LAST window row can carry the shift mask (bands of window_row * 8 + token_column, see
    // win_attn_kernel): every other window takes the mask-free path (a wave-uniform branch).
    const float c2 = p.scale * 1.44269504088896f;
    // bias of (i = 16 fi + lr, j = 16 fj + 4 lg + r) = tb[30 (fi - fj) - r]
    const float* tb = btab + h * 256 + ((lr >> 3) - (lg >> 1) + 7) * 15 + (lr & 7) - 4 * (lg & 1) + 7;
    auto softmax = [&](auto MK) {
        constexpr bool MASK = decltype(MK)::value;
        int rid_i = 0, rid_j[4] = {0, 0, 0, 0};
        if constexpr (MASK) {   // region ids of the (quirky) shift mask: band of window_row*8 + token_column (see win_attn_kernel)
            auto band = [&](int c) { const int yq = wy * WS + c; return yq < H - WS ? 0 : (yq < H - shift ? 1 : 2); };
            rid_i = band(lr & 7);
#pragma unroll
            for (int r = 0; r < 4; ++r) rid_j[r] = band(4 * (lg & 1) + r);
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            __builtin_amdgcn_sched_barrier(0);   // one query fragment's 16 table reads at a time (all 64 up front do not fit 256 registers)
            float m = -3.0e38f;
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(s[fj][fi][r], c2, tb[30 * (fi - fj) - r]);
                    if constexpr (MASK) { if (rid_j[r] != rid_i) v += -100.0f * 1.44269504088896f; }
                    s[fj][fi][r] = v;
                    m = fmaxf(m, v);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
#pragma unroll
            for (int fj = 0; fj < 4; ++fj)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[fj][fi][r] = __builtin_amdgcn_exp2f(s[fj][fi][r] - m);
        }
    };
    if (shift > 0 && wy == H / WS - 1) softmax(std::true_type{}); else softmax(std::false_type{});
    RS_ATTN_STAMP(7);   // scores + softmax done
    f32x4 om[2][4], oc[2][4];    // [fd][fi] main / cross
    f32x4 lm[4], lc[4];          // the row of ones: sum over the keys of P's hi / lo parts, for every lane of column i
    const f16x8 ones = f16x8{(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        lm[fi] = f32x4{0.f, 0.f, 0.f, 0.f}; lc[fi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) { om[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f16x8 vah[2], val[2], pbh[4], pbl[4];
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) {
            const int o = (16 * fd + lr) * VP + 32 * ks + 4 * lg;
            const f16x4 a0 = *(const f16x4*)(vth + o), a1 = *(const f16x4*)(vth + o + 16);
            const f16x4 b0 = *(const f16x4*)(vtl + o), b1 = *(const f16x4*)(vtl + o + 16);
            vah[fd] = f16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            val[fd] = f16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16 a, bq;
                rs_split(s[2 * ks][fi][r], a, bq);     pbh[fi][r] = a;     pbl[fi][r] = bq;
                rs_split(s[2 * ks + 1][fi][r], a, bq); pbh[fi][4 + r] = a; pbl[fi][4 + r] = bq;
            }
#pragma unroll
        for (int fd = 0; fd < 2; ++fd)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                om[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[fd], pbh[fi], om[fd][fi], 0, 0, 0);
                oc[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[fd], pbl[fi], oc[fd][fi], 0, 0, 0);
                oc[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(val[fd], pbh[fi], oc[fd][fi], 0, 0, 0);
            }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            lm[fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pbh[fi], lm[fi], 0, 0, 0);
            lc[fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pbl[fi], lc[fi], 0, 0, 0);
        }
    }
    float inv[4];
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) inv[fi] = 1.0f / fmaf(lc[fi][0], RS_LO_INV, lm[fi][0]);
    f16* out = (f16*)p.out;
    const long long ro = 2LL * p.ldo;   // output pixel record [ldo hi | ldo lo]
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) {
            f16x4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16 a, bq;
                rs_split(fmaf(oc[fd][fi][r], RS_LO_INV, om[fd][fi][r]) * inv[fi], a, bq);
                hv[r] = a; lv[r] = bq;
            }
            const int t = 16 * fi + lr, c = h * HD + 16 * fd + 4 * lg;     // token row, feature
            if (!p.wproj) {
                f16* dst = out + pixel(t) * ro + c;
                *(f16x4*)dst = hv;
                *(f16x4*)(dst + p.ldo) = lv;
            } else {
                // fused output projection: the heads' results meet in LDS (the token tile's space, same row / swizzle format)
                char* cell = smem + (c >> 6) * XS_STAGE + t * 128 + ((((c & 63) >> 3) ^ (t & 7)) << 4) + (c & 7) * 2;
                *(f16x4*)cell = hv;
                *(f16x4*)(cell + XS_PLANE) = lv;
            }
        }
    if (!p.wproj) return;
    RS_ATTN_STAMP(8);   // P V done, results in LDS
    // ---- fused output projection: wave h produces output features 32 h .. 32 h + 31 for all tokens, weights straight from L2
    load_b(p.bproj, h * HD, bq);
#ifndef RS_ATTN_SPLIT_EARLYW
    load_w((const f16*)p.wproj, h * HD, wfh, wfl);
#endif
    const bool has_res = p.res != nullptr;
    __syncthreads();   // all heads' attention results are in LDS
    RS_ATTN_STAMP(9);
    f32x4 acc2[2][4];
    project(wfh, wfl, bq, acc2);
    float s1[2][4], s2[2][4];   // per-channel sums of the stored values (the pair reproduces v to 2^-23) over this lane's four tokens
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[f][r] = 0.f; s2[f][r] = 0.f; }
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // cell of (token t, features c .. c + 3) in the residual / output tile: read and rewritten by this lane only
            const int t = 16 * fi + lr, c = h * HD + 16 * f + 4 * lg;
            char* cell = rt + (c >> 6) * XS_STAGE + t * 128 + ((((c & 63) >> 3) ^ (t & 7)) << 4) + (c & 7) * 2;
            f16x4 hv, lv;
            if (has_res) { hv = *(const f16x4*)cell; lv = *(const f16x4*)(cell + XS_PLANE); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16 a, c2;
                const float v = acc2[f][fi][r] + (has_res ? rs_join(hv[r], lv[r]) : 0.f);
                rs_split(v, a, c2);
                hv[r] = a; lv[r] = c2;
                if (p.ystats) { s1[f][r] += v; s2[f][r] = fmaf(v, v, s2[f][r]); }
            }
            *(f16x4*)cell = hv;
            *(f16x4*)(cell + XS_PLANE) = lv;
        }
    // Statistics for norm2 (and, with WinAttnParams::tail, its coefficients: gn_tail.h): every wave holds its 32 features of all 64 tokens
    // of the window; the six waves' sums meet in LDS (the V^T space: dead since the projection began), wave 0 alone publishes the window's
    // 192 pairs and draws ONE ticket per window behind the barrier that completes the output tile - the other waves go on to the stores.
    // (A ticket per wave - 384 arrivals per image on one address - cost the kernel 24 us per launch: same-address atomics serialise.)
    float* const st_lds = (float*)(smem + 2 * XS_PLANE);
    if (p.ystats) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = rs_sum16(s1[f][r]), q = rs_sum16(s2[f][r]);   // (DPP adds: same bits as the xor-shuffle butterfly)
                if (lr == 0) { st_lds[(h * HD + 16 * f + 4 * lg + r) * 2] = a; st_lds[(h * HD + 16 * f + 4 * lg + r) * 2 + 1] = q; }
            }
    }
    __syncthreads();   // the output tile is complete (and the window's statistics are in LDS)
    if (p.ystats && h == 0) {
        float* dst = p.ystats + (((long long)b * (nwx * (H / WS)) + blockIdx.x) * p.ystats_ld) * 2;
        for (int c = lane; c < E; c += 64) {
            const float a = st_lds[2 * c], q = st_lds[2 * c + 1];
            if (tail_on) rs_pub_pair(dst + 2 * c, a, q);   // write-through: read by the image's last arriver inside this launch
            else { dst[2 * c] = a; dst[2 * c + 1] = q; }
        }
        if (tail_on && rs_gn_tail_arrive(p.tail, b) && lane == 0) *tail_flag = 1u;
    }
    // whole rows out: 2 planes x 24 pieces of 8 token rows x 128 B (8 full cache lines per wave instruction), 8 per wave
    {
        const int rsub = lane >> 3, ps = lane & 7, chunk = ps ^ (rsub & 7);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int it = h * 8 + q, plane = it / 24, r24 = it - plane * 24, st = r24 >> 3, grp = r24 & 7;
            const f16x8 v = *(const f16x8*)(rt + plane * XS_PLANE + st * XS_STAGE + (grp * 8 + rsub) * 128 + ps * 16);
            *(f16x8*)(out + pixel(grp * 8 + rsub) * ro + plane * p.ldo + st * 64 + chunk * 8) = v;
        }
    }
    RS_ATTN_STAMP(10);
    if (tail_on) {   // (kernel-uniform) behind this barrier the LDS is free: the image's last workgroup turns the window partials into coefficients
        __syncthreads();
        if (*tail_flag) rs_gn_tail_finish<384>(p.tail, b, (float*)smem);
    }
}

}  // namespace

#ifdef RS_SPLIT_ABLATE
extern "C" int rs_attn_split_phase_cycles(int nwg, int nst, double* out) {
    static long long h[16 * 4096];
    if (nwg < 1 || nwg > 4096 || nst < 2 || nst > 16) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attns_clk), sizeof(long long) * 16 * nwg) != hipSuccess) return -1;
    for (int k = 0; k + 1 < nst; ++k) {
        out[k] = 0.0;
        for (int i = 0; i < nwg; ++i) out[k] += (double)(h[16 * i + k + 1] - h[16 * i + k]) / nwg;
    }
    long long lo = h[0], hi = h[nst - 1];   // span of the launch: first start .. last end over the stamped workgroups
    for (int i = 0; i < nwg; ++i) { lo = h[16 * i] < lo ? h[16 * i] : lo; hi = h[16 * i + nst - 1] > hi ? h[16 * i + nst - 1] : hi; }
    out[nst - 1] = (double)(hi - lo);
    return 0;
}
#endif

// fused qkv projection + window attention (+ output projection + shortcut) in split storage: 6 heads of 32; x / res / out are
// split-storage NHWC tensors, wqkv / wproj split weight rows [K hi | K lo]
extern "C" int rs_win_attn_qkv_split_launch(const WinAttnParams* pp, hipStream_t st) {
    const WinAttnParams& p = *pp;
    if ((p.H % 8) || (p.W % 8) || p.heads != 6 || (p.ldx % 8) || (p.ldo % 8) || !p.bias_c || !p.x || !p.wqkv || !p.bqkv) return -2;
    if (p.shift != 0 && p.shift != 4) return -2;
    if (p.wproj && (!p.bproj || (p.res && (p.ldres % 4)))) return -2;
    const size_t xb = (size_t)p.B * p.H * p.W * p.ldx * 4;
    if (xb >= 0xF0000000ull) return -2;
    const int nwin = (p.H / 8) * (p.W / 8);
    const size_t rb = p.res ? (size_t)p.B * p.H * p.W * p.ldres * 4 : 0;
    if (rb >= 0xF0000000ull || (p.wproj && p.res && (p.ldres % 8))) return -2;
    // token tile (hi, lo) + V^T (hi, lo) + the bias table (+ with the fused projection the residual / output tile)
    const size_t lds_max = (size_t)2 * 3 * 64 * 128 + (size_t)6 * 2 * 32 * (64 + 8) * sizeof(f16) + (6 * 1024 + 256 + 2048) + (size_t)2 * 3 * 64 * 128;
    const size_t lds = lds_max - (p.wproj ? 0 : (size_t)2 * 3 * 64 * 128);
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) { (void)hipFuncSetAttribute((const void*)win_attn_qkv_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max); }
    WinAttnParams q = p;
    if (q.tail.coef) {   // GroupNorm tail: this launch's statistics are segment 0; every window of an image arrives once
        if (!q.ystats || !q.wproj) return -2;
        q.tail.expected = nwin;   // (wave 0 of every window's workgroup arrives once)
        q.tail.st0 = q.ystats; q.tail.S0 = nwin; q.tail.ld0 = q.ystats_ld; q.tail.n0 = 32 * p.heads;
    }
    hipLaunchKernelGGL(win_attn_qkv_split_kernel, dim3(nwin, p.B), dim3(64 * p.heads), lds, st, q, (unsigned)xb, (unsigned)rb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
