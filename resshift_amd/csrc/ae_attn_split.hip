// Streaming ("flash") attention for the autoencoder's mid-block AttnBlock in SPLIT storage (ldm/modules/diffusionmodules/model.py:179-203:
// single head, d = C = 512 channels, T = h * w tokens, softmax(q k^T / sqrt(C)) v; the reference's own memory-efficient variant is
// :205-268) - the fp32-class counterpart of ae_attn.hip, for the encoder of the parity policy (everything in front of the VQ argmin runs
// in split precision).  Without it the split encoder materialises the T x T score matrix in HBM one block of query rows at a time (fp32
// scores + (hi, lo) probabilities: 8 bytes per score, 34 GB per 4096-row block at T = 262 144 - the reference's default 512-pixel tile).
//
// Every value is a (hi, lo) fp16 pair, x = hi + lo 2^-11 (common.h); a product is three fp16 MFMAs.  What does not carry over from the fp16
// kernel is the register budget: 16 queries x 512 channels of (hi, lo) query fragments are 128 VGPRs, the O^T accumulators another 128.
// So a workgroup is 64 queries and its 8 waves are 4 query groups x 2 CHANNEL HALVES:
//   * S^T = K Q^T: wave (qg, cw) contracts channels [256 cw, 256 cw + 256) only (query fragments: 64 VGPRs), the two partial sums meet
//     through 16 KB of LDS and are added in a fixed order (half 0 + half 1), so both waves of a query group hold the SAME bits and run the
//     same online softmax - redundant VALU work, no second exchange;
//   * O^T = V^T P: wave (qg, cw) owns output channels [256 cw, 256 cw + 256) (16 accumulator fragments).  P lies in [0, 1], so ITS hi half
//     can be scaled by 2^11 exactly in fp16 and the three products share ONE accumulator, acc = Vh.(2^11 Ph) + Vh.Pl + Vl.Ph (the halo
//     conv's form, igemm4_kernel.h, with the roles swapped: there the weight is the bounded operand, here the probability) - 64 VGPRs
//     instead of 128;
//   * key blocks of 32 tokens: K tile [32 keys][512 ch] and V^T tile [512 ch][32 keys] as (hi, lo) pairs are 64 KB each, in the
//     128-byte-row / XOR-swizzle format of the split implicit-GEMM kernels (a row = [32 hi | 32 lo] halfs), fetched by LDS-DMA; the K
//     tile's rows are permuted like the fp16 kernel's so that a lane's S^T accumulators are eight CONSECUTIVE keys of the V^T tile.
// LDS: 64 + 64 + 16 KB.  Per key block and wave 48 + 48 MFMAs against 32 + 32 ds_read_b128.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr4_t;
__device__ __forceinline__ void lds_dma16_fs(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr4_t)lds, 16, voff, 0, 0, 0);
}

struct FlashSplitParams {
    const f16* q;      // [z][T] pixel records [ldq halfs hi | ldq halfs lo] (C channels used)
    const f16* k;      // [z][T] records [ldk hi | ldk lo]
    const f16* vt;     // [z][C] rows [T halfs hi | T halfs lo]: v transposed (no bias)
    const float* bv;   // [C] v bias or null
    f16* o;            // [z][T] records [ldo hi | ldo lo]
    int T, ldq, ldk, ldo;
    float scale;       // 1 / sqrt(C)
};

template <int C>
__global__ __launch_bounds__(512, 2) void ae_flash_attn_split_kernel(FlashSplitParams p) {
    constexpr int BQ = 64, BK = 32, CH = C / 2, KSH = CH / 32, NST = C / 32, FDH = CH / 16;
    constexpr int KT = NST * BK * 128;     // K tile: NST stages (32 channels, hi | lo) of 32 rows x 128 B
    constexpr int VT = C * 128;            // V^T tile: C rows x 128 B (32 keys hi | 32 keys lo)
    constexpr int XB = 8 * 2 * 64 * 16;    // exchange of the S^T partials: [wave][row fragment][lane] x f32x4
    static_assert(C % 64 == 0 && KT + VT + XB <= 160 * 1024 && NST % 8 == 0 && (C / 8) % 8 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ks_ = smem;
    char* const vs_ = smem + KT;
    f32x4* const xs_ = (f32x4*)(smem + KT + VT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int qg = wave & 3, cw = wave >> 2;          // query group, channel half
    const long long z = blockIdx.y;
    const int q0 = blockIdx.x * BQ + qg * 16;         // this wave's queries: q0 .. q0 + 15 (lane lr -> query q0 + lr)
    const int T = p.T;
    const f16* qz = p.q + z * (long long)T * p.ldq * 2;
    const f16* kz = p.k + z * (long long)T * p.ldk * 2;
    const f16* vz = p.vt + z * (long long)C * T * 2;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kz, 0, (unsigned)min((long long)T * p.ldk * 4, 0xF0000000LL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vz, 0, (unsigned)min((long long)C * T * 4, 0xF0000000LL), 0x00020000);

    // K tile: LDS row rho (the MFMA row of S^T) holds key kappa(rho) = 8 ((rho & 15) >> 2) + 4 ((rho >> 4) & 1) + (rho & 3) of the block;
    // 1 KB DMA pieces = 8 LDS rows of one 32-channel stage: NST * 4 pieces, wave w owns pieces w, w + 8, ...  Position (lane & 7) of a row
    // receives logical 16-byte chunk kcp: plane kcp >> 2 (hi, lo), channels 8 (kcp & 3) .. of the stage (swizzle on the source side).
    const int rsub = lane >> 3, kcp = (lane & 7) ^ (rsub & 7);
    auto issue_k = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NST / 2; ++i) {
            const int piece = wave + 8 * i, st = piece >> 2, grp = piece & 3;
            const int rho = grp * 8 + rsub;
            const int kappa = 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
            const unsigned off = (unsigned)(((long long)(kb * BK + kappa) * p.ldk * 2 + (kcp >> 2) * p.ldk + st * 32 + (kcp & 3) * 8) * 2);
            lds_dma16_fs(rk, ks_ + st * (BK * 128) + (grp * 8) * 128, off);
        }
    };
    // V^T tile: LDS row d holds keys kb * 32 .. + 31 of channel d (natural order), hi half then lo half; C / 8 pieces of 8 rows
    auto issue_v = [&](int kb) {
#pragma unroll
        for (int i = 0; i < C / 64; ++i) {
            const int piece = wave + 8 * i;
            const int d = piece * 8 + rsub;
            const unsigned off = (unsigned)(((long long)d * T * 2 + (kcp >> 2) * (long long)T + kb * BK + (kcp & 3) * 8) * 2);
            lds_dma16_fs(rv, vs_ + (piece * 8) * 128, off);
        }
    };
    const int nkb = T / BK;
    // the wave's queries, channel half cw, as B-operand fragments: lane (lr, lg) holds channels 256 cw + 32 ks + 8 lg .. + 7 of query q0 + lr
    // (requested BEFORE the tiles: the counted waits below rely on the order q, K(0), V(0), K(1), V(1), ... of this wave's requests)
    f16x8 qh[KSH], ql[KSH];
    {
        const f16* qr = qz + (long long)(q0 + lr) * p.ldq * 2 + cw * CH + lg * 8;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) { qh[ks] = *(const f16x8*)(qr + ks * 32); ql[ks] = *(const f16x8*)(qr + p.ldq + ks * 32); }
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_k(0);
    issue_v(0);
    f32x4 o[FDH];   // O^T of channels 256 cw + 16 fd + 4 lg + r, query lr - carrying 2^11 x the sum
#pragma unroll
    for (int fd = 0; fd < FDH; ++fd) o[fd] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_run = 0.f;
    const int swz0 = (lg ^ (lr & 7)) << 4, swz1 = ((4 + lg) ^ (lr & 7)) << 4;   // hi / lo fragment of a row whose index is lr mod 8
    const float sc = p.scale * 1.44269504088896341f;   // scores in log2 units: exp(x) = exp2(x log2 e)

    for (int kb = 0; kb < nkb; ++kb) {
        // ---- partial S^T = K Q^T over this wave's channel half (the K tile's DMA - and every older request - has landed)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C / 64) : "memory");   // only this block's V^T pieces (the youngest C / 64 requests) may be in flight
        __builtin_amdgcn_s_barrier();
        f32x4 sm[2], sx[2];   // main (hi.hi) and cross (hi.lo + lo.hi, x 2^11) sums of the two 16-key row fragments
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) { sm[fj] = f32x4{0.f, 0.f, 0.f, 0.f}; sx[fj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            const char* stg = ks_ + (cw * KSH + ks) * (BK * 128);
            f16x8 ah[2], al[2];
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) { ah[fj] = *(const f16x8*)(stg + (16 * fj + lr) * 128 + swz0); al[fj] = *(const f16x8*)(stg + (16 * fj + lr) * 128 + swz1); }
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                sm[fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[fj], qh[ks], sm[fj], 0, 0, 0);
                sx[fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[fj], ql[ks], sx[fj], 0, 0, 0);
                sx[fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[fj], qh[ks], sx[fj], 0, 0, 0);
            }
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[fj][r] = fmaf(sx[fj][r], RS_LO_INV, sm[fj][r]);
            xs_[(wave * 2 + fj) * 64 + lane] = sm[fj];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave is done with the K tile; the partials are in LDS
        if (kb + 1 < nkb) issue_k(kb + 1);       // ... the next one arrives during the softmax and the PV product
        f32x4 s[2];
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {         // half 0 + half 1, in that order in BOTH waves of the query group: identical bits
            const f32x4 a = xs_[(qg * 2 + fj) * 64 + lane], b = xs_[((4 + qg) * 2 + fj) * 64 + lane];
            s[fj] = a + b;
        }
        // ---- online softmax of the 32 scores of query lr held by the four lane groups
        float mx = -3.0e38f;
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[fj][r] *= sc; mx = fmaxf(mx, s[fj][r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ls = 0.f;
        // B operand of the PV product (one k-step of 32 keys): element e <-> key 8 lg + e  <->  S^T row 16 (e >> 2) + 4 lg + (e & 3)
        f16x8 ph, pl, ps;   // hi, lo, and 2^11 hi (exact: P <= 1)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[fj][r] - m_new);
                f16 h, l;
                rs_split(e, h, l);
                ph[4 * fj + r] = h; pl[4 * fj + r] = l;
                ls += rs_join(h, l);   // the sum of what the PV product actually uses
            }
        ps = ph * (f16)RS_LO_SCALE;
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int fd = 0; fd < FDH; ++fd) o[fd] = o[fd] * alpha;
        // ---- O^T += V^T P for this wave's channel half (the V^T tile has landed: only the next K tile's pieces may be in flight)
        if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST / 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int fd = 0; fd < FDH; ++fd) {
            const char* row = vs_ + (cw * CH + 16 * fd + lr) * 128;
            const f16x8 vh = *(const f16x8*)(row + swz0), vl = *(const f16x8*)(row + swz1);
            o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ps, o[fd], 0, 0, 0);
            o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[fd], 0, 0, 0);
            o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[fd], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();            // every wave is done with the V^T tile (and with the exchange buffer)
        if (kb + 1 < nkb) issue_v(kb + 1);
    }
    // ---- O = O^T 2^-11 / l + bias: lane (lr, lg) holds channels 256 cw + 16 fd + 4 lg + r of query q0 + lr
    const float inv = RS_LO_INV / l_run;
    f16* orow = p.o + z * (long long)T * p.ldo * 2 + (long long)(q0 + lr) * p.ldo * 2 + cw * CH + 4 * lg;
#pragma unroll
    for (int fd = 0; fd < FDH; ++fd) {
        f32x4 bvv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bv) bvv = *(const f32x4*)(p.bv + cw * CH + 16 * fd + 4 * lg);
        f16x4 hv, lv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { f16 h, l; rs_split(fmaf(o[fd][r], inv, bvv[r]), h, l); hv[r] = h; lv[r] = l; }
        *(f16x4*)(orow + 16 * fd) = hv;
        *(f16x4*)(orow + p.ldo + 16 * fd) = lv;
    }
}

template <int C>
int launch_flash_split(const FlashSplitParams& p, int nz, hipStream_t st) {
    constexpr size_t lds = (size_t)(C / 32) * 32 * 128 + (size_t)C * 128 + 8 * 2 * 64 * 16;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) { (void)hipFuncSetAttribute((const void*)ae_flash_attn_split_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
    hipLaunchKernelGGL((ae_flash_attn_split_kernel<C>), dim3(p.T / 64, nz), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// split-storage streaming attention of the autoencoder's AttnBlock: C = 512 channels (the shipped autoencoders' mid block), T a multiple
// of 64, 16-byte aligned rows
extern "C" int rs_ae_flash_split_supported(int C, int T) { return C == 512 && T >= 64 && (T % 64) == 0; }
extern "C" int rs_ae_flash_split_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, const float* bv, void* o, int ldo, int nz, int T,
                                        int C, float scale, hipStream_t st) {
    if (!rs_ae_flash_split_supported(C, T) || (ldq & 7) || (ldk & 7) || (ldo & 3)) return -2;
    if ((long long)T * ldk * 4 >= 0xF0000000LL || (long long)C * T * 4 >= 0xF0000000LL) return -2;   // 32-bit buffer offsets per image
    FlashSplitParams p{};
    p.q = (const f16*)q; p.k = (const f16*)k; p.vt = (const f16*)vt; p.bv = bv; p.o = (f16*)o;
    p.T = T; p.ldq = ldq; p.ldk = ldk; p.ldo = ldo; p.scale = scale;
    return launch_flash_split<512>(p, nz, st);
}
