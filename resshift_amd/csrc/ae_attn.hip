// Streaming ("flash") attention for the autoencoder's mid-block AttnBlock in fp16 storage (ldm/modules/diffusionmodules/model.py:
// 179-203: single head, d = C = 512 channels, T = h * w tokens, softmax(q k^T / sqrt(C)) v; the reference's own memory-efficient variant
// is :205-268).  The row-block path of engine.hip materialises the T x T score matrix in HBM one block of query rows at a time (fp32
// scores + fp16 probabilities: 12.9 GB per 4096-row block at T = 262 144, the reference's default 512-pixel tile) - more than half of
// such a tile's 1.40 s (profiles/r3_tiled_chop512.txt).  Here S never leaves the chip.
//
// One workgroup = 128 query tokens (8 waves x 16 queries), looping over key blocks of 64 tokens with an online softmax:
//   * the wave's 16 queries live in registers for the whole kernel as MFMA B-operand fragments (C / 32 fragments: 64 VGPRs at C = 512);
//   * S^T = K Q^T (keys as rows): the K tile [64 keys][C] sits in LDS in the 128-byte-row / XOR-swizzle format of the implicit-GEMM
//     kernels (C / 64 stages of 64 rows), fetched by LDS-DMA; a lane ends with scores of ONE query (column lr) - so the running max, the
//     running sum and the rescale factor of the output are per-lane scalars, reduced over the four lane groups with two xor-shuffles;
//   * O^T = V^T P (channels as rows): the accumulators of S^T ARE the B operand of this product once converted to fp16; their key order per
//     lane group is {4 lg + r} u {16 + 4 lg + r} within each 32 keys, so the K tile is loaded with its ROWS permuted (LDS-DMA rows are
//     free to come from anywhere) such that those eight keys are eight CONSECUTIVE tokens - and the V^T tile [C][64 keys] (v is produced
//     transposed by the v-projection GEMM) is read with plain 16-byte fragment loads.  Softmax is invariant under the permutation;
//   * O^T stays in registers (C / 16 fragments: 128 VGPRs at C = 512), rescaled by exp(m_old - m_new) per key block, divided by the sum
//     and given the v bias at the end (softmax rows sum to 1).
// LDS: K tile + V^T tile = 2 x 64 KB at C = 512; the next K tile is requested as soon as S^T is done, the next V^T tile as soon as PV is.
// Per key block and wave: 64 + 64 MFMAs against 64 + 64 ds_read_b128 - every wave reads both tiles completely, so LDS bandwidth and the
// matrix pipe are about equally loaded (1 MB of fragment reads per block and CU): the price of d = 512 on 160 KB of LDS.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
__device__ __forceinline__ void lds_dma16_fa(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr3_t)lds, 16, voff, 0, 0, 0);
}

struct FlashParams {
    const f16* q;      // [z][T][ldq]   (C channels used)
    const f16* k;      // [z][T][ldk]
    const f16* vt;     // [z][C][T]     v transposed (no bias)
    const float* bv;   // [C] v bias or null
    f16* o;            // [z][T][ldo]
    int T, ldq, ldk, ldo;
    float scale;       // 1 / sqrt(C)
};

template <int C>
__global__ __launch_bounds__(512, 2) void ae_flash_attn_kernel(FlashParams p) {
    constexpr int BQ = 128, BK = 64, KS = C / 32, NST = C / 64, FD = C / 16;
    constexpr int KT = NST * BK * 128;     // K tile bytes: NST stages of 64 rows x 128 B
    constexpr int VT = C * 128;            // V^T tile bytes: C rows x 128 B (64 keys)
    static_assert(C % 64 == 0 && KT + VT <= 160 * 1024, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ks_ = smem;
    char* const vs_ = smem + KT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const long long z = blockIdx.y;
    const int q0 = blockIdx.x * BQ + wave * 16;       // this wave's queries: q0 .. q0 + 15 (lane lr -> query q0 + lr)
    const int T = p.T;
    const f16* qz = p.q + z * (long long)T * p.ldq;
    const f16* kz = p.k + z * (long long)T * p.ldk;
    const f16* vz = p.vt + z * (long long)C * T;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kz, 0, (unsigned)min((long long)T * p.ldk * 2, 0xF0000000LL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vz, 0, (unsigned)min((long long)C * T * 2, 0xF0000000LL), 0x00020000);

    // K tile: LDS row rho (the MFMA row of S^T) holds key kappa(rho) = 32 (rho >> 5) + 8 ((rho & 15) >> 2) + 4 ((rho >> 4) & 1) + (rho & 3)
    // of the block; 1 KB DMA pieces = 8 LDS rows of one 64-channel stage: NST * 8 pieces, wave w owns pieces w, w + 8, ...
    const int rsub = lane >> 3, kcp = (lane & 7) ^ (rsub & 7);
    auto issue_k = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int piece = wave + 8 * i, st = piece >> 3, grp = piece & 7;
            const int rho = grp * 8 + rsub;
            const int kappa = 32 * (rho >> 5) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
            const unsigned off = (unsigned)(((long long)(kb * BK + kappa) * p.ldk + st * 64 + kcp * 8) * 2);
            lds_dma16_fa(rk, ks_ + st * (BK * 128) + (grp * 8) * 128, off);
        }
    };
    // V^T tile: LDS row d holds keys kb * 64 .. + 63 of channel d (natural order); C / 8 pieces of 8 rows, wave w owns pieces w, w + 8, ...
    auto issue_v = [&](int kb) {
#pragma unroll
        for (int i = 0; i < C / 64; ++i) {
            const int piece = wave + 8 * i;
            const int d = piece * 8 + rsub;
            const unsigned off = (unsigned)(((long long)d * T + kb * BK + kcp * 8) * 2);
            lds_dma16_fa(rv, vs_ + (piece * 8) * 128, off);
        }
    };
    const int nkb = T / BK;
    // the wave's queries as B-operand fragments: lane (lr, lg) holds channels 32 ks + 8 lg .. + 7 of query q0 + lr.  (Requested BEFORE the
    // tiles: the counted waits below rely on the order q, K(0), V(0), K(1), V(1), ... of this wave's vector-memory requests.)
    f16x8 qf[KS];
    {
        const f16* qr = qz + (long long)(q0 + lr) * p.ldq + lg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const f16x8*)(qr + ks * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_k(0);
    issue_v(0);
    f32x4 o[FD];
#pragma unroll
    for (int fd = 0; fd < FD; ++fd) o[fd] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_run = 0.f;
    const int swz[2] = {(lg ^ (lr & 7)) << 4, ((4 + lg) ^ (lr & 7)) << 4};
    const float sc = p.scale * 1.44269504088896341f;   // scores in log2 units: exp(x) = exp2(x log2 e)

    for (int kb = 0; kb < nkb; ++kb) {
        // ---- S^T = K Q^T for this key block (the K tile's DMA - and every older request - has landed)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C / 64) : "memory");   // only this block's V^T pieces (the youngest C / 64 requests) may be in flight
        __builtin_amdgcn_s_barrier();
        f32x4 s[4];
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) s[fj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 a[4];
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) a[fj] = *(const f16x8*)(ks_ + (ks >> 1) * (BK * 128) + (16 * fj + lr) * 128 + swz[ks & 1]);
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) s[fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[fj], qf[ks], s[fj], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();            // every wave is done with the K tile
        if (kb + 1 < nkb) issue_k(kb + 1);       // ... the next one arrives during the softmax and the PV product
        // ---- online softmax of the 64 scores of query lr held by the four lane groups
        float mx = -3.0e38f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[fj][r] *= sc; mx = fmaxf(mx, s[fj][r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ls = 0.f;
        f16x8 pb[2];   // B operand of PV k-step ks: rows 32 ks + 8 lg + e  <->  S^T rows {16 (2 ks) + 4 lg + r, 16 (2 ks + 1) + 4 lg + r}
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e0 = __builtin_amdgcn_exp2f(s[2 * ks][r] - m_new), e1 = __builtin_amdgcn_exp2f(s[2 * ks + 1][r] - m_new);
                const f16 h0 = (f16)e0, h1 = (f16)e1;
                pb[ks][r] = h0; pb[ks][4 + r] = h1;
                ls += (float)h0 + (float)h1;   // the sum of what the PV product actually uses
            }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int fd = 0; fd < FD; ++fd) o[fd] = o[fd] * alpha;
        // ---- O^T += V^T P (the V^T tile has landed: only the next K tile's pieces may be in flight)
        if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int fd = 0; fd < FD; ++fd) {
                const f16x8 va = *(const f16x8*)(vs_ + (16 * fd + lr) * 128 + swz[ks]);
                o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[ks], o[fd], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();            // every wave is done with the V^T tile
        if (kb + 1 < nkb) issue_v(kb + 1);
    }
    // ---- O = O^T / l + bias: lane (lr, lg) holds channels 16 fd + 4 lg + r of query q0 + lr
    const float inv = 1.0f / l_run;
    f16* orow = p.o + z * (long long)T * p.ldo + (long long)(q0 + lr) * p.ldo + 4 * lg;
#pragma unroll
    for (int fd = 0; fd < FD; ++fd) {
        f32x4 bvv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bv) bvv = *(const f32x4*)(p.bv + 16 * fd + 4 * lg);
        f16x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (f16)fmaf(o[fd][r], inv, bvv[r]);
        *(f16x4*)(orow + 16 * fd) = hv;
    }
}

template <int C>
int launch_flash(const FlashParams& p, int nz, hipStream_t st) {
    constexpr size_t lds = (size_t)(C / 64) * 64 * 128 + (size_t)C * 128;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) { (void)hipFuncSetAttribute((const void*)ae_flash_attn_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
    hipLaunchKernelGGL((ae_flash_attn_kernel<C>), dim3(p.T / 128, nz), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

// fp16 streaming attention of the autoencoder's AttnBlock: C in {128, 256, 512} channels, T a multiple of 128, 16-byte aligned rows
extern "C" int rs_ae_flash_supported(int C, int T) { return (C == 128 || C == 256 || C == 512) && T >= 128 && (T % 128) == 0; }
extern "C" int rs_ae_flash_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, const float* bv, void* o, int ldo, int nz, int T, int C,
                                  float scale, hipStream_t st) {
    if (!rs_ae_flash_supported(C, T) || (ldq & 7) || (ldk & 7) || (ldo & 3)) return -2;
    if ((long long)T * ldk * 2 >= 0xF0000000LL || (long long)C * T * 2 >= 0xF0000000LL) return -2;   // 32-bit buffer offsets per image
    FlashParams p{};
    p.q = (const f16*)q; p.k = (const f16*)k; p.vt = (const f16*)vt; p.bv = bv; p.o = (f16*)o;
    p.T = T; p.ldq = ldq; p.ldk = ldk; p.ldo = ldo; p.scale = scale;
    if (C == 512) return launch_flash<512>(p, nz, st);
    if (C == 256) return launch_flash<256>(p, nz, st);
    return launch_flash<128>(p, nz, st);
}
