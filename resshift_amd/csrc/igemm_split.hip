// Implicit-GEMM kernel for SPLIT storage (RS_F16S, common.h): fp32-class convolutions / linears / batched GEMMs on the
// fp16 matrix cores.  Both operands arrive as (hi, lo) fp16 pairs, x = hi + lo * 2^-11, and every product is formed as
//     W.X = Wh.Xh  +  2^-11 (Wh.Xl + Wl.Xh)            (the dropped Wl.Xl term is <= 2^-24 |W.X|)
// with three v_mfma_f32_16x16x32_f16 per fragment pair into two fp32 accumulators (main, cross): 1/3 of the fp16 matrix
// rate, against 1/16 for the exact v_mfma_f32_16x16x4_f32 path - the precision policy that reproduces the reference's
// VQ codes (ldm/modules/vqvae/quantize.py:276-285) without paying for exact fp32 MFMAs.
//
// Same formulation, operand gather, swizzled LDS image and LDS-DMA ring as igemm2.hip (see there), with
//   * a K stage = 128 bytes of hi AND 128 bytes of lo per row: LDS slot = [X hi | W hi | X lo | W lo], two LDS-DMA
//     instructions per row round; pixel records in HBM are [ld halfs hi | ld halfs lo], weight rows [K hi | K lo];
//   * a 2-slot ring, one workgroup per CU (<= 160 KB of LDS, <= 256 VGPRs): 128 pixels x BC in {64,128,160,192} on 8
//     waves, or 64 pixels x BC on 4 waves for the launches that cannot fill the chip (split-K, 16x16 / 8x8 UNet levels);
//   * epilogue in exact fp32 arithmetic (IEEE division, libm erff): main + 2^-11 cross, scale, bias, activation,
//     residual (split storage), then either fp32 output or a fresh (hi, lo) split, transposed through LDS into 16-byte
//     stores.
#include "igemm_common.h"
#include "gn_tail.h"
#include <algorithm>
#include <type_traits>

extern "C" int rs_splitk_reduce_launch(const IGemmParams* p, int out_dt, hipStream_t st);
extern "C" int rs_splitk_reduce_stats_launch(const IGemmParams* p, int out_dt, hipStream_t st);   // igemm4.hip: reduce + statistics (+ GroupNorm tail)

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// NB: keep the LDS-DMA builtin inside a plain __device__ function (see igemm2.hip)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL (builds with -DRS_SPLIT_ABLATE only): timing ablations of the K loop, results wrong: 1 = no refill loads after the
// prologue, 2 = no ds_read / MFMA, 4 = no barrier (compile-time: runtime switches in the loop cost registers)
#ifdef RS_SPLIT_ABLATE
__device__ long long g_igs_clk[4 * 8192];   // per workgroup: cycle counter at kernel start / K loop start / K loop end / kernel end (ablate builds)
#define RS_IGS_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0) g_igs_clk[4 * blockIdx.x + (k)] = clock64()
#else
#define RS_IGS_STAMP(k)
#endif
template <typename TO, int BP, int BC, int NWV, bool PIPE, int ABL = 0>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 2 : 1) void igemm_split_kernel(IGemmParams p) {
    RS_IGS_STAMP(0);
    constexpr int WPN = NWV / 2;               // pixel-waves (x 2 channel-waves)
    constexpr int RND = 8 * NWV;               // rows covered by one LDS-DMA instruction of every wave
    constexpr int BK = 64;                     // halfs of K per stage (128 bytes of hi + 128 bytes of lo per row)
    constexpr int RWP = BC % RND;              // rows of the partial weight round (whole waves: a multiple of 8)
    static_assert(RWP % 8 == 0, "partial round must be whole waves");
    constexpr int RX = BP / RND, RW = (BC + RND - 1) / RND;
    constexpr int L = 2 * (RX + RW);           // LDS-DMA instructions per thread per stage (hi + lo)
    constexpr int FP = BP / WPN / 16;          // wave tile = (BP/WPN) pixels x (BC/2) channels
    constexpr int FC = BC / 32;
    constexpr int PLANE = (BP + BC) * 128;     // [X rows | W rows] of one half (hi or lo)
    constexpr int STAGE = 2 * PLANE;
    static_assert(BP % RND == 0 && (BP / WPN) % 16 == 0 && BC % 32 == 0 && 2 * STAGE <= 160 * 1024, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave % WPN, wc = wave / WPN;
    const int rr = 8 * wave + (lane >> 3);             // row inside a load round
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);    // source K-chunk of this lane (swizzle on the source side)
    const bool wpart = !RWP || wave < RWP / 8;         // this wave owns rows of the last (partial) weight round

    const int nby = (p.Cout + BC - 1) / BC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nby) * BP;
    const int n0 = (tile % nby) * BC;
    const long long z = blockIdx.z;
    const bool split = p.splitk > 1;

    constexpr unsigned INV = 0xF0000000u;   // beyond num_records: the hardware returns zeros
    const f16* x0 = (const f16*)p.x0 + (split ? 0 : 2 * z * p.bs_x0);
    const f16* w = (const f16*)p.w + (split ? 0 : 2 * z * p.bs_w);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x0, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, p.w_bytes, 0x00020000);

    const int Ctot = p.C0;
    const int ntaps = p.KH * p.KW;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const int HoWo = p.Ho * p.Wo;
    const unsigned xlo = (unsigned)p.ld0 * 2u;     // byte distance hi -> lo inside a pixel record
    const unsigned wlo = (unsigned)p.Ktot * 2u;    // ... inside a weight row

    int pixbase[RX], iy0[RX], ix0[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int m = m0 + RND * i + rr;
        if (m < p.M) {
            int b, rem, oy, ox;
            if (p.sh_wo >= 0) {
                b = m >> p.sh_howo; rem = m & (HoWo - 1); oy = rem >> p.sh_wo; ox = rem & (p.Wo - 1);
            } else {
                b = m / HoWo; rem = m - b * HoWo; oy = rem / p.Wo; ox = rem - oy * p.Wo;
            }
            pixbase[i] = b * p.Hs * p.Ws;
            iy0[i] = oy * p.stride - p.pad_t;
            ix0[i] = ox * p.stride - p.pad_l;
        } else {
            pixbase[i] = -1; iy0[i] = 0; ix0[i] = 0;
        }
    }
    unsigned woff[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int n = n0 + RND * i + rr;
        woff[i] = (RND * i + rr < BC && n < p.Cout) ? (unsigned)n * (unsigned)p.Ktot * 4u : INV;
    }
    const int nk_total = (p.Ktot + BK - 1) / BK;
    int kt0 = 0, nk = nk_total;
    if (split) {
        const int per = (nk_total + p.splitk - 1) / p.splitk;
        kt0 = min(nk_total, (int)z * per);
        nk = min(nk_total, kt0 + per) - kt0;
    }
    int kk = kt0 * BK + kcp * 8;
    int tap = kk / Ctot;
    int cc = kk - tap * Ctot;

    unsigned off[RX];
    int ky = tap / p.KW, kx = tap - ky * p.KW;
    auto set_tap = [&]() {
        const bool kvalid = tap < ntaps;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kvalid && pixbase[i] >= 0 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            const unsigned pix = (unsigned)(pixbase[i] + (iy >> ush) * p.Ws + (ix >> ush));
            off[i] = ok ? pix * (unsigned)p.ld0 * 4u : INV;
        }
    };
    set_tap();

    auto issue = [&](int slot) {
        char* sbase = smem + slot * STAGE + (8 * wave) * 128;   // wave-uniform
        const unsigned cb = (unsigned)cc * 2u;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            lds_dma16(rx, sbase + (RND * i) * 128, off[i] + cb);
            lds_dma16(rx, sbase + PLANE + (RND * i) * 128, off[i] == INV ? INV : off[i] + cb + xlo);
        }
        const bool wk = kk < p.Ktot;
        const unsigned kb = (unsigned)kk * 2u;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (RWP && i == RW - 1 && !wpart) continue;   // wave-uniform: this wave's rows of the last round are >= BC
            const bool ok = wk && woff[i] != INV;
            lds_dma16(rw, sbase + (BP + RND * i) * 128, ok ? woff[i] + kb : INV);
            lds_dma16(rw, sbase + PLANE + (BP + RND * i) * 128, ok ? woff[i] + kb + wlo : INV);
        }
        kk += BK;
        cc += BK;
        if (cc >= Ctot) {
            do {
                cc -= Ctot; ++tap;
                if (++kx == p.KW) { kx = 0; ++ky; }
            } while (cc >= Ctot);
            set_tap();
        }
    };

    const int swz0 = ((lg ^ (lr & 7)) << 4), swz1 = (((4 + lg) ^ (lr & 7)) << 4);
    const int la = BP * 128 + (wc * (BC / 2) + lr) * 128, lb = (wp * (BP / WPN) + lr) * 128;

    f32x4 am[FC][FP], ac[FC][FP];   // main (hi.hi) and cross (hi.lo + lo.hi, scaled by 2^11) accumulators
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) { am[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    RS_IGS_STAMP(1);
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    if constexpr (!PIPE) {
        // reference schedule (round 2's A/B partner, not instantiated any more): the whole refill is issued right behind the barrier
        for (int kt = 0; kt < nk; ++kt) {
            if (kt == 0 && nk > 1) {   // stage 1 was issued with stage 0; its loads (two fewer on the waves that skip the partial round) may fly on
                if (RWP && !wpart) wait_vmcnt<L - 2>(); else wait_vmcnt<L>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (kt >= 1 && kt + 1 < nk) issue((kt + 1) & 1);   // refill the slot every wave finished reading before this barrier
            const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = ks ? swz1 : swz0;
                f16x8 ah[FC], al[FC], bh[FP], bl[FP];
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    ah[i] = *(const f16x8*)(sb + la + sw + i * 2048);
                    al[i] = *(const f16x8*)(sb + PLANE + la + sw + i * 2048);
                }
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    bh[j] = *(const f16x8*)(sb + lb + sw + j * 2048);
                    bl[j] = *(const f16x8*)(sb + PLANE + lb + sw + j * 2048);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j) am[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], am[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], ac[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], ac[i][j], 0, 0, 0);
            }
        }
    } else {
        // Pipelined schedule.  A stage is 2 * FC blocks of 3 * FP MFMAs (one channel fragment of one k-step each).  The L
        // LDS-DMA instructions of the refill are NOT issued in one burst behind the barrier - every wave would sit in the
        // texture-address queue for ~L KB / (52 B/clk) while the matrix pipe idles - but one or two per block, in front of
        // that block's MFMAs, over the first DBLK blocks (the rest of the stage gives the last pieces time to land before
        // the next barrier's vmcnt(0)).  The fragments of k-step 1 are read into a second register set while k-step 0
        // computes.  sched_barrier(0) pins the block order; inside a block the compiler schedules freely.
        constexpr int NBLK = 2 * FC;
        constexpr int DBLK = NBLK - (NBLK >= 8 ? 3 : 1);            // blocks that carry DMA pieces
        constexpr bool DBUF = (FC + FP) * 16 + FC * FP * 8 <= 200;  // second fragment set only where it fits 256 VGPRs
        // piece d of the refill: d < 2 RX -> pixel rows (round d/2, half d&1), then weight rows
        auto piece = [&](char* sbase, unsigned cb, unsigned kb, bool wk, int d) {
            if (d < 2 * RX) {
                const int i = d >> 1, half = d & 1;
                lds_dma16(rx, sbase + half * PLANE + (RND * i) * 128, off[i] == INV ? INV : off[i] + cb + (half ? xlo : 0u));
            } else {
                const int e = d - 2 * RX, i = e >> 1, half = e & 1;
                if (RWP && i == RW - 1 && !wpart) return;
                const bool ok = wk && woff[i] != INV;
                lds_dma16(rw, sbase + half * PLANE + (BP + RND * i) * 128, ok ? woff[i] + kb + (half ? wlo : 0u) : INV);
            }
        };
        auto stage = [&](int kt, auto refill_tag, auto compute_tag) {
            constexpr bool REFILL = decltype(refill_tag)::value;
            constexpr bool COMPUTE = decltype(compute_tag)::value;   // false: timing ablation (RS_IGEMM_DBG & 2), results wrong
            const char* sb = smem + (kt & 1) * STAGE;
            char* sbase = smem + ((kt + 1) & 1) * STAGE + (8 * wave) * 128;
            const unsigned cb = (unsigned)cc * 2u, kb = (unsigned)kk * 2u;
            const bool wk = kk < p.Ktot;
            f16x8 ah[2][FC], al[2][FC], bh[2][FP], bl[2][FP];
            auto read_set = [&](int ks, int set) {
                const int sw = ks ? swz1 : swz0;
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    bh[set][j] = *(const f16x8*)(sb + lb + sw + j * 2048);
                    bl[set][j] = *(const f16x8*)(sb + PLANE + lb + sw + j * 2048);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    ah[set][i] = *(const f16x8*)(sb + la + sw + i * 2048);
                    al[set][i] = *(const f16x8*)(sb + PLANE + la + sw + i * 2048);
                }
            };
            if (COMPUTE) read_set(0, 0);
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const int ks = blk / FC, i = blk % FC;
                const int set = DBUF ? ks : 0;
                if (REFILL) {
#pragma unroll
                    for (int d = 0; d < L; ++d)
                        if (d * DBLK / L == blk) piece(sbase, cb, kb, wk, d);
                }
                if (COMPUTE) {
                    if (DBUF ? blk == 1 : blk == FC) read_set(1, DBUF ? 1 : 0);
#pragma unroll
                    for (int j = 0; j < FP; ++j) am[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[set][i], bh[set][j], am[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FP; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[set][i], bl[set][j], ac[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FP; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[set][i], bh[set][j], ac[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (REFILL) {   // step this lane's K cursor past the stage just requested
                kk += BK;
                cc += BK;
                if (cc >= Ctot) {
                    do {
                        cc -= Ctot; ++tap;
                        if (++kx == p.KW) { kx = 0; ++ky; }
                    } while (cc >= Ctot);
                    set_tap();
                }
            }
        };
        if (nk > 0) {
            if (nk > 1) { if (RWP && !wpart) wait_vmcnt<L - 2>(); else wait_vmcnt<L>(); } else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            stage(0, std::false_type{}, std::true_type{});
        }
        for (int kt = 1; kt + 1 < nk; ++kt) {
            wait_vmcnt<0>();
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            stage(kt, std::integral_constant<bool, !(ABL & 1)>{}, std::integral_constant<bool, !(ABL & 2)>{});
        }
        if (nk > 1) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            stage(nk - 1, std::false_type{}, std::true_type{});
        }
    }
    RS_IGS_STAMP(2);
    __syncthreads();  // all waves done with the ring: the epilogue reuses it as staging space

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) am[i][j][r] = fmaf(ac[i][j][r], RS_LO_INV, am[i][j][r]);

    if (split) {
        float* part = p.partial + z * (long long)p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / WPN) + j * 16 + lr;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
                if (n >= p.Cout) continue;
                float* pp = part + (long long)m * p.Cout + n;
                if (n + 3 < p.Cout && (p.Cout & 3) == 0) *(f32x4*)pp = am[i][j];
                else for (int r = 0; r < 4 && n + r < p.Cout; ++r) pp[r] = am[i][j][r];
            }
        }
        RS_IGS_STAMP(3);
        return;
    }
    const f16* res = p.res ? (const f16*)p.res + 2 * z * p.bs_res : nullptr;   // residual: split storage, pixel stride ldres
    const bool quad = (p.Cout & 3) == 0 && (p.ldres & 3) == 0;
    // The epilogue is VALU / memory work that no MFMA overlaps (one workgroup per CU), so it is kept lean like igemm2's: the bias of
    // every channel fragment is fetched once up front, the residual of channel fragment i+1 is in flight while fragment i is
    // finished (one L2 round trip per channel fragment instead of one per 16 x 16 fragment: 24 in a row for a 128 x 192 tile
    // were most of a short-K launch), and the activation is a compile-time choice behind one uniform branch.
    const bool res_fast = res && quad;
    f32x4 bvs[FC];
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
        bvs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            if ((p.Cout & 3) == 0) bvs[i] = *(const f32x4*)(p.bias + min(n, p.Cout - 4));
            else for (int r = 0; r < 4; ++r) bvs[i][r] = n + r < p.Cout ? p.bias[n + r] : 0.f;
        }
    }
    long long mres[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) mres[j] = (long long)min(m0 + wp * (BP / WPN) + j * 16 + lr, p.M - 1) * p.ldres * 2;
    // (the whole wave tile's residual at once - 48 registers - was measured too: no further gain, the remaining epilogue time of
    // the short-K launches is the latency of the first residual / bias round trip itself and the two staging passes)
    constexpr int RSLOT = 2;
    f16x4 rh[RSLOT][FP], rl[RSLOT][FP];
    auto load_res = [&](int i) __attribute__((always_inline)) {   // (clamped, branch-free: rows / channels outside the output are never stored)
        const int nr = min(n0 + wc * (BC / 2) + i * 16 + lg * 4, p.Cout - 4);
#pragma unroll
        for (int j = 0; j < FP; ++j) { rh[i % RSLOT][j] = *(const f16x4*)(res + mres[j] + nr); rl[i % RSLOT][j] = *(const f16x4*)(res + mres[j] + p.ldres + nr); }
    };
    // scale, bias, activation, residual of channel fragment row i, in place
    auto finish_row = [&](int i, auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            f32x4 v = am[i][j] * p.out_scale + bvs[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ACT == RS_ACT_GELU ? rs_gelu(v[r]) : (ACT == RS_ACT_SILU ? rs_silu(v[r]) : v[r]);
            if (res_fast) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += rs_join(rh[i % RSLOT][j][r], rl[i % RSLOT][j][r]);
            } else if (res) {
                const int m = m0 + wp * (BP / WPN) + j * 16 + lr;
                if (m < p.M)
                    for (int r = 0; r < 4 && n + r < p.Cout; ++r) v[r] += rs_join(res[(long long)m * p.ldres * 2 + n + r], res[(long long)m * p.ldres * 2 + p.ldres + n + r]);
            }
            am[i][j] = v;
        }
    };
    auto finish_all = [&](auto act_tag) __attribute__((always_inline)) {
        if (res_fast) load_res(0);
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            if (res_fast && i + 1 < FC) load_res(i + 1);
            finish_row(i, act_tag);
        }
    };
    if (p.act == RS_ACT_GELU) finish_all(std::integral_constant<int, RS_ACT_GELU>{});
    else if (p.act == RS_ACT_SILU) finish_all(std::integral_constant<int, RS_ACT_SILU>{});
    else finish_all(std::integral_constant<int, RS_ACT_NONE>{});
    bool tail_last = false;
    int tail_img = 0;
    if constexpr (std::is_same<TO, h2s>::value) {
        // Per-channel statistics of the output for the GroupNorm that consumes it (IGemmParams::ystats, [B][HW / BP][ystats_ld][2]; the
        // launcher guarantees whole tiles inside one image: M % BP == 0, HW % BP == 0): the stored (hi, lo) pair reproduces the value to
        // 2^-23, so the sums are taken from the registers - as in the halo kernel's epilogue (igemm4_kernel.h), same reduction: the 16
        // pixel lanes by xor-shuffles, the pixel-waves through LDS in a fixed order, no atomics.  With IGemmParams::tail the workgroup
        // that completes an image's statistics also writes the consuming GroupNorm's coefficients (gn_tail.h).
        if (p.ystats) {
            constexpr int WT = (BP / WPN) * ((BC / 2) * 2 + 16);
            float* const sb = (float*)(smem + NWV * 2 * WT);         // behind the staging tiles: [NWV][BC / 2][2]
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int j = 0; j < FP; ++j) { a += am[i][j][r]; q = fmaf(am[i][j][r], am[i][j][r], q); }
                    a = rs_sum16(a); q = rs_sum16(q);
                    if (lr == 0) { sb[(wave * (BC / 2) + i * 16 + lg * 4 + r) * 2] = a; sb[(wave * (BC / 2) + i * 16 + lg * 4 + r) * 2 + 1] = q; }
                }
            __syncthreads();
            const int hw = p.Ho * p.Wo;
            tail_img = m0 / hw;
            if (wave == 0) {   // (the other waves go on to stage and store the tile; the tail's round trip to memory is wave 0's alone: gn_tail.h)
                for (int c = lane; c < BC; c += 64) {
                    if (n0 + c >= p.Cout) break;
                    const int hw_ = c / (BC / 2), cl = c - hw_ * (BC / 2);   // channel-wave, channel inside its half
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int w4 = 0; w4 < WPN; ++w4) { a += sb[((hw_ * WPN + w4) * (BC / 2) + cl) * 2]; q += sb[((hw_ * WPN + w4) * (BC / 2) + cl) * 2 + 1]; }
                    // (osc == 2: this launch is parity class 2 ooy + oox of four; an image's slabs are the four classes' tiles one after the other)
                    const long long slab = p.osc == 2 ? ((long long)tail_img * 4 + 2 * p.ooy + p.oox) * (hw / BP) + (m0 - tail_img * hw) / BP
                                                      : (long long)tail_img * (hw / BP) + (m0 - tail_img * hw) / BP;
                    float* dst = p.ystats + (slab * p.ystats_ld + n0 + c) * 2;
                    if (p.tail.coef) rs_pub_pair(dst, a, q);
                    else { dst[0] = a; dst[1] = q; }
                }
                if (p.tail.coef) { const bool last = rs_gn_tail_arrive(p.tail, tail_img); if (lane == 0) *(unsigned*)(sb + NWV * (BC / 2) * 2) = last ? 1u : 0u; }
            }
            tail_last = p.tail.coef != nullptr;   // (here: "a tail is attached"; the flag word decides behind the end-of-kernel barrier)
        }
        f16* y = (f16*)p.y + 2 * z * p.bs_y;
        // wave tile (BP/WPN rows x BC/2 channels) staged twice (hi, lo) with a padded row pitch, then 16-byte stores
        constexpr int ROWB = (BC / 2) * 2 + 16;
        constexpr int WTILE = (BP / WPN) * ROWB;
        char* stg = smem + wave * 2 * WTILE;
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                f16x4 h, l;
#pragma unroll
                for (int r = 0; r < 4; ++r) { f16 a, b; rs_split(am[i][j][r], a, b); h[r] = a; l[r] = b; }
                *(f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = h;
                *(f16x4*)(stg + WTILE + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = l;
            }
        RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
        constexpr int CPR = (BC / 2) / 8;
        constexpr int NITEM = (BP / WPN) * CPR;
        const bool vec_ok = (p.ldy & 7) == 0;
        for (int idx = lane; idx < 2 * NITEM; idx += 64) {
            const int half = idx >= NITEM ? 1 : 0;
            const int it = idx - half * NITEM;
            const int row = it / CPR, c8 = it - row * CPR;
            const int m = m0 + wp * (BP / WPN) + row;
            const int n = n0 + wc * (BC / 2) + c8 * 8;
            if (m >= p.M || n >= p.Cout) continue;
            const uint4 v = *(const uint4*)(stg + half * WTILE + row * ROWB + c8 * 16);
            f16* yp = y + rs_out_m(p, m) * p.ldy * 2 + half * p.ldy + n;
            if (vec_ok && n + 7 < p.Cout) {
                *(uint4*)yp = v;
            } else {
                const f16x8 hv = __builtin_bit_cast(f16x8, v);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (n + r < p.Cout) yp[r] = hv[r];
            }
        }
    } else {
        float* y = (float*)p.y + z * p.bs_y;
        const bool vec_ok = ((p.ldy & 3) == 0);
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / WPN) + j * 16 + lr;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
                if (m >= p.M || n >= p.Cout) continue;
                float* yp = y + rs_out_m(p, m) * p.ldy + n;
                if (n + 3 < p.Cout && vec_ok) *(f32x4*)yp = am[i][j];
                else for (int r = 0; r < 4 && n + r < p.Cout; ++r) yp[r] = am[i][j][r];
            }
        }
    }
    if (tail_last) {   // (kernel-uniform) every wave is done with its staging tile behind this barrier: the image's last workgroup turns to the coefficients
        constexpr int WT2 = (BP / WPN) * ((BC / 2) * 2 + 16);
        __syncthreads();
        if (*(const unsigned*)(smem + NWV * 2 * WT2 + NWV * (BC / 2) * 2 * sizeof(float))) rs_gn_tail_finish<64 * NWV>(p.tail, tail_img, (float*)smem);
    }
    RS_IGS_STAMP(3);
}

template <typename TO, int BP, int BC, int NWV, bool PIPE, int ABL = 0>
hipError_t launch_cfg2(IGemmParams p, int nz, hipStream_t st) {
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    constexpr size_t lds = (size_t)2 * 2 * (BP + BC) * 128;
    static_assert(lds <= 160 * 1024, "LDS");
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)igemm_split_kernel<TO, BP, BC, NWV, PIPE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * 4, wb = (size_t)p.Cout * p.Ktot * 4;
    if (xb >= 0xF0000000ull || wb >= 0xF0000000ull) return hipErrorInvalidValue;  // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    {
        const int howo = p.Ho * p.Wo;
        const bool pow2 = howo > 0 && (howo & (howo - 1)) == 0 && (p.Wo & (p.Wo - 1)) == 0;
        p.sh_howo = pow2 ? __builtin_ctz(howo) : -1;
        p.sh_wo = pow2 ? __builtin_ctz(p.Wo) : -1;
    }
    // output statistics (+ GroupNorm tail): from this kernel's epilogue, or - split-K - from the reduce kernel that finishes the slabs
    float* const want_stats = p.ystats;
    const GNTail want_tail = p.tail;
    if (want_tail.coef && !want_stats) return hipErrorInvalidValue;
    if (want_stats) {
        const int hw = p.Ho * p.Wo;
        if (!std::is_same<TO, h2s>::value || nz != 1) return hipErrorInvalidValue;
        if (p.splitk > 1) { p.ystats = nullptr; p.tail.coef = nullptr; }
        else {
            if ((p.M % BP) || (hw % BP)) return hipErrorInvalidValue;   // whole tiles inside one image (rs_igemm_split_stats_px)
            if (p.tail.coef) {
                if (p.tail.C > 2048 || p.tail.groups < 1 || p.tail.groups > 64 || (p.tail.C % p.tail.groups)) return hipErrorInvalidValue;   // (the finish's LDS scratch: 2 C + 2 groups floats)
                const int classes = p.osc == 2 ? 4 : 1;   // (scattered launches: the four parity classes share one ticket and one slab array)
                p.tail.expected = classes * (hw / BP) * ((p.Cout + BC - 1) / BC);
                p.tail.st0 = p.ystats; p.tail.S0 = classes * (hw / BP); p.tail.ld0 = p.ystats_ld; p.tail.n0 = p.Cout;
            }
        }
    }
    hipLaunchKernelGGL((igemm_split_kernel<TO, BP, BC, NWV, PIPE, ABL>), dim3(tiles, 1, p.splitk > 1 ? p.splitk : nz), dim3(64 * NWV), lds, st, p);
    if (p.splitk > 1) {
        p.ystats = want_stats; p.tail = want_tail;
        const int out_dt = std::is_same<TO, h2s>::value ? RS_F16S : RS_F32;
        if ((want_stats ? rs_splitk_reduce_stats_launch(&p, out_dt, st) : rs_splitk_reduce_launch(&p, out_dt, st)) != 0) return hipErrorLaunchFailure;
    }
    return hipGetLastError();
}

template <typename TO, int BP, int BC, int NWV>
hipError_t launch_cfg(const IGemmParams& p, int nz, hipStream_t st) {
#ifdef RS_SPLIT_ABLATE
    if constexpr (std::is_same<TO, h2s>::value && NWV == 8) {
        static const int abl = []() { const char* e = getenv("RS_IGEMM_DBG"); return e ? atoi(e) : 0; }();
        switch (abl) {
            case 1: return launch_cfg2<TO, BP, BC, NWV, true, 1>(p, nz, st);
            case 2: return launch_cfg2<TO, BP, BC, NWV, true, 2>(p, nz, st);
            case 4: return launch_cfg2<TO, BP, BC, NWV, true, 4>(p, nz, st);
            case 6: return launch_cfg2<TO, BP, BC, NWV, true, 6>(p, nz, st);
            default: break;
        }
    }
#endif
    return launch_cfg2<TO, BP, BC, NWV, true>(p, nz, st);   // (the un-pipelined reference schedule, PIPE = false, is no longer instantiated)
}

template <typename TO>
hipError_t launch_t(const IGemmParams& p, int BP, int BC, int nz, hipStream_t st) {
    if (BP == 64) {
        switch (BC) {
            case 64: return launch_cfg<TO, 64, 64, 4>(p, nz, st);
            case 160: return launch_cfg<TO, 64, 160, 4>(p, nz, st);
            case 192: return launch_cfg<TO, 64, 192, 4>(p, nz, st);
            default: return launch_cfg<TO, 64, 128, 4>(p, nz, st);
        }
    }
    switch (BC) {
        case 64: return launch_cfg<TO, 128, 64, 8>(p, nz, st);
        case 160: return launch_cfg<TO, 128, 160, 8>(p, nz, st);
        case 192: return launch_cfg<TO, 128, 192, 8>(p, nz, st);
        default: return launch_cfg<TO, 128, 128, 8>(p, nz, st);
    }
}

}  // namespace

// tile choice: channel tile with the least padding (64 for the 3-channel heads), 64-pixel tiles on 4 waves where 128-pixel
// tiles would leave most CUs idle (the 16x16 / 8x8 UNet levels at batch 32: M <= 8192)
extern "C" void rs_igemm_split_pick(int M, int Cout, int nz, int* BP, int* BC) {
    auto waste = [&](int bc) { return ((Cout + bc - 1) / bc) * bc - Cout; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    if (Cout <= 64) best = 64;
    *BC = best;
    const long long tiles128 = (long long)((M + 127) / 128) * ((Cout + best - 1) / best) * nz;
    *BP = (tiles128 < 256) ? 64 : 128;
    // Down to 64 tiles the 128-pixel tile stays on 8 waves and the split-K planner fills the chip with K slices instead (M = 8192, N = 320:
    // 128 tiles x 2 slices of 8 waves - two waves per SIMD on every CU, half the K loop per workgroup, + the reduce kernel - against 256
    // tiles of 4 waves, one wave per SIMD: 68 us for 18 us of MFMA work).  Measured in round 5 (profiles/r5_fold_ab.txt): igemm_split family
    // 51.9 -> 49.0 ms per parity pass, the pass - 2.1 ms, + 210 reduce launches.  RS_SPLIT_BP128_SK=0: the round-4 choice.
    static const int bp128_min = []() { const char* e = getenv("RS_SPLIT_BP128_SK"); return e ? atoi(e) : 64; }();
    if (bp128_min > 0 && nz == 1 && tiles128 >= bp128_min) *BP = 128;
}

// pixels per statistics slab of an igemm_split launch that is asked for IGemmParams::ystats (0: it cannot produce them): the pixel tile,
// or - split-K launches, whose reduce kernel produces them - 256 consecutive pixels / the whole small image (as rs_igemm4_stats_px)
extern "C" int rs_igemm_split_stats_px(const IGemmParams* pp, int splitk) {
    const IGemmParams& p = *pp;
    if (p.C1 != 0 || (p.Cout & 3) || (p.ldy & 3)) return 0;
    const int hw = p.Ho * p.Wo;
    if (splitk > 1) return (hw > 256 && (hw % 256)) ? 0 : std::min(hw, 256);
    int BP, BC;
    rs_igemm_split_pick(p.M, p.Cout, 1, &BP, &BC);
    return ((p.M % BP) || (hw % BP)) ? 0 : BP;
}

// in: split storage; out_dt: RS_F16S or RS_F32.  Single source only (C1 == 0), C0 / ld0 / Ktot multiples of 8.
#ifdef RS_SPLIT_ABLATE
// ablate builds: mean cycles of the three kernel phases over the first `nwg` workgroups of the last igemm_split launch
extern "C" int rs_igemm_split_phase_cycles(int nwg, double* out3) {
    static long long h[4 * 8192];
    if (nwg < 1 || nwg > 8192) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_igs_clk), sizeof(long long) * 4 * nwg) != hipSuccess) return -1;
    out3[0] = out3[1] = out3[2] = 0.0;
    for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 3; ++k) out3[k] += (double)(h[4 * i + k + 1] - h[4 * i + k]) / nwg;
    return 0;
}
#endif

extern "C" int rs_igemm_split_launch(const IGemmParams* pp, int out_dt, int nz, hipStream_t st) {
    const IGemmParams& p = *pp;
    if (p.C1 != 0 || (p.C0 % 8) || (p.ld0 % 8) || (p.Ktot % 8)) return -2;
    int BP, BC;
    rs_igemm_split_pick(p.M, p.Cout, nz, &BP, &BC);   // (same arguments as the split-K planner: identical tile choice)
    hipError_t e;
    if (out_dt == RS_F16S) e = launch_t<h2s>(p, BP, BC, nz, st);
    else if (out_dt == RS_F32) e = launch_t<float>(p, BP, BC, nz, st);
    else return -2;
    return e == hipSuccess ? 0 : -1;
}
