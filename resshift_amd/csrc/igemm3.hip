// Implicit-GEMM kernel, third generation (fp16 storage, long-K layers that fill the chip): same formulation, operand
// gather, LDS image and epilogue as igemm2.hip, but sized and scheduled for ONE workgroup per CU:
//   * tile 256 pixels x BC channels (BC = 128 / 160), 8 waves as 4 pixel-waves x 2 channel-waves, wave tile 64 x BC/2.
//     Against the 128-pixel tile this moves 25-35 % fewer bytes per FLOP over the CU's L1->LDS path (LDS-DMA) and reads
//     40 % fewer LDS bytes per MFMA (profiles/r1_igemm_ablation.txt: both were within 20 % of the MFMA time per stage);
//   * 3-slot LDS ring (3 x (256+BC) x 128 B <= 160 KB) and TWO fragment register sets per wave: the ds_reads of the next
//     k-step are always issued before the MFMAs of the current one, across the stage boundary as well, so the MFMA
//     pipe never waits on LDS latency with only two waves per SIMD;
//   * one `s_waitcnt vmcnt(0)` + raw `s_barrier` per K stage (40-64 MFMAs per wave), placed where the next MFMA group
//     already has its operands in registers.  Loop body after barrier(i):
//         X1: LDS-DMA of stage i+2 (slot freed by stage i-1)  |  ds_read k-step 0 of stage i+1  |  MFMA k-step 1 of stage i
//         X2: ds_read k-step 1 of stage i+1                     |  MFMA k-step 0 of stage i+1
//     barrier(i+1) then publishes stage i+2 and retires every read of stage i.
#include "igemm_common.h"
#include <type_traits>

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// NB: keep the LDS-DMA builtin inside a plain __device__ function (see igemm2.hip)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename TO, int BC>
__global__ __launch_bounds__(512, 2) void igemm3_kernel(IGemmParams p) {
    typedef f16 TI;
    constexpr int BP = 256, WPN = 4;
    constexpr int BK = 64;                      // halfs of K per stage (128 bytes per LDS row)
    constexpr int RX = BP / 64;                 // LDS-DMA rounds (64 rows each: 8 waves x 8 rows) of pixel rows
    constexpr int RWF = BC / 64;                // full rounds of weight rows
    constexpr int RWP = BC % 64;                // rows of the last, partial round (waves 0 .. RWP/8-1 take part)
    constexpr int RW = RWF + (RWP ? 1 : 0);
    constexpr int FP = BP / WPN / 16;           // 4
    constexpr int FC = BC / 32;
    constexpr int STAGE = (BP + BC) * 128;
    static_assert(RWP % 8 == 0 && BC % 32 == 0 && 3 * STAGE <= 160 * 1024, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave % WPN, wc = wave / WPN;
    const int rr = 8 * wave + (lane >> 3);             // row inside a 64-row load round
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);    // source K-chunk of this lane (swizzle on the source side)
    const bool wpart = wave < RWP / 8;                 // this wave owns rows of the partial weight round

    const int nby = (p.Cout + BC - 1) / BC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nby) * BP;
    const int n0 = (tile % nby) * BC;

    constexpr unsigned INV = 0xF0000000u;   // beyond num_records: the hardware returns zeros
    constexpr unsigned SZ = sizeof(TI);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    const int Ctot = p.C0;
    const int ntaps = p.KH * p.KW;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const int HoWo = p.Ho * p.Wo;

    int pixbase[RX], iy0[RX], ix0[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int m = m0 + 64 * i + rr;
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
        const int n = n0 + 64 * i + rr;
        woff[i] = (64 * i + rr < BC && n < p.Cout) ? (unsigned)n * (unsigned)p.Ktot * SZ : INV;
    }
    const int nk = (p.Ktot + BK - 1) / BK;
    int kk = kcp * 8;
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
            off[i] = ok ? pix * (unsigned)p.ld0 * SZ : INV;
        }
    };
    set_tap();

    // issue the LDS-DMA loads of one K stage into the ring slot at byte offset `slot`, then step this lane's K cursor
    auto issue = [&](int slot) {
        char* sbase = smem + slot + (8 * wave) * 128;   // wave-uniform
        const unsigned cb = (unsigned)cc * SZ;
#pragma unroll
        for (int i = 0; i < RX; ++i) lds_dma16(rx, sbase + (64 * i) * 128, off[i] + cb);
        const bool wk = kk < p.Ktot;
        const unsigned kb = (unsigned)kk * SZ;
#pragma unroll
        for (int i = 0; i < RWF; ++i) lds_dma16(rw, sbase + (BP + 64 * i) * 128, wk ? woff[i] + kb : INV);
        if (RWP && wpart) lds_dma16(rw, sbase + (BP + 64 * RWF) * 128, wk ? woff[RW - 1] + kb : INV);
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

    // per-lane LDS fragment bases (relative to a ring slot): weight rows wc*(BC/2)+lr, pixel rows wp*64+lr; k-step 0 / 1
    const int swz0 = ((lg ^ (lr & 7)) << 4), swz1 = (((4 + lg) ^ (lr & 7)) << 4);
    const int la = BP * 128 + (wc * (BC / 2) + lr) * 128, lb = (wp * (BP / WPN) + lr) * 128;

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment register sets of k-step 0 / 1, carried across loop iterations as 4 x i32 (a loop-carried <8 x half> is
    // legalised element-wise by the compiler: v_perm / v_lshr repacking between the MFMAs)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 a0[FC], b0[FP], a1[FC], b1[FP];
    auto rd = [&](int slot, int swz, i32x4 (&a)[FC], i32x4 (&b)[FP]) {
        const char* pa = smem + slot + la + swz;
        const char* pb = smem + slot + lb + swz;
#pragma unroll
        for (int i = 0; i < FC; ++i) a[i] = *(const i32x4*)(pa + i * 2048);
#pragma unroll
        for (int j = 0; j < FP; ++j) b[j] = *(const i32x4*)(pb + j * 2048);
    };
    auto mm = [&](const i32x4 (&a)[FC], const i32x4 (&b)[FP]) {
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
    };

    // Period i (one barrier each), with stage i+1 already in LDS and stage i+2 in flight when it starts:
    //     ds_read k-step 1 of stage i -> set 1 | MFMA k-step 0 of stage i (set 0)
    //     lgkmcnt(0): slot i is now entirely in registers;  vmcnt(Lw): this wave's loads of stage i+1 have landed
    //     s_barrier: stage i+1 visible to everyone, slot i free for everyone
    //     LDS-DMA of stage i+3 -> slot i  |  ds_read k-step 0 of stage i+1 -> set 0  |  MFMA k-step 1 of stage i (set 1)
    // so every stage is in flight for two full periods (80-128 MFMAs per wave) before anybody waits for it, and both MFMA
    // groups start with their operands already in registers.
    int s0 = 0, s1 = STAGE, s2 = 2 * STAGE;   // ring slots of stage i, i+1, i+2
    issue(s0);
    if (nk > 1) issue(s1);
    if (nk > 2) issue(s2);
    auto wait_keep = [&](int later) {   // wait until at most `later` (0..2) younger stages of this wave's loads are outstanding
        if (later >= 2) { if (RWP && wpart) wait_vmcnt<2 * (RX + RW)>(); else wait_vmcnt<2 * (RX + RWF)>(); }
        else if (later == 1) { if (RWP && wpart) wait_vmcnt<RX + RW>(); else wait_vmcnt<RX + RWF>(); }
        else wait_vmcnt<0>();
    };
    wait_keep(min(2, nk - 1));
    __builtin_amdgcn_s_barrier();
    rd(s0, swz0, a0, b0);
    for (int i = 0; i < nk; ++i) {
        if (!(p.dbg & 2)) {
            rd(s0, swz1, a1, b1);
            mm(a0, b0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_keep(min(1, nk - 2 - i));         // stage i+1 landed; stage i+2 may still be in flight
        if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();
        if (i + 3 < nk && !(p.dbg & 1)) issue(s0);
        if (!(p.dbg & 2)) {
            if (i + 1 < nk) rd(s1, swz0, a0, b0);
            mm(a1, b1);
        }
        const int t = s0; s0 = s1; s1 = s2; s2 = t;
    }
    __syncthreads();  // all waves done with the ring: the epilogue reuses it as staging space

    // ---------------------------------------------------------------- epilogue (as in igemm2.hip)
    TO* y = (TO*)p.y;
    const TO* res = (const TO*)p.res;
    const bool res_vec = res && (p.ldres & 3) == 0;
    const bool quad = (p.Cout & 3) == 0;
    if constexpr (sizeof(TO) == 2) {
        constexpr int ROWB = (BC / 2) * 2 + 16;
        char* stg = smem + wave * (BP / WPN) * ROWB;
        const bool res_fast = res && res_vec && quad;
        // residual: clamped (branch-free) vector loads, all issued before the arithmetic (the fragment registers are dead)
        f16x4 rv[FC][FP];
        if (res_fast) {
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const long long mr = (long long)min(m0 + wp * (BP / WPN) + j * 16 + lr, p.M - 1) * p.ldres;
#pragma unroll
                for (int i = 0; i < FC; ++i)
                    rv[i][j] = *(const f16x4*)((const f16*)res + mr + min(n0 + wc * (BC / 2) + i * 16 + lg * 4, p.Cout - 4));
            }
        }
        // bias of every channel fragment up front as well (a load + wait per fragment inside the loop exposes its latency FC times)
        f32x4 bvs[FC];
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
            bvs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                if (quad) bvs[i] = *(const f32x4*)(p.bias + min(n, p.Cout - 4));
                else for (int r = 0; r < 4; ++r) bvs[i][r] = n + r < p.Cout ? p.bias[n + r] : 0.f;
            }
        }
        auto finish = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int n = n0 + wc * (BC / 2) + i * 16 + lg * 4;
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    f32x4 v = acc[i][j] * p.out_scale + bvs[i];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rs_act_t<ACT, true>(v[r]);
                    if (res_fast) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[i][j][r];
                    } else if (res) {
                        const int m = m0 + wp * (BP / WPN) + j * 16 + lr;
                        if (m < p.M)
                            for (int r = 0; r < 4 && n + r < p.Cout; ++r) v[r] += (float)res[(long long)m * p.ldres + n + r];
                    }
                    f16x4 h;
                    h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)v[3];
                    *(f16x4*)(stg + (j * 16 + lr) * ROWB + (i * 16 + lg * 4) * 2) = h;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (p.act == RS_ACT_GELU) finish(std::integral_constant<int, RS_ACT_GELU>{});
        else if (p.act == RS_ACT_SILU) finish(std::integral_constant<int, RS_ACT_SILU>{});
        else finish(std::integral_constant<int, RS_ACT_NONE>{});
        __syncthreads();
        constexpr int CPR = (BC / 2) / 8;
        constexpr int NITEM = (BP / WPN) * CPR;
        const bool vec_ok = (p.ldy & 7) == 0;
        for (int idx = lane; idx < NITEM; idx += 64) {
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int m = m0 + wp * (BP / WPN) + row;
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
        // fp32 output of fp16 operands (AE attention logits and the like): direct stores, 16-byte vectors
        const bool vec_ok = ((p.ldy & 3) == 0) && quad;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / WPN) + j * 16 + lr;
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
                if (vec_ok && (!res || res_vec)) {
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

template <typename TO, int BC>
hipError_t launch3_cfg(IGemmParams p, hipStream_t st) {
    const int tiles = ((p.M + 255) / 256) * ((p.Cout + BC - 1) / BC);
    const size_t lds = (size_t)3 * (256 + BC) * 128;
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)igemm3_kernel<TO, BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * 2, wb = (size_t)p.Cout * p.Ktot * 2;
    if (xb >= 0xF0000000ull || wb >= 0xF0000000ull) return hipErrorInvalidValue;  // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    const int howo = p.Ho * p.Wo;
    const bool pow2 = howo > 0 && (howo & (howo - 1)) == 0 && (p.Wo & (p.Wo - 1)) == 0;
    p.sh_howo = pow2 ? __builtin_ctz(howo) : -1;
    p.sh_wo = pow2 ? __builtin_ctz(p.Wo) : -1;
    { static const int dbg = []() { const char* e = getenv("RS_IGEMM_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }   // timing ablations
    hipLaunchKernelGGL((igemm3_kernel<TO, BC>), dim3(tiles), dim3(512), lds, st, p);
    return hipGetLastError();
}

}  // namespace

// Which launches go to the 256-pixel kernel: fp16 operands, one source, no batching / split-K, an output width that
// tiles by 160 or 128 channels, a K loop long enough to amortise the deeper prologue, and a tile count in the range
// where ONE round of one-workgroup-per-CU tiles covers the launch (the 32x32 UNet level at batch 32: 256 tiles).
// Interleaved A/B runs (profiles/r1_igemm_ablation.txt): -7 % there; a tie on the AE convs (>= 2048 tiles) and +5 % on
// the 64x64 UNet level (512 tiles = two full rounds against one round of co-resident 128-pixel workgroups), which
// therefore stay on igemm2.  RS_IGEMM_V3=0 disables it; RS_IGEMM_V3_MINTILES / _MAXTILES / _MINK move the thresholds.
extern "C" int rs_igemm3_pick(int M, int Cout, int Ktot, int in_dt, int nz, int splitk, int* BC) {
    static const int on = []() { const char* e = getenv("RS_IGEMM_V3"); return e ? atoi(e) : 1; }();
    static const int min_tiles = []() { const char* e = getenv("RS_IGEMM_V3_MINTILES"); return e ? atoi(e) : 224; }();
    static const int max_tiles = []() { const char* e = getenv("RS_IGEMM_V3_MAXTILES"); return e ? atoi(e) : 320; }();
    static const int min_k = []() { const char* e = getenv("RS_IGEMM_V3_MINK"); return e ? atoi(e) : 1024; }();
    if (!on || in_dt != RS_F16 || nz != 1 || splitk > 1) return 0;
    int bc = 0;
    if (Cout % 160 == 0) bc = 160;
    else if (Cout % 128 == 0) bc = 128;
    else return 0;
    const long long tiles = (long long)((M + 255) / 256) * (Cout / bc);
    if (tiles < min_tiles || tiles > max_tiles || Ktot < min_k) return 0;
    *BC = bc;
    return 1;
}

extern "C" int rs_igemm3_launch(const IGemmParams* pp, int out_dt, int BC, hipStream_t st) {
    hipError_t e;
    if (out_dt == RS_F16) e = BC == 160 ? launch3_cfg<f16, 160>(*pp, st) : launch3_cfg<f16, 128>(*pp, st);
    else if (out_dt == RS_F32) e = BC == 160 ? launch3_cfg<float, 160>(*pp, st) : launch3_cfg<float, 128>(*pp, st);
    else return -2;
    return e == hipSuccess ? 0 : -1;
}
