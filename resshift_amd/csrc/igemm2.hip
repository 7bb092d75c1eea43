// Implicit-GEMM kernel, second generation: same formulation / operand gather / LDS image as igemm.hip, but
//   * operands travel global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR staging, no ds_write pass.
//     The LDS image of one instruction is lane-linear (wave-uniform base + lane*16 B), so the XOR swizzle that keeps
//     the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side: lane (row r, slot c) fetches the
//     K-chunk c ^ (r & 7) - free here because every lane computes its own gather address anyway.  Out-of-image taps,
//     ragged rows and the K tail use a byte offset beyond the descriptor's num_records: the hardware returns zeros;
//   * template <TI, TO, BP, BC, NS, NWV>: BP x BC output tile, NS-slot ring, NWV waves (NWV/2 pixel-waves x 2
//     channel-waves).  Default instantiation: 128 x {128,160,192}, 2 slots, 8 waves, <= 80 KB of LDS and <= 128 VGPRs so
//     that TWO workgroups share a CU and overlap each other's prologue / barrier / epilogue phases; 64 x BC on 4 waves for
//     the 16x16 / 8x8 levels (split-K); the deeper rings / 256-pixel / 16-wave instantiations are kept behind environment
//     knobs for A/B runs (profiles/r1_igemm_ablation.txt);
//   * raw `s_barrier` + counted `s_waitcnt vmcnt` (never __syncthreads() in the K loop: its fence would drain the DMA
//     prefetch), one barrier per K stage; stages 0 and 1 are issued together in the prologue;
//   * epilogue: bias / residual fetched up front, activation chosen at compile time behind one uniform branch, fragments
//     finished one at a time (register budget), fp16 results transposed through LDS into 16-byte NHWC stores.
#include "igemm_common.h"
#include <algorithm>
#include <type_traits>

extern "C" int rs_splitk_reduce_launch(const IGemmParams* p, int out_dt, hipStream_t st);

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// s_setprio(1) around the MFMA clusters measured neutral here (4 waves/SIMD already interleave); kept as a build-time knob
#ifndef RS_SETPRIO
#define RS_SETPRIO 0
#endif
// One K stage of MFMAs with precomputed per-lane LDS base pointers: fragment i sits at base + i*2048 (16 rows x 128 B),
// so every ds_read_b128 uses an immediate offset and the stage costs no address arithmetic beyond the 4 bases.
template <typename T, int FC, int FP> struct Stage2;
template <int FC, int FP> struct Stage2<f16, FC, FP> {
    static __device__ __forceinline__ void run(const char* a0, const char* a1, const char* b0, const char* b1, f32x4 (&acc)[FC][FP]) {
        // FC = 6 (BC = 192): both k-steps' fragments in flight would need > 128 VGPRs next to the epilogue state and spill
        // inside the loop; a sched_barrier between the k-steps keeps one fragment set live at a time (these launches are the
        // short-K, HBM-bound Swin GEMMs, where MFMA scheduling slack is irrelevant)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (FC >= 6 && ks == 1) __builtin_amdgcn_sched_barrier(0);
            const char* pa = ks ? a1 : a0;
            const char* pb = ks ? b1 : b0;
            f16x8 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) a[i] = *(const f16x8*)(pa + i * 2048);
#pragma unroll
            for (int j = 0; j < FP; ++j) b[j] = *(const f16x8*)(pb + j * 2048);
            if (RS_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (RS_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
    }
};
template <int FC, int FP> struct Stage2<float, FC, FP> {
    static __device__ __forceinline__ void run(const char* a0, const char* a1, const char* b0, const char* b1, f32x4 (&acc)[FC][FP]) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const char* pa = ss ? a1 : a0;
            const char* pb = ss ? b1 : b0;
            f32x4 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) a[i] = *(const f32x4*)(pa + i * 2048);
#pragma unroll
            for (int j = 0; j < FP; ++j) b[j] = *(const f32x4*)(pb + j * 2048);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][st], b[j][st], acc[i][j], 0, 0, 0);
        }
    }
};

// NB: keep the LDS-DMA builtin inside a plain __device__ function: called directly from the kernel template hipcc 7.2
// silently drops the template's HOST stubs (undefined __device_stub__ symbols at load time).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef RS_SPLIT_ABLATE
__device__ long long g_ig2_clk[4 * 8192];   // per workgroup: cycle counter at kernel start / K loop start / K loop end / kernel end (ablate builds)
#define RS_IG2_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0) g_ig2_clk[4 * blockIdx.x + (k)] = clock64()
#else
#define RS_IG2_STAMP(k)
#endif
template <typename TI, typename TO, int BP, int BC, int NS, int NWV>
// second launch bound = waves per SIMD the register allocation must allow: the 2-stage 128-pixel variant lives on TWO
// co-resident workgroups per CU (4 waves per SIMD, <= 128 VGPRs); one spilled-over register halves its occupancy
__global__ __launch_bounds__(64 * NWV, ((NS == 2 && BP == 128 && NWV == 8) || NWV == 16) ? 4 : 2) void igemm2_kernel(IGemmParams p) {
    RS_IG2_STAMP(0);
    constexpr int WPN = NWV / 2;               // pixel-waves (x 2 channel-waves)
    constexpr int RND = 8 * NWV;               // rows covered by one LDS-DMA instruction of every wave
    constexpr int CH = MfmaOps<TI>::CH;
    constexpr int BK = 8 * CH;                 // elements of K per stage (128 bytes per row)
    // Weight rows per stage: the 2-stage ring only ever waits with vmcnt(0), so its last load round may be partial (only the
    // waves that own rows < BC issue it: no zero rows DMA'd for BC = 160); the deeper rings count loads per wave and keep
    // whole rounds (rows beyond BC are fetched as hardware zeros).
    constexpr bool EXACT = (NS == 2);
    constexpr int BCP = EXACT ? BC : (BC + RND - 1) / RND * RND;
    constexpr int RWP = BC % RND;                     // rows of the partial round (a multiple of 8: one wave = 8 rows)
    static_assert(RWP % 8 == 0, "partial round must be whole waves");
    constexpr int RX = BP / RND, RW = (BC + RND - 1) / RND;   // load rounds (one LDS-DMA instruction per thread per round)
    constexpr int L = RX + RW;                 // LDS-DMA instructions per thread per stage
    constexpr int FP = BP / WPN / 16;          // wave tile = (BP/WPN) pixels x (BC/2) channels
    constexpr int FC = BC / 32;
    constexpr int STAGE = (BP + BCP) * 128;
    static_assert(BP % RND == 0 && (BP / WPN) % 16 == 0 && BC % 32 == 0 && (NS >= 2 && NS <= 4) && (NWV == 4 || NWV == 8 || NWV == 16), "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases (M0) are then pure SALU work
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave % WPN, wc = wave / WPN;
    const int rr = 8 * wave + (lane >> 3);             // row inside a 64-row load round
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);    // source K-chunk of this lane (swizzle on the source side)

    const int nby = (p.Cout + BC - 1) / BC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nby) * BP;
    const int n0 = (tile % nby) * BC;
    const long long z = blockIdx.z;

    // Operands are fetched with `buffer_load_dwordx4 ... lds`: a wave-uniform buffer descriptor per operand plus a 32-bit
    // per-lane BYTE offset.  Out-of-image taps, rows beyond M / Cout and the K tail use an offset beyond num_records, for
    // which the hardware returns zeros - no zero buffer, no pointer selects, and every wave still issues exactly L loads.
    constexpr unsigned INV = 0xF0000000u;   // > any tensor size handled here (checked by the launcher)
    constexpr unsigned SZ = sizeof(TI);
    const TI* x0 = (const TI*)p.x0 + (p.splitk > 1 ? 0 : z * p.bs_x0);
    const TI* w = (const TI*)p.w + (p.splitk > 1 ? 0 : z * p.bs_w);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x0, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, p.w_bytes, 0x00020000);

    const int Ctot = p.C0;                  // single source only (the launcher routes channel-concat convs to igemm.hip)
    const int ntaps = p.KH * p.KW;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const int HoWo = p.Ho * p.Wo;

    int pixbase[RX], iy0[RX], ix0[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int m = m0 + RND * i + rr;
        if (m < p.M) {
            int b, rem, oy, ox;
            if (p.sh_wo >= 0) {   // power-of-two output planes (every layer of the shipped models): shifts, no divisions
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
    unsigned woff[RW];                      // byte offset of this lane's weight rows
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int n = n0 + RND * i + rr;
        woff[i] = (RND * i + rr < BC && n < p.Cout) ? (unsigned)n * (unsigned)p.Ktot * SZ : INV;
    }
    // K range of this workgroup (split-K: grid.z slices the K stages; batch strides are unused then)
    const bool split = p.splitk > 1;
    const int nk_total = (p.Ktot + BK - 1) / BK;
    int kt0 = 0, nk = nk_total;
    if (split) {
        const int per = (nk_total + p.splitk - 1) / p.splitk;
        kt0 = min(nk_total, (int)z * per);
        nk = min(nk_total, kt0 + per) - kt0;
    }
    int kk = kt0 * BK + kcp * CH;
    int tap = kk / Ctot;
    int cc = kk - tap * Ctot;

    // Address generation is incremental: the per-row pixel byte offsets only change when this lane's K cursor crosses
    // into the next filter tap; between tap changes a stage costs one add per row.
    unsigned off[RX];
    int ky = tap / p.KW, kx = tap - ky * p.KW;   // kept incrementally afterwards
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

    // issue the L LDS-DMA loads of one K stage into ring slot `slot`, then step this lane's K cursor
    auto issue = [&](int slot) {
        char* sbase = smem + slot * STAGE + (8 * wave) * 128;   // wave-uniform
        const unsigned cb = (unsigned)cc * SZ;
#pragma unroll
        for (int i = 0; i < RX; ++i)
            lds_dma16(rx, sbase + (RND * i) * 128, off[i] + cb);
        const bool wk = kk < p.Ktot;
        const unsigned kb = (unsigned)kk * SZ;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (EXACT && RWP && i == RW - 1 && wave >= RWP / 8) continue;   // wave-uniform: this wave's rows of the last round are >= BC
            lds_dma16(rw, sbase + (BP + RND * i) * 128, wk ? woff[i] + kb : INV);
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

    // per-lane LDS fragment bases (relative to the ring slot): rows wrow0+lr / xrow0+lr, swizzled chunk for k-step 0/1
    const int swz0 = ((lg ^ (lr & 7)) << 4), swz1 = (((4 + lg) ^ (lr & 7)) << 4);
    const int la = BP * 128 + (wc * (BC / 2) + lr) * 128, lb = (wp * (BP / WPN) + lr) * 128;

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: fill NS-1 ring slots - and, with only two slots, the second one as well: both are empty at this point, so
    // stages 0 and 1 travel together and every workgroup saves one full load round trip (short-K launches have only 3)
    RS_IG2_STAMP(1);
    if (nk > 0) issue(0);
    const bool pro2 = NS >= 3 || !(p.dbg & 64);   // (dbg 64: the 2-slot ring starts with one stage, for A/B timing)
    if (nk > 1 && pro2) issue(1);
    if (NS >= 4 && nk > 2) issue(2);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the loads of the later stages already issued are still outstanding
        if (NS == 2) {
            if (kt == 0 && nk > 1 && pro2) {   // stage 1 was issued with stage 0; its loads (one fewer on the waves that skip the partial round) may fly on
                if (EXACT && RWP && wave >= RWP / 8) wait_vmcnt<L - 1>(); else wait_vmcnt<L>();
            } else {
                wait_vmcnt<0>();
            }
        } else {
            const int later = min(NS - 2, nk - 1 - kt);   // stages issued after kt that may still be in flight
            if (later >= 2) wait_vmcnt<2 * L>(); else if (later == 1) wait_vmcnt<L>(); else wait_vmcnt<0>();
        }
        if (!(p.dbg & 4)) __builtin_amdgcn_s_barrier();
        // refill the slot every wave finished reading before this barrier
        if (kt + NS - 1 < nk && !(NS == 2 && kt == 0 && pro2) && !(p.dbg & 1)) issue((kt + NS - 1) % NS);
        const char* sb = smem + (kt % NS) * STAGE;
        if (!(p.dbg & 2)) Stage2<TI, FC, FP>::run(sb + la + swz0, sb + la + swz1, sb + lb + swz0, sb + lb + swz1, acc);
    }
    RS_IG2_STAMP(2);
    __syncthreads();  // all waves done with the ring: the epilogue reuses it as staging space

    // ---------------------------------------------------------------- epilogue (as in igemm.hip)
    if (split) {
        // raw fp32 partial sums; scale / bias / activation / residual are applied by the split-K reduce kernel
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
                if (n + 3 < p.Cout && (p.Cout & 3) == 0) *(f32x4*)pp = acc[i][j];
                else for (int r = 0; r < 4 && n + r < p.Cout; ++r) pp[r] = acc[i][j][r];
            }
        }
        RS_IG2_STAMP(3);
        return;
    }
    TO* y = (TO*)p.y + z * p.bs_y;
    const TO* res = p.res ? (const TO*)p.res + z * p.bs_res : nullptr;
    const bool res_vec = res && (p.ldres & 3) == 0;
    if constexpr (sizeof(TO) == 2) {
        constexpr int ROWB = (BC / 2) * 2 + 16;
        char* stg = smem + wave * (BP / WPN) * ROWB;
        // The epilogue is VALU work that no MFMA overlaps inside this workgroup, so it is kept lean: the residual is
        // fetched up front with clamped (branch-free) vector loads that fly during the arithmetic, the bias once per
        // channel fragment, the activation is a compile-time choice behind ONE uniform branch, and fragments are
        // finished one at a time (sched_barrier) so that the register allocation stays within the 128 VGPRs that two
        // co-resident workgroups per CU allow.
        const bool quad = (p.Cout & 3) == 0;   // every 4-channel group is then entirely inside or outside the output
        const bool res_fast = res && res_vec && quad;
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
        RS_STAGING_SYNC();   // wave-private staging tile: the wave's own LDS order suffices, no workgroup barrier
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
        const bool vec_ok = ((p.ldy & 3) == 0);
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
    RS_IG2_STAMP(3);
}

template <typename TI, typename TO, int BP, int BC, int NS, int NWV = 8>
hipError_t launch2_cfg(IGemmParams p, int nz, hipStream_t st) {
    constexpr int RND = 8 * NWV;
    constexpr int BCP = NS == 2 ? BC : (BC + RND - 1) / RND * RND;
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    const size_t lds = (size_t)NS * (BP + BCP) * 128;
    static_assert(NS * (BP + BCP) * 128 <= 160 * 1024, "LDS");
    static RsAttrFlags attr_flags;
    if (attr_flags.need()) {
        (void)hipFuncSetAttribute((const void*)igemm2_kernel<TI, TO, BP, BC, NS, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const size_t esz = sizeof(TI);
    const size_t xb = (size_t)p.B * p.Hs * p.Ws * p.ld0 * esz, wb = (size_t)p.Cout * p.Ktot * esz;
    if (xb >= 0xF0000000ull || wb >= 0xF0000000ull) return hipErrorInvalidValue;  // 32-bit buffer offsets
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    { static const int dbg = []() { const char* e = getenv("RS_IGEMM_DBG"); return e ? atoi(e) : 0; }(); p.dbg = dbg; }
    {
        const int howo = p.Ho * p.Wo;
        const bool pow2 = howo > 0 && (howo & (howo - 1)) == 0 && (p.Wo & (p.Wo - 1)) == 0;
        p.sh_howo = pow2 ? __builtin_ctz(howo) : -1;
        p.sh_wo = pow2 ? __builtin_ctz(p.Wo) : -1;
    }
    hipLaunchKernelGGL((igemm2_kernel<TI, TO, BP, BC, NS, NWV>), dim3(tiles, 1, p.splitk > 1 ? p.splitk : nz), dim3(64 * NWV), lds, st, p);
    if (p.splitk > 1 && rs_splitk_reduce_launch(&p, sizeof(TO) == 2 ? RS_F16 : RS_F32, st) != 0) return hipErrorLaunchFailure;
    return hipGetLastError();
}

template <typename TI, typename TO>
hipError_t launch2_t(const IGemmParams& p, int BP, int BC, int nz, hipStream_t st) {
    if (BP == 256) {
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 256, 160, 2>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 256, 192, 2>(p, nz, st);
            default: return launch2_cfg<TI, TO, 256, 128, 3>(p, nz, st);
        }
    }
    if (BP == 132) {
        // launches with fewer workgroups than CUs (16x16 / 8x8 UNet levels): latency-bound K loop -> deepest ring that fits
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 128, 160, 3>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 128, 192, 3>(p, nz, st);
            default: return launch2_cfg<TI, TO, 128, 128, 4>(p, nz, st);
        }
    }
    if (BP == 131) {
        // 16-wave variant: 256-pixel tile (25 % less L2->LDS traffic per FLOP than 128x128) with the same 32 x BC/2 wave tiles
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 256, 160, 2, 16>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 256, 192, 2, 16>(p, nz, st);
            default: return launch2_cfg<TI, TO, 256, 128, 2, 16>(p, nz, st);
        }
    }
    if (BP == 130) {
        // 4-wave variant: wave tile 64 x BC/2 (fewer LDS fragment reads per MFMA), two workgroups per CU
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 128, 160, 2, 4>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 128, 192, 2, 4>(p, nz, st);
            default: return launch2_cfg<TI, TO, 128, 128, 2, 4>(p, nz, st);
        }
    }
    if (BP == 133) {
        // small-M launches (16x16 / 8x8 UNet levels): 64-pixel tiles on 4 waves give twice the workgroups per split-K slice,
        // i.e. half the fp32 partial-slab traffic for the same number of workgroups
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 64, 160, 2, 4>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 64, 192, 2, 4>(p, nz, st);
            default: return launch2_cfg<TI, TO, 64, 128, 2, 4>(p, nz, st);
        }
    }
    if (BP == 129) {
        // short-K GEMMs (K <= 4 stages: Swin qkv/proj/fc1, patch embeds): the launch is dominated by prologue + epilogue,
        // so use the 2-stage ring (<= 80 KB of LDS) that lets TWO workgroups share a CU and overlap each other
        switch (BC) {
            case 160: return launch2_cfg<TI, TO, 128, 160, 2>(p, nz, st);
            case 192: return launch2_cfg<TI, TO, 128, 192, 2>(p, nz, st);
            default: return launch2_cfg<TI, TO, 128, 128, 2>(p, nz, st);
        }
    }
    switch (BC) {
        case 160: return launch2_cfg<TI, TO, 128, 160, 3>(p, nz, st);
        case 192: return launch2_cfg<TI, TO, 128, 192, 3>(p, nz, st);
        default: return launch2_cfg<TI, TO, 128, 128, 3>(p, nz, st);
    }
}

}  // namespace

// Tile choice of the second-generation kernel; returns 0 when the launch should stay on igemm.hip (tiny Cout,
// too few tiles to fill the chip -> split-K there).
#ifdef RS_SPLIT_ABLATE
// ablate builds: mean cycles of the three kernel phases over the first `nwg` workgroups of the last igemm2 launch
extern "C" int rs_igemm2_phase_cycles(int nwg, double* out3) {
    static long long h[4 * 8192];
    if (nwg < 1 || nwg > 8192) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ig2_clk), sizeof(long long) * 4 * nwg) != hipSuccess) return -1;
    out3[0] = out3[1] = out3[2] = 0.0;
    for (int i = 0; i < nwg; ++i)
        for (int k = 0; k < 3; ++k) out3[k] += (double)(h[4 * i + k + 1] - h[4 * i + k]) / nwg;
    return 0;
}
#endif

extern "C" int rs_igemm2_pick(int M, int Cout, int Kbytes, int nz, int* BP, int* BC) {
    if (Cout <= 64) return 0;
    auto waste = [&](int bc) { return ((Cout + bc - 1) / bc) * bc - Cout; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    *BC = best;
    const long long tiles128 = (long long)((M + 127) / 128) * ((Cout + best - 1) / best) * nz;
    (void)tiles128;
    *BP = (tiles128 >= 512) ? 256 : 128;   // 256-pixel tiles once they still give >= 1 workgroup per CU
    // measured (profiles/r1_igemm_microbench_*): the 128-pixel, 2-stage, two-workgroups-per-CU variant wins on every layer
    // shape of the model; RS_IGEMM_SHORTK=<bytes> restricts it to short-K GEMMs (256-pixel tiles otherwise) for A/B runs
    static const int shortk = []() { const char* e = getenv("RS_IGEMM_SHORTK"); return e ? atoi(e) : (1 << 30); }();
    if (Kbytes <= shortk) *BP = 129;
    static const int var4 = []() { const char* e = getenv("RS_IGEMM_4WAVE"); return e ? atoi(e) : 0; }();
    static const int deep = []() { const char* e = getenv("RS_IGEMM_DEEP"); return e ? atoi(e) : 0; }();  // measured: no gain on the 8x8/16x16 levels (fixed launch cost dominates)
    if (tiles128 <= deep) *BP = 132;
    // measured (profiles/r1_igemm_microbench_v6_smallm.txt): -16 % on the 16x16 / 8x8 level launches of one pass
    static const int smallm = []() { const char* e = getenv("RS_IGEMM_SMALLM"); return e ? atoi(e) : 8192; }();
    static const int force64 = []() { const char* e = getenv("RS_IGEMM_FORCE64"); return e ? atoi(e) : 0; }();   // A/B knob
    if ((M <= smallm && tiles128 < 512) || force64) *BP = 133;   // (batched GEMMs with many small batches keep the 128-pixel tile)
    if (var4 == 1) *BP = 130;
    if (var4 == 16 && tiles128 >= 1024) *BP = 131;   // marker for the 128-pixel / 2-stage / 2-workgroups-per-CU variant
    return 1;
}

extern "C" int rs_igemm2_launch(const IGemmParams* pp, int in_dt, int out_dt, int BP, int BC, int nz, hipStream_t st) {
    const IGemmParams& p = *pp;
    hipError_t e;
    if (in_dt == RS_F16 && out_dt == RS_F16) e = launch2_t<f16, f16>(p, BP, BC, nz, st);
    else if (in_dt == RS_F16 && out_dt == RS_F32) e = launch2_t<f16, float>(p, BP, BC, nz, st);
    else if (in_dt == RS_F32 && out_dt == RS_F32) e = launch2_t<float, float>(p, BP, BC, nz, st);
    else return -2;
    return e == hipSuccess ? 0 : -1;
}
