// Implicit-GEMM kernel, second generation: same formulation / operand gather / epilogue as igemm.hip, but the K
// pipeline is built for latency hiding on CDNA4:
//   * 512 threads = 8 waves (4 pixel-waves x 2 channel-waves), tile BP x BC with BP in {128,256};
//   * operands travel global -> LDS with `global_load_lds_dwordx4` (LDS-DMA): no VGPR staging, no ds_write pass.
//     The LDS image of one instruction is lane-linear (wave-uniform base + lane*16 B), so the XOR swizzle that keeps
//     the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side: lane (row r, slot c) fetches the
//     K-chunk c ^ (r & 7) — free here because every lane computes its own gather address anyway;
//   * NS = 3 stage ring, prefetch distance 2 stages, counted `s_waitcnt vmcnt(L)` + raw `s_barrier` (one barrier per
//     stage; never vmcnt(0) inside the main loop), out-of-image / out-of-range lanes read a 16-byte zero buffer so
//     that every wave issues exactly L loads per stage and the counted waits stay valid.
#include "igemm_common.h"
#include <algorithm>

namespace {

using namespace igemm_detail;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename TI, typename TO, int BP, int BC, int NS>
__global__ __launch_bounds__(512) void igemm2_kernel(IGemmParams p) {
    constexpr int CH = MfmaOps<TI>::CH;
    constexpr int BK = 8 * CH;                 // elements of K per stage (128 bytes per row)
    constexpr int BCP = (BC + 63) / 64 * 64;   // weight rows rounded to whole 64-row load rounds
    constexpr int RX = BP / 64, RW = BCP / 64; // load rounds (one global_load_lds per thread per round)
    constexpr int L = RX + RW;                 // LDS-DMA instructions per thread per stage
    constexpr int FP = BP / 64;                // wave tile = (BP/4) pixels x (BC/2) channels
    constexpr int FC = BC / 32;
    constexpr int STAGE = (BP + BCP) * 128;
    constexpr unsigned INVALID = 0xFFFFFFFFu;
    static_assert(BP % 64 == 0 && BC % 32 == 0 && (NS == 2 || NS == 3), "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 3, wc = wave >> 2;
    const int rr = 8 * wave + (lane >> 3);             // row inside a 64-row load round
    const int kcp = (lane & 7) ^ ((lane >> 3) & 7);    // source K-chunk of this lane (swizzle on the source side)

    const int nby = (p.Cout + BC - 1) / BC;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nby) * BP;
    const int n0 = (tile % nby) * BC;
    const long long z = blockIdx.z;

    const TI* x0 = (const TI*)p.x0 + z * p.bs_x0;
    const TI* x1 = (const TI*)p.x1;
    const TI* w = (const TI*)p.w + z * p.bs_w;
    const TI* zeros = (const TI*)p.zeros;

    const int Ctot = p.C0 + p.C1;
    const int ntaps = p.KH * p.KW;
    const int Hv = p.Hs * p.up, Wv = p.Ws * p.up;
    const int ush = p.up == 2 ? 1 : 0;
    const int HoWo = p.Ho * p.Wo;

    int pixbase[RX], iy0[RX], ix0[RX];
#pragma unroll
    for (int i = 0; i < RX; ++i) {
        const int m = m0 + 64 * i + rr;
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
    // weight rows: 32-bit element offsets from w (INVALID -> zero row)
    unsigned woff[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int n = n0 + 64 * i + rr;
        woff[i] = (64 * i + rr < BC && n < p.Cout) ? (unsigned)n * (unsigned)p.Ktot : INVALID;
    }
    int kk = kcp * CH;
    int tap = kk / Ctot;
    int cc = kk - tap * Ctot;

    // Address generation is incremental: the per-row pixel offsets (in elements, per source) only change when this
    // lane's K cursor crosses into the next filter tap; between tap changes a stage costs one add per row.
    unsigned off0[RX], off1[RX];
    auto set_tap = [&]() {
        const bool kvalid = tap < ntaps;
        const int ky = tap / p.KW;
        const int kx = tap - ky * p.KW;
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kvalid && pixbase[i] >= 0 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            const unsigned pix = (unsigned)(pixbase[i] + (iy >> ush) * p.Ws + (ix >> ush));
            off0[i] = ok ? pix * (unsigned)p.ld0 : INVALID;
            off1[i] = ok ? pix * (unsigned)p.ld1 : INVALID;
        }
    };
    set_tap();

    // issue the L LDS-DMA loads of one K stage into ring slot `slot`, then step this lane's K cursor
    auto issue = [&](int slot) {
        char* sbase = smem + slot * STAGE + (8 * wave) * 128;   // wave-uniform
        const bool s0 = cc < p.C0;
        const TI* base = s0 ? x0 + cc : x1 + (cc - p.C0);
#pragma unroll
        for (int i = 0; i < RX; ++i) {
            const unsigned o = s0 ? off0[i] : off1[i];
            const TI* g = (o != INVALID) ? base + o : zeros;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(sbase + (64 * i) * 128), 16, 0, 0);
        }
        const TI* wb = w + kk;
        const bool wk = kk < p.Ktot;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const TI* g = (wk && woff[i] != INVALID) ? wb + woff[i] : zeros;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(sbase + (BP + 64 * i) * 128), 16, 0, 0);
        }
        kk += BK;
        cc += BK;
        if (cc >= Ctot) {
            do { cc -= Ctot; ++tap; } while (cc >= Ctot);
            set_tap();
        }
    };

    f32x4 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.Ktot + BK - 1) / BK;
    // prologue: fill NS-1 ring slots
    issue(0);
    if (NS == 3 && nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the loads of the NS-2 later stages are still outstanding
        if (NS == 3 && kt + 1 < nk) wait_vmcnt<L>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        // refill the slot every wave finished reading before this barrier
        if (kt + NS - 1 < nk) issue((kt + NS - 1) % NS);
        const char* sb = smem + (kt % NS) * STAGE;
        MfmaOps<TI>::template stage<FC, FP>(sb + BP * 128, sb, wc * (BC / 2), wp * (BP / 4), lr, lg, acc);
    }
    __syncthreads();  // all waves done with the ring: the epilogue reuses it as staging space

    // ---------------------------------------------------------------- epilogue (as in igemm.hip)
    TO* y = (TO*)p.y + z * p.bs_y;
    const TO* res = p.res ? (const TO*)p.res + z * p.bs_res : nullptr;
    const bool res_vec = res && (p.ldres & 3) == 0;
    if constexpr (sizeof(TO) == 2) {
        constexpr int ROWB = (BC / 2) * 2 + 16;
        char* stg = smem + wave * (BP / 4) * ROWB;
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / 4) + j * 16 + lr;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int nl = i * 16 + lg * 4;
                const int n = n0 + wc * (BC / 2) + nl;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.Cout) {
                    if (n + 3 < p.Cout) { const f32x4 t = *(const f32x4*)(p.bias + n); bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3]; }
                    else for (int r = 0; r < 4 && n + r < p.Cout; ++r) bv[r] = p.bias[n + r];
                }
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = epi_act<TO>(fmaf(acc[i][j][r], p.out_scale, bv[r]), p.act);
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
        constexpr int CPR = (BC / 2) / 8;
        constexpr int NITEM = (BP / 4) * CPR;
        const bool vec_ok = (p.ldy & 7) == 0;
        for (int idx = lane; idx < NITEM; idx += 64) {
            const int row = idx / CPR, c8 = idx - row * CPR;
            const int m = m0 + wp * (BP / 4) + row;
            const int n = n0 + wc * (BC / 2) + c8 * 8;
            if (m >= p.M || n >= p.Cout) continue;
            const uint4 v = *(const uint4*)(stg + row * ROWB + c8 * 16);
            TO* yp = y + (long long)m * p.ldy + n;
            if (vec_ok && n + 7 < p.Cout) {
                *(uint4*)yp = v;
            } else {
                const f16* hv = (const f16*)&v;
                for (int r = 0; r < 8 && n + r < p.Cout; ++r) yp[r] = (TO)hv[r];
            }
        }
    } else {
        const bool vec_ok = ((p.ldy & 3) == 0);
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int m = m0 + wp * (BP / 4) + j * 16 + lr;
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
                TO* yp = y + (long long)m * p.ldy + n;
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

const void* zero_buffer() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return z;
}

template <typename TI, typename TO, int BP, int BC, int NS>
hipError_t launch2_cfg(IGemmParams p, int nz, hipStream_t st) {
    constexpr int BCP = (BC + 63) / 64 * 64;
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    const size_t lds = (size_t)NS * (BP + BCP) * 128;
    static_assert(NS * (BP + BCP) * 128 <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm2_kernel<TI, TO, BP, BC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    p.zeros = zero_buffer();
    if (!p.zeros) return hipErrorOutOfMemory;
    hipLaunchKernelGGL((igemm2_kernel<TI, TO, BP, BC, NS>), dim3(tiles, 1, nz), dim3(512), lds, st, p);
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
    switch (BC) {
        case 160: return launch2_cfg<TI, TO, 128, 160, 3>(p, nz, st);
        case 192: return launch2_cfg<TI, TO, 128, 192, 3>(p, nz, st);
        default: return launch2_cfg<TI, TO, 128, 128, 3>(p, nz, st);
    }
}

}  // namespace

// Tile choice of the second-generation kernel; returns 0 when the launch should stay on igemm.hip (tiny Cout,
// too few tiles to fill the chip -> split-K there).
extern "C" int rs_igemm2_pick(int M, int Cout, int nz, int* BP, int* BC) {
    if (Cout <= 64) return 0;
    auto waste = [&](int bc) { return ((Cout + bc - 1) / bc) * bc - Cout; };
    int best = 128, bw = waste(128);
    if (waste(160) < bw) { best = 160; bw = waste(160); }
    if (waste(192) < bw) { best = 192; bw = waste(192); }
    *BC = best;
    const long long tiles128 = (long long)((M + 127) / 128) * ((Cout + best - 1) / best) * nz;
    if (tiles128 < 200) return 0;
    *BP = (tiles128 >= 512) ? 256 : 128;   // 256-pixel tiles once they still give >= 1 workgroup per CU
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
