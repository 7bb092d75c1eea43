// GroupNorm(32) [+FiLM] [+SiLU], Swin window attention core, and row softmax for gfx950.
//
//   GroupNorm: models/basic_ops.py:15-17,89-96 (fp32 statistics, eps 1e-5) and
//              ldm/modules/diffusionmodules/model.py:46-47 (eps 1e-6); FiLM modulation
//              models/unet.py:198-202; SiLU models/basic_ops.py:10-12, model.py:41-43.
//   Window attention: models/swin_transformer.py:114-145 (scale, QK^T, relative position bias,
//              shift mask, softmax, PV) with roll / window_partition / window_reverse
//              (swin_transformer.py:35-63,252-275) folded into the load/store addressing.
//   Row softmax: ldm/modules/diffusionmodules/model.py:193 (AE mid-block attention).
#include "common.h"
#include "gn_tail.h"
#include <type_traits>

namespace {

// ---------------------------------------------------------------- GroupNorm
// stats: grid (S, B).  Thread (pp, j) owns 8-channel chunk j and walks pixels pp, pp+PP, ...
// Deterministic: per-thread partial sums go to LDS, one thread per group reduces them in a fixed order.
// `tail.coef` set (coefficient-only GroupNorms of tensors whose producer leaves no statistics): the partials are published write-through
// and the last workgroup of every image turns them into the affine coefficients (gn_tail.h) - statistics + coefficients in ONE launch.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(GNParams p, GNTail tail) {
    __shared__ float red[256][17];
    const int s = blockIdx.x, b = blockIdx.y;
    const int nchunk = p.C >> 3;
    const int PP = 256 / nchunk;
    const int tid = threadIdx.x;
    const int j = tid % nchunk, pp = tid / nchunk;
    const int pps = (p.HW + p.S - 1) / p.S;
    const int p0 = s * pps;
    const int p1 = min(p.HW, p0 + pps);
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
    if (pp < PP) {
        const long long rx = (long long)p.ldx * Store<T>::PM;   // pixel record length (split storage: hi + lo)
        const T* x = (const T*)p.x + ((long long)b * p.HW) * rx + j * 8;
        // 4 independent 16-byte loads in flight per thread (the loop is latency-, not bandwidth-bound otherwise)
        int pix = p0 + pp;
        for (; pix + 3 * PP < p1; pix += 4 * PP) {
            Vec8<T> v0, v1, v2, v3;
            v0.load(x + (long long)pix * rx, p.ldx);
            v1.load(x + (long long)(pix + PP) * rx, p.ldx);
            v2.load(x + (long long)(pix + 2 * PP) * rx, p.ldx);
            v3.load(x + (long long)(pix + 3 * PP) * rx, p.ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f0 = v0.get(e), f1 = v1.get(e), f2 = v2.get(e), f3 = v3.get(e);
                sum[e] += (f0 + f1) + (f2 + f3);
                sq[e] = fmaf(f0, f0, fmaf(f1, f1, fmaf(f2, f2, fmaf(f3, f3, sq[e]))));
            }
        }
        for (; pix < p1; pix += PP) {
            Vec8<T> v;
            v.load(x + (long long)pix * rx, p.ldx);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = v.get(e); sum[e] += f; sq[e] = fmaf(f, f, sq[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = sum[e]; red[tid][8 + e] = sq[e]; }
    __syncthreads();
    if (tid < p.groups) {
        const int cpg = p.C / p.groups;
        float a = 0.f, q = 0.f;
        for (int r = 0; r < PP; ++r)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                const int t = r * nchunk + (c >> 3);
                a += red[t][c & 7];
                q += red[t][8 + (c & 7)];
            }
        float* out = p.partial + (((long long)b * p.S + s) * p.groups + tid) * 2;
        if (tail.coef) rs_pub_pair(out, a, q);
        else { out[0] = a; out[1] = q; }
    }
    // (wave 0 stored the partials - tid < groups <= 32 - and draws the ticket; `red`, 256 x 17 floats, is the last workgroup's scratch)
    if (tail.coef) {
        unsigned* const flag = (unsigned*)&red[255][16];   // (behind the coefficient scratch: 2 C + 2 groups <= 4160 floats)
        if (tid < 64) { const bool last = rs_gn_tail_arrive(tail, b); if (tid == 0) *flag = last ? 1u : 0u; }
        __syncthreads();
        if (*flag) rs_gn_tail_finish<256>(tail, b, &red[0][0]);
    }
}

// Element pass of the apply kernel.  Thread (pp, j) owns the 8-channel chunk j for good - its 16 affine coefficients
// live in registers - and walks pixels pp, pp+PP, ...: no index divisions and no LDS reads in the loop, four 16-byte
// loads in flight per thread.  ACT is compile-time (no per-element branches); fp16 storage uses the v_rcp/v_exp SiLU.
template <typename T, int ACT>
__device__ __forceinline__ void gn_apply_rows(const GNParams& p, const float* ca, const float* cb, int b) {
    constexpr bool FAST = Store<T>::FAST;
    const int nchunk = p.C >> 3;
    const int PP = 256 / nchunk;
    const int tid = threadIdx.x;
    const int j = tid % nchunk, pp = tid / nchunk;
    if (pp >= PP) return;
    const long long rx = (long long)p.ldx * Store<T>::PM, ry = (long long)p.ldy * Store<T>::PM;
    float a[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = ca[j * 8 + e]; c[e] = cb[j * 8 + e]; }
    const int nslab = gridDim.x;
    const int pps = (p.HW + nslab - 1) / nslab;
    const int p0 = blockIdx.x * pps;
    const int p1 = min(p.HW, p0 + pps);
    const T* x = (const T*)p.x + ((long long)b * p.HW) * rx + j * 8;
    T* y = (T*)p.y + ((long long)b * p.HW) * ry + j * 8;
    int pix = p0 + pp;
    for (; pix + 3 * PP < p1; pix += 4 * PP) {
        Vec8<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].load(x + (long long)(pix + u * PP) * rx, p.ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Vec8<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, rs_act_t<ACT, FAST>(fmaf(v[u].get(e), a[e], c[e])));
            o.store(y + (long long)(pix + u * PP) * ry, p.ldy);
        }
    }
    for (; pix < p1; pix += PP) {
        Vec8<T> v, o;
        v.load(x + (long long)pix * rx, p.ldx);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, rs_act_t<ACT, FAST>(fmaf(v.get(e), a[e], c[e])));
        o.store(y + (long long)pix * ry, p.ldy);
    }
}

// apply: grid (S2, B).  y = act( ((x-mean)*rstd*gamma+beta) * (1+scale) + shift )
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ca = (float*)smem;          // [C]
    float* cb = ca + p.C;              // [C]
    float* gm = cb + p.C;              // [groups]
    float* gr = gm + p.groups;         // [groups]
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.groups;
    // reduce the S partial sums of every group: 256/groups threads per group, then a fixed-order tree (deterministic)
    float* ps = gr + p.groups;         // [nsl][groups][2]
    const int nsl = p.cpartial ? 1 : 256 / p.groups;
    if (p.cpartial) {
        // per-channel partials of the producing kernel's epilogue ([B][S][cp_ld][2]: halo conv tiles, split-K reduce slabs, attention
        // windows, MLP wave tiles - up to 128 sets per image): thread = channel, consecutive threads read consecutive channels of one
        // set (coalesced), S independent loads in flight per thread; then channels -> groups through LDS.  (One thread per (group,
        // slice) walking cpg strided pairs per set made this launch as long as the statistics pass it replaces.)
        float* chs = ca;               // [C][2] scratch in the coefficient arrays (2 * C floats, rewritten below)
        for (int c = tid; c < p.C; c += 256) {
            // (two producers: the second one's partials cover the channels from cp_n0 on - same loop as the tails', gn_tail.h)
            const bool s0 = !p.cpartial2 || c < p.cp_n0;
            const float* in = s0 ? p.cpartial + (((long long)b * p.S) * p.cp_ld + c) * 2 : p.cpartial2 + (((long long)b * p.cp2_S) * p.cp2_ld + (c - p.cp_n0)) * 2;
            const int S = s0 ? p.S : p.cp2_S;
            const long long step = 2ll * (s0 ? p.cp_ld : p.cp2_ld);
            float a = 0.f, q = 0.f;
            for (int s2 = 0; s2 < S; ++s2) { const float2 v = *(const float2*)(in + s2 * step); a += v.x; q += v.y; }
            chs[2 * c] = a; chs[2 * c + 1] = q;
        }
        __syncthreads();
        if (tid < p.groups) {
            float a = 0.f, q = 0.f;
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += chs[2 * c]; q += chs[2 * c + 1]; }
            ps[tid * 2] = a; ps[tid * 2 + 1] = q;
        }
    } else {
        const int g = tid % p.groups, sl = tid / p.groups;
        float a = 0.f, q = 0.f;
        if (sl < nsl) {
            for (int s = sl; s < p.S; s += nsl) {
                const float* in = p.partial + (((long long)b * p.S + s) * p.groups + g) * 2;
                a += in[0]; q += in[1];
            }
            ps[(sl * p.groups + g) * 2] = a; ps[(sl * p.groups + g) * 2 + 1] = q;
        }
    }
    __syncthreads();
    if (tid < p.groups) {
        float a = 0.f, q = 0.f;
        for (int sl = 0; sl < nsl; ++sl) { a += ps[(sl * p.groups + tid) * 2]; q += ps[(sl * p.groups + tid) * 2 + 1]; }
        const float n = (float)cpg * (float)p.HW;
        rs_gn_group(a, q, n, p.eps, gm[tid], gr[tid]);   // (gn_tail.h: the tails use the same expressions)
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / cpg;
        float a, bb;
        rs_gn_channel(p.gamma[c], p.beta[c], gm[g], gr[g], p.film, c, p.C, a, bb);
        ca[c] = a; cb[c] = bb;
    }
    __syncthreads();
    if (p.coef) {   // coefficient-only mode (launched with one slab per image): hand the affine to the consumer kernel
        for (int c = tid; c < p.C; c += 256) { p.coef[((long long)b * 2) * p.C + c] = ca[c]; p.coef[((long long)b * 2 + 1) * p.C + c] = cb[c]; }
        return;
    }
    if (p.act == RS_ACT_SILU) gn_apply_rows<T, RS_ACT_SILU>(p, ca, cb, b);
    else if (p.act == RS_ACT_GELU) gn_apply_rows<T, RS_ACT_GELU>(p, ca, cb, b);
    else gn_apply_rows<T, RS_ACT_NONE>(p, ca, cb, b);
}

// Fused GroupNorm for small planes (HW <= 256: the 16x16 / 8x8 UNet levels): grid (C / SC, B), one workgroup owns a slice of
// SC channels (whole groups, a multiple of 8 channels) of ONE image, keeps it in registers (<= MAXI 16-byte items per
// thread), derives the group statistics through LDS and writes the normalised result - one launch and one read of x
// instead of stats + apply.  Thread (pp, j) owns chunk j of the slice for pixels pp, pp+PP, ...
template <typename T, int ACT, int MAXI, int NT>
__device__ __forceinline__ void gn_fused_body(const GNParams& p, int SC, float (*red)[17], float* ca, float* cb) {
    constexpr bool FAST = Store<T>::FAST;
    const long long rx = (long long)p.ldx * Store<T>::PM, ry = (long long)p.ldy * Store<T>::PM;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.groups;
    const int nch = SC >> 3, PP = NT / nch;
    const int j = tid % nch, pp = tid / nch;
    const int cs = blockIdx.x * SC;               // first channel of the slice
    const bool active = pp < PP;
    const T* x = (const T*)p.x + ((long long)b * p.HW) * rx + cs + j * 8;
    T* y = (T*)p.y + ((long long)b * p.HW) * ry + cs + j * 8;
    Vec8<T> v[MAXI];
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int pix = pp + k * PP;
        if (active && pix < p.HW) v[k].load(x + (long long)pix * rx, p.ldx);
    }
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int pix = pp + k * PP;
        if (active && pix < p.HW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = v[k].get(e); sum[e] += f; sq[e] = fmaf(f, f, sq[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid][e] = sum[e]; red[tid][8 + e] = sq[e]; }
    __syncthreads();
    // fixed-order (deterministic) reduction in two short steps: per-channel column sums over the PP pixel lanes, then
    // channels -> groups (a single thread per group walking PP x cpg entries would be a 500-deep dependent LDS chain)
    if (tid < nch * 16) {
        const int jj = tid >> 4, e16 = tid & 15;
        float acc = 0.f;
        for (int r = 0; r < PP; ++r) acc += red[r * nch + jj][e16];
        cb[tid] = acc;                            // cb doubles as the column-sum scratch: [chunk][sum 0..7 | sq 0..7]
    }
    __syncthreads();
    const int gps = SC / cpg;                     // groups in this slice
    float gmean = 0.f, grstd = 0.f;
    if (tid < gps) {
        float a = 0.f, q = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
            a += cb[(c >> 3) * 16 + (c & 7)];
            q += cb[(c >> 3) * 16 + 8 + (c & 7)];
        }
        const float n = (float)cpg * (float)p.HW;
        rs_gn_group(a, q, n, p.eps, gmean, grstd);
    }
    __syncthreads();                              // column sums consumed: ca / cb now take the affine coefficients
    if (tid < gps) {
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
            float ga, be;
            rs_gn_channel(p.gamma[cs + c], p.beta[cs + c], gmean, grstd, p.film, cs + c, p.C, ga, be);
            ca[c] = ga; cb[c] = be;
        }
    }
    __syncthreads();
    if (p.coef) {   // coefficient-only mode
        for (int c = tid; c < SC; c += NT) { p.coef[((long long)b * 2) * p.C + cs + c] = ca[c]; p.coef[((long long)b * 2 + 1) * p.C + cs + c] = cb[c]; }
        return;
    }
    if (!active) return;
    float a[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = ca[j * 8 + e]; c[e] = cb[j * 8 + e]; }
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int pix = pp + k * PP;
        if (pix < p.HW) {
            Vec8<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, rs_act_t<ACT, FAST>(fmaf(v[k].get(e), a[e], c[e])));
            o.store(y + (long long)pix * ry, p.ldy);
        }
    }
}

template <typename T, int MAXI, int NT>
__global__ __launch_bounds__(NT) void gn_fused_kernel(GNParams p, int SC) {
    __shared__ float red[NT][17];
    __shared__ float ca[256], cb[512];
    if (p.act == RS_ACT_SILU) gn_fused_body<T, RS_ACT_SILU, MAXI, NT>(p, SC, red, ca, cb);
    else if (p.act == RS_ACT_GELU) gn_fused_body<T, RS_ACT_GELU, MAXI, NT>(p, SC, red, ca, cb);
    else gn_fused_body<T, RS_ACT_NONE, MAXI, NT>(p, SC, red, ca, cb);
}

// slice width for the fused kernel: whole groups, a multiple of 8 channels, >= 40 channels where possible (80-byte runs per
// pixel), and narrow enough that the plane fits MAXI items per thread; 0 when no such width exists (two-kernel path)
template <int MAXI, int NT> int gn_fused_slice(const GNParams& p) {
    const int cpg = p.C / p.groups;
    int g = 1;
    while ((g * cpg) % 8) ++g;                       // g in {1,2,4,8}
    if (g > p.groups || p.groups % g) return 0;
    int best = 0;
    for (int SC = g * cpg; SC <= 256 && SC <= p.C; SC *= 2) {
        if (p.C % SC) break;
        const int PP = NT / (SC / 8);
        if (PP < 1 || (p.HW + PP - 1) / PP > MAXI) break;
        best = SC;
        if (SC >= 40) break;
    }
    return best;
}

}  // namespace

// Returns the number of kernels launched (1 or 2), or a negative error code.
// `pp->ticket` (with `coef`): coefficient-only GroupNorm of a tensor without producer statistics in ONE launch - the statistics kernel's
// last workgroup per image writes the coefficients (gn_tail.h).  RS_GN_TAIL=0 keeps the two launches (A/B runs).
extern "C" int rs_groupnorm_launch(const GNParams* pp, int dt, int apply_slabs, hipStream_t st) {
    const GNParams& p = *pp;
    if ((p.C % 8) || (p.C % p.groups) || p.C / 8 > 256 || p.groups > 256 || (p.ldx % 8) || (p.ldy % 8)) return -2;
    GNTail tail{};
    // small planes: one fused launch (RS_GN_FUSED=0 keeps the two-kernel path for A/B runs)
    static const bool fused_on = []() { const char* e = getenv("RS_GN_FUSED"); return !(e && e[0] == '0'); }();
    // measured at B=32 (8-step bench, ms/step): fused up to 256 px 174.1, up to 1024 px 174.1, up to 4096 px 176.8 (too few,
    // too fat workgroups at 64x64) -> the 1024-thread variant stays off by default
    static const int fused_max = []() { const char* e = getenv("RS_GN_FUSED_MAXHW"); return e ? atoi(e) : 256; }();
    if (fused_on && p.HW <= 256 && !p.cpartial) {
        constexpr int MAXI = 12;
        const int SC = gn_fused_slice<MAXI, 256>(p);
        if (SC > 0) {
            dim3 g(p.C / SC, p.B);
            if (dt == RS_F16) hipLaunchKernelGGL((gn_fused_kernel<f16, MAXI, 256>), g, dim3(256), 0, st, p, SC);
            else if (dt == RS_F16S) hipLaunchKernelGGL((gn_fused_kernel<h2s, MAXI, 256>), g, dim3(256), 0, st, p, SC);
            else hipLaunchKernelGGL((gn_fused_kernel<float, MAXI, 256>), g, dim3(256), 0, st, p, SC);
            return hipGetLastError() == hipSuccess ? 1 : -1;
        }
    } else if (fused_on && dt == RS_F16 && p.HW <= fused_max && !p.cpartial) {
        // 32x32 / 64x64 planes in fp16: 1024 threads hold up to 20 items (80 VGPRs) each; the tensor is read once instead of twice
        constexpr int MAXI = 20;
        const int SC = gn_fused_slice<MAXI, 1024>(p);
        if (SC > 0) {
            hipLaunchKernelGGL((gn_fused_kernel<f16, MAXI, 1024>), dim3(p.C / SC, p.B), dim3(1024), 0, st, p, SC);
            return hipGetLastError() == hipSuccess ? 1 : -1;
        }
    }
    dim3 g1(p.S, p.B), g2(p.coef ? 1 : apply_slabs, p.B);
    const size_t lds = (2 * p.C + 2 * p.groups + 2 * 256) * sizeof(float);
    const bool need_stats = p.cpartial == nullptr;   // (else the producing conv's epilogue already left per-channel partial sums)
    static const bool tail_on = []() { const char* e = getenv("RS_GN_TAIL"); return !(e && e[0] == '0'); }();
    // (the tail's finish works in gn_stats_kernel's red[256][17] array: 2 C + 2 groups floats below the flag word at red[255][16], a slice
    // tree of 256 / groups lanes per group - ADVICE r4: enforce what the engine's groups = 32 satisfies by construction)
    const bool use_tail = tail_on && need_stats && p.coef && p.ticket && 2 * p.C + 2 * p.groups <= 4351 && p.groups > 0 && (256 % p.groups) == 0;
    if (use_tail) {
        tail.gamma = p.gamma; tail.beta = p.beta; tail.film = p.film; tail.coef = p.coef; tail.ticket = p.ticket; tail.expected = p.S;
        tail.C = p.C; tail.groups = p.groups; tail.HW = p.HW; tail.eps = p.eps; tail.stg = p.partial; tail.Sg = p.S;
    }
    if (dt == RS_F16) {
        if (need_stats) hipLaunchKernelGGL((gn_stats_kernel<f16>), g1, dim3(256), 0, st, p, tail);
        if (!use_tail) hipLaunchKernelGGL((gn_apply_kernel<f16>), g2, dim3(256), lds, st, p);
    } else if (dt == RS_F16S) {
        if (need_stats) hipLaunchKernelGGL((gn_stats_kernel<h2s>), g1, dim3(256), 0, st, p, tail);
        if (!use_tail) hipLaunchKernelGGL((gn_apply_kernel<h2s>), g2, dim3(256), lds, st, p);
    } else {
        if (need_stats) hipLaunchKernelGGL((gn_stats_kernel<float>), g1, dim3(256), 0, st, p, tail);
        if (!use_tail) hipLaunchKernelGGL((gn_apply_kernel<float>), g2, dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? (need_stats && !use_tail ? 2 : 1) : -1;
}

// ---------------------------------------------------------------- window attention

namespace {

// grid (windows per image, B); block = 64*heads threads; thread (h, i) owns query token i of head h.
// K and V of the whole window are staged in LDS as fp32; a wave holds one head, so every LDS read in
// the score / PV loops is a broadcast.
template <typename T>
__global__ __launch_bounds__(512) void win_attn_kernel(WinAttnParams p) {
    constexpr int HD = 32, WS = 8, NT = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = (float*)smem;                      // [heads][64][32]
    float* vs = ks + p.heads * NT * HD;            // [heads][64][32]
    int* rid = (int*)(vs + p.heads * NT * HD);     // [64]
    const int tid = threadIdx.x;
    const int h = tid >> 6, i = tid & 63;
    const int nwx = p.W / WS;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x - wy * nwx;
    const int b = blockIdx.y;
    const int E = p.heads * HD;
    // token i of this window on the shifted grid, and its source pixel (roll by -shift == read (+shift) % size)
    const int ys = wy * WS + (i >> 3), xs = wx * WS + (i & 7);
    int sy = ys + p.shift; if (sy >= p.H) sy -= p.H;
    int sx = xs + p.shift; if (sx >= p.W) sx -= p.W;
    const long long pix = ((long long)b * p.H + sy) * p.W + sx;
    const T* src = (const T*)p.qkv + pix * p.ldq * Store<T>::PM + h * HD;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        Vec8<T> vq, vk, vv;
        vq.load(src + c, p.ldq);
        vk.load(src + E + c, p.ldq);
        vv.load(src + 2 * E + c, p.ldq);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q[c + e] = vq.get(e) * p.scale;
            ks[(h * NT + i) * HD + c + e] = vk.get(e);
            vs[(h * NT + i) * HD + c + e] = vv.get(e);
        }
    }
    if (tid < NT && p.shift > 0) {
        // region ids of calculate_mask (swin_transformer.py:214-236) on the shifted grid.  The reference indexes its
        // (1,1,H,W) mask image as [:, h, w, :], so its "h" slices hit the singleton dim and its "w" slices hit the ROW
        // axis: the effective region id is the row band only (pinned against the reference, oracle/make_golden.py).
        // Second quirk: window_partition(img_mask) is followed by a further .permute(0,2,3,1) (:230), which swaps the
        // row/column roles inside each window before flattening, so token (r,c) receives the label of position (c,r).
        // Net effect: region(token i) = row band of (window_row*8 + token_column).
        const int yq = wy * WS + (i & 7);
        rid[tid] = yq < p.H - WS ? 0 : (yq < p.H - p.shift ? 1 : 2);
    }
    __syncthreads();
    float sc[NT];
    const float* kh = ks + h * NT * HD;
    const float* bt = p.bias_t + (long long)h * NT * NT + i;
    const int myrid = p.shift > 0 ? rid[i] : 0;
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const f32x4* kr = (const f32x4*)(kh + j * HD);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            const f32x4 kv = kr[c];
            a = fmaf(q[4 * c + 0], kv[0], a);
            a = fmaf(q[4 * c + 1], kv[1], a);
            a = fmaf(q[4 * c + 2], kv[2], a);
            a = fmaf(q[4 * c + 3], kv[3], a);
        }
        a += bt[j * NT];
        if (p.shift > 0 && rid[j] != myrid) a += -100.0f;
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    float denom = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { sc[j] = expf(sc[j] - mx); denom += sc[j]; }
    const float inv = 1.0f / denom;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    const float* vh = vs + h * NT * HD;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float pj = sc[j] * inv;
        const f32x4* vr = (const f32x4*)(vh + j * HD);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            const f32x4 vv = vr[c];
            o[4 * c + 0] = fmaf(pj, vv[0], o[4 * c + 0]);
            o[4 * c + 1] = fmaf(pj, vv[1], o[4 * c + 1]);
            o[4 * c + 2] = fmaf(pj, vv[2], o[4 * c + 2]);
            o[4 * c + 3] = fmaf(pj, vv[3], o[4 * c + 3]);
        }
    }
    // window_reverse + reverse roll: the result goes back to the pixel the token was read from
    T* dst = (T*)p.out + pix * p.ldo * Store<T>::PM + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        Vec8<T> ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov.set(e, o[c + e]);
        ov.store(dst + c, p.ldo);
    }
}

}  // namespace

namespace {

// fp16 window attention on the matrix cores: one wave per (window, head), everything between the qkv tensor and the
// output stays in registers except V, which is transposed through LDS.
//   S^T = K Q^T   : 16 x v_mfma_f32_16x16x32_f16, A = K rows (16 B straight from global), B = Q rows (ditto);
//                   lane (lr,lg) ends with S^T[j = 16fj+4lg+r][i = 16fi+lr]: all 64 keys of query i live in the 4
//                   lanes lr, lr+16, lr+32, lr+48 -> the softmax row reduction is 2 xor-shuffles (16, 32).
//   O^T = V^T P^T : 16 more MFMAs.  The B operand wants, per lane, 8 keys of ONE query: exactly the registers the lane
//                   already holds (frags 2s and 2s+1), because the k-slot -> key map may be any bijection as long as
//                   the A operand (V^T, read from LDS with two ds_read_b64) uses the same one.  No P round trip.
//   The output lane holds 4 consecutive channels of one token -> 8-byte stores to the un-shifted, un-partitioned pixel.
__global__ __launch_bounds__(512) void win_attn_mfma_kernel(WinAttnParams p) {
    constexpr int HD = 32, WS = 8, NT = 64, VP = NT + 8;  // V^T row pitch in halfs (144 B)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;
    f16* vt = (f16*)smem + (size_t)h * HD * VP;  // [HD][VP] of this head
    const int nwx = p.W / WS;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x - wy * nwx;
    const int b = blockIdx.y;
    const int E = p.heads * HD;
    const f16* qkv = (const f16*)p.qkv;
    auto pixel = [&](int t) -> long long {
        int sy = wy * WS + (t >> 3) + p.shift; if (sy >= p.H) sy -= p.H;
        int sx = wx * WS + (t & 7) + p.shift; if (sx >= p.W) sx -= p.W;
        return ((long long)b * p.H + sy) * p.W + sx;
    };
    // V^T staging: lane = token
    {
        const f16* vsrc = qkv + pixel(lane) * p.ldq + 2 * E + h * HD;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const f16x8 v = *(const f16x8*)(vsrc + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(8 * c + e) * VP + lane] = v[e];
        }
    }
    long long pix[4];
    f16x8 kf[4], qf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        pix[f] = pixel(16 * f + lr);
        const f16* src = qkv + pix[f] * p.ldq + h * HD + 8 * lg;
        qf[f] = *(const f16x8*)src;
        kf[f] = *(const f16x8*)(src + E);
    }
    f32x4 s[4][4];  // [fj][fi]
#pragma unroll
    for (int fj = 0; fj < 4; ++fj)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
            s[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[fj], qf[fi], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    // region ids of the (quirky) shift mask: band of window_row*8 + token_column (see win_attn_kernel)
    int rid_i = 0, rid_j[4] = {0, 0, 0, 0};
    if (p.shift > 0) {
        auto band = [&](int c) { const int yq = wy * WS + c; return yq < p.H - WS ? 0 : (yq < p.H - p.shift ? 1 : 2); };
        rid_i = band(lr & 7);
#pragma unroll
        for (int r = 0; r < 4; ++r) rid_j[r] = band(4 * (lg & 1) + r);
    }
    float inv[4];
    const float* bn = p.bias_n + (long long)h * NT * NT;  // [i][j]
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const int i = 16 * fi + lr;
        float m = -3.0e38f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const f32x4 bv = *(const f32x4*)(bn + i * NT + 16 * fj + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(s[fj][fi][r], p.scale, bv[r]);
                if (p.shift > 0 && rid_j[r] != rid_i) v += -100.0f;
                s[fj][fi][r] = v;
                m = fmaxf(m, v);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[fj][fi][r] - m);
                s[fj][fi][r] = e;
                l += e;
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv[fi] = 1.0f / l;
    }
    __syncthreads();  // V^T of every head is in LDS
    f32x4 o[2][4];    // [fd][fi]
#pragma unroll
    for (int fd = 0; fd < 2; ++fd)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) o[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f16x8 va[2], pb[4];
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) {
            const f16* row = vt + (16 * fd + lr) * VP + 32 * ks + 4 * lg;
            const f16x4 lo = *(const f16x4*)row, hi = *(const f16x4*)(row + 16);
            va[fd] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const f32x4 a = s[2 * ks][fi], c = s[2 * ks + 1][fi];
            pb[fi] = f16x8{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
        }
#pragma unroll
        for (int fd = 0; fd < 2; ++fd)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) o[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[fd], pb[fi], o[fd][fi], 0, 0, 0);
    }
    f16* out = (f16*)p.out;
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int fd = 0; fd < 2; ++fd) {
            f16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (f16)(o[fd][fi][r] * inv[fi]);
            *(f16x4*)(out + pix[fi] * p.ldo + h * HD + 16 * fd + 4 * lg) = hv;
        }
}

}  // namespace

namespace {

// Split-storage window attention on the matrix cores (RS_F16S): the structure of win_attn_mfma_kernel with every product
// formed from (hi, lo) fp16 pairs - S^T = Kh Qh^T + 2^-11 (Kh Ql^T + Kl Qh^T), O^T likewise with P split on the fly - so
// the attention core stays fp32-class like the split implicit GEMM around it (3 MFMAs per product, two accumulators).
__global__ __launch_bounds__(512) void win_attn_split_kernel(WinAttnParams p) {
    constexpr int HD = 32, WS = 8, NT = 64, VP = NT + 8;  // V^T row pitch in halfs (144 B)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lg = lane >> 4;
    f16* vth = (f16*)smem + (size_t)h * 2 * HD * VP;  // [HD][VP] hi, then [HD][VP] lo of this head
    f16* vtl = vth + HD * VP;
    const int nwx = p.W / WS;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x - wy * nwx;
    const int b = blockIdx.y;
    const int E = p.heads * HD;
    const f16* qkv = (const f16*)p.qkv;
    const long long rq = 2LL * p.ldq;     // pixel record: [ldq hi | ldq lo]
    auto pixel = [&](int t) -> long long {
        int sy = wy * WS + (t >> 3) + p.shift; if (sy >= p.H) sy -= p.H;
        int sx = wx * WS + (t & 7) + p.shift; if (sx >= p.W) sx -= p.W;
        return ((long long)b * p.H + sy) * p.W + sx;
    };
    {   // V^T staging: lane = token
        const f16* vsrc = qkv + pixel(lane) * rq + 2 * E + h * HD;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const f16x8 vh = *(const f16x8*)(vsrc + 8 * c), vl = *(const f16x8*)(vsrc + p.ldq + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { vth[(8 * c + e) * VP + lane] = vh[e]; vtl[(8 * c + e) * VP + lane] = vl[e]; }
        }
    }
    long long pix[4];
    f16x8 kh[4], kl[4], qh[4], ql[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        pix[f] = pixel(16 * f + lr);
        const f16* src = qkv + pix[f] * rq + h * HD + 8 * lg;
        qh[f] = *(const f16x8*)src;           ql[f] = *(const f16x8*)(src + p.ldq);
        kh[f] = *(const f16x8*)(src + E);     kl[f] = *(const f16x8*)(src + E + p.ldq);
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
    int rid_i = 0, rid_j[4] = {0, 0, 0, 0};
    if (p.shift > 0) {   // region ids of the (quirky) shift mask: band of window_row*8 + token_column (see win_attn_kernel)
        auto band = [&](int c) { const int yq = wy * WS + c; return yq < p.H - WS ? 0 : (yq < p.H - p.shift ? 1 : 2); };
        rid_i = band(lr & 7);
#pragma unroll
        for (int r = 0; r < 4; ++r) rid_j[r] = band(4 * (lg & 1) + r);
    }
    float inv[4];
    const float* bn = p.bias_n + (long long)h * NT * NT;  // [i][j]
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const int i = 16 * fi + lr;
        float m = -3.0e38f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const f32x4 bv = *(const f32x4*)(bn + i * NT + 16 * fj + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(s[fj][fi][r], p.scale, bv[r]);
                if (p.shift > 0 && rid_j[r] != rid_i) v += -100.0f;
                s[fj][fi][r] = v;
                m = fmaxf(m, v);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int fj = 0; fj < 4; ++fj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(s[fj][fi][r] - m);
                s[fj][fi][r] = e;
                l += e;
            }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv[fi] = 1.0f / l;
    }
    __syncthreads();  // V^T of every head is in LDS
    f32x4 om[2][4], oc[2][4];    // [fd][fi] main / cross
#pragma unroll
    for (int fd = 0; fd < 2; ++fd)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) { om[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
        for (int fi = 0; fi < 4; ++fi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16 a, bq;
                rs_split(s[2 * ks][fi][r], a, bq);     pbh[fi][r] = a;     pbl[fi][r] = bq;
                rs_split(s[2 * ks + 1][fi][r], a, bq); pbh[fi][4 + r] = a; pbl[fi][4 + r] = bq;
            }
        }
#pragma unroll
        for (int fd = 0; fd < 2; ++fd)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                om[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[fd], pbh[fi], om[fd][fi], 0, 0, 0);
                oc[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vah[fd], pbl[fi], oc[fd][fi], 0, 0, 0);
                oc[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(val[fd], pbh[fi], oc[fd][fi], 0, 0, 0);
            }
    }
    f16* out = (f16*)p.out;
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
            f16* dst = out + pix[fi] * 2 * p.ldo + h * HD + 16 * fd + 4 * lg;
            *(f16x4*)dst = hv;
            *(f16x4*)(dst + p.ldo) = lv;
        }
}

}  // namespace

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
__device__ __forceinline__ void lds_dma16_na(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr3_t)lds, 16, voff, 0, 0, 0);
}
// per-lane byte offset + wave-uniform byte offset (the instruction's scalar offset operand: no VGPR arithmetic for the uniform part)
__device__ __forceinline__ void lds_dma16_vs(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr3_t)lds, 16, voff, (int)soff, 0, 0);
}

// Window attention with the qkv projection fused in (fp16 storage, E = 32 * heads <= 192, heads = 6 in every shipped config):
// one workgroup per window, one wave per head.  The window's 64 normalised tokens are gathered (roll + partition folded
// into the addressing) into LDS once; wave h multiplies them with ITS 96 rows of the qkv weight (q_h, k_h, v_h: no other
// wave needs them, so they come straight from L2 into registers in MFMA A-operand layout) - 144 MFMAs - and the
// accumulators ARE the attention operands: a lane ends with 8 head-dim values of one token for q and for k, the same 8 for
// both, which is all S^T = K Q^T needs (the contraction index may be permuted consistently).  V goes through the same LDS
// transposition as in win_attn_mfma_kernel, everything after that is identical.  The [M][3E] qkv tensor (151 MB at batch 32
// on the 64x64 level) never exists.
#ifdef RS_SPLIT_ABLATE
__device__ int g_attn_abl = 0;   // timing ablation (RS_ATTN_ABL=1, ablate builds): every weight fragment comes from one cached line
__device__ long long g_attn_clk[16 * 4096];   // phase stamps of wave 0 of the first 4096 workgroups
#define RS_ATTN_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) { const int wg_ = blockIdx.y * gridDim.x + blockIdx.x; if (wg_ < 4096) g_attn_clk[16 * wg_ + (k)] = clock64(); } } while (0)
#else
#define RS_ATTN_STAMP(k)
#endif
// NW windows per workgroup (1 or 2): a wave's q / k / v / proj weight fragments are fetched from L2 once and applied to the tokens
// of NW windows.  The kernel is bound by that weight stream (295 KB per window, 2.4 GB per launch on the 64 x 64 level at batch
// 32 - with every fragment read from one cached line instead it is 29 % faster, profiles/r2_swin_mlp_split_ablation.txt), so two
// windows per workgroup halve its dominant traffic; the attention itself runs window after window on the same registers.
template <int NW>
__global__ __launch_bounds__(384) void win_attn_qkv_kernel(WinAttnParams p, unsigned x_bytes, unsigned res_bytes) {
    constexpr int HD = 32, WS = 8, NT = 64, VP = NT + 8, E = 192, KS = E / 32;
    constexpr int XS_STAGE = NT * 128;                       // 64 token rows x 128 B per 64-wide K stage
    constexpr int XS_WIN = 3 * XS_STAGE;                     // token tile of one window
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int nwx = p.W / WS;
    const int b = blockIdx.y;
    RS_ATTN_STAMP(0);
    auto vt_of = [&](int w) { return (f16*)(smem + NW * XS_WIN) + (size_t)(w * 6 + h) * HD * VP; };   // [HD][VP] of (window, head)
    // The relative position bias of (query i, key j) depends on (y_i - y_j, x_i - x_j) only: 225 values per head
    // (swin_transformer.py:93-102).  They are copied once per workgroup from the dense [h][i][j] table into LDS (5.4 KB) and the softmax
    // reads them from there with constant offsets - instead of 16 KB per head and window through L2 (a third of this kernel's traffic).
    constexpr int BT_BYTES = 6144;                           // [6 heads][256 floats]: 225 values per head, zero padded (WinAttnParams::bias_c)
    float* const btab = (float*)(smem + NW * (XS_WIN + 6 * HD * VP * 2));
    // residual / output tile of window w (fused projection only): the shortcut rows arrive here by LDS-DMA (token tile format), the
    // projection adds its result in place and the finished tile leaves as whole 128-byte lines
    auto rt_of = [&](int w) { return smem + NW * (XS_WIN + 6 * HD * VP * 2) + BT_BYTES + w * XS_WIN; };
    // (round 6: the table arrives compact and in units of log2 - the softmax below works in base 2 - by ONE LDS-DMA instruction per wave, in front
    // of the token requests; the gather loop of dependent loads it replaces stood in front of them, see win_attn_split.hip)
    {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias_c, 0, 6 * 1024, 0x00020000);
        lds_dma16_vs(rb, (char*)btab + h * 1024, (unsigned)(h * 1024 + lane * 16), 0u);
    }
    auto win_y = [&](int w) { return (int)(blockIdx.x * NW + w) / nwx; };
    auto pixel = [&](int w, int t) -> long long {
        const int wi = blockIdx.x * NW + w;
        const int wy = wi / nwx, wx = wi - wy * nwx;
        int sy = wy * WS + (t >> 3) + p.shift; if (sy >= p.H) sy -= p.H;
        int sx = wx * WS + (t & 7) + p.shift; if (sx >= p.W) sx -= p.W;
        return ((long long)b * p.H + sy) * p.W + sx;
    };
    // ---- tokens of the windows -> LDS: per window 8 row groups x 3 K stages = 24 LDS-DMA instructions, 4 per wave.  A piece is 8 tokens of
    // one window row x 64 channels: the row (and the stage) is wave-uniform -> scalar offset; the column and the chunk are the lane's
    auto dma_tile = [&](const __amdgpu_buffer_rsrc_t r, int ld, char* dst, int w) {
        const int wi = blockIdx.x * NW + w, wy = wi / nwx, wx = wi - wy * nwx;
        const int rsub = lane >> 3, kcp = (lane & 7) ^ (rsub & 7);
        int sx = wx * WS + rsub + p.shift; if (sx >= p.W) sx -= p.W;
        const unsigned voff = (unsigned)(sx * ld + kcp * 8) * 2u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = h * 4 + q, st = it >> 3, grp = it & 7;
            int sy = wy * WS + grp + p.shift; if (sy >= p.H) sy -= p.H;
            lds_dma16_vs(r, dst + st * XS_STAGE + (grp * 8) * 128, voff, (unsigned)(((b * p.H + sy) * p.W) * ld + st * 64) * 2u);
        }
    };
    {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
#pragma unroll
        for (int w = 0; w < NW; ++w) dma_tile(rx, p.ldx, smem + w * XS_WIN, w);
    }
    const f16* wq = (const f16*)p.wqkv;
    const int swz[2] = {(lg ^ (lr & 7)) << 4, ((4 + lg) ^ (lr & 7)) << 4};
    // this head's 32 output features starting at weight row n0: fragments straight from L2 in MFMA A-operand layout.  The weights come
    // in FRAGMENT-MAJOR order (engine.hip ConvW::wh_frag: [16-row block][k step][lane] x 16 B): one contiguous 1 KB per wave instruction
    // (8 cache lines) where the row-major weight costs 16 lines of which half the bytes are used
    auto load_w = [&](const f16* wsrc, int n0, f16x8 (&wf)[2][KS]) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#ifdef RS_SPLIT_ABLATE
                if (g_attn_abl) { wf[f][ks] = *(const f16x8*)(wsrc + lg * 8); continue; }
#endif
                wf[f][ks] = *(const f16x8*)(wsrc + ((long long)((n0 >> 4) + f) * KS + ks) * 512 + lane * 8);   // fragment-major: 1 KB per wave
            }
    };
    // one projection pass over the token tile of window w -> acc[2 feature frags][4 token frags] (bias in the accumulator)
    auto project = [&](int w, const f16x8 (&wf)[2][KS], const float* bias, int n0, f32x4 (&acc)[2][4]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const f32x4 bv = *(const f32x4*)(bias + n0 + 16 * f + 4 * lg);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) acc[f][fi] = bv;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 xb[4];
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) xb[fi] = *(const f16x8*)(smem + w * XS_WIN + (ks >> 1) * XS_STAGE + (16 * fi + lr) * 128 + swz[ks & 1]);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) acc[f][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f][ks], xb[fi], acc[f][fi], 0, 0, 0);
        }
    };
    auto pack = [&](const f32x4 (&acc)[2][4], f16x8 (&out)[4]) {
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const f32x4 a = acc[0][fi], c = acc[1][fi];
            out[fi] = f16x8{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the windows' tokens are in LDS
    RS_ATTN_STAMP(1);
    if (p.xcoef) {
        // GroupNorm (norm1) folded in: x * scale[b][c] + shift[b][c], rounded to fp16 exactly where the separate apply kernel
        // rounds.  Per window 64 rows x 24 chunks of 8 channels, 4 chunks per thread; LDS position ps of row t holds chunk ps ^ (t & 7).
        const float* sc = p.xcoef + (long long)b * 2 * E;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = tid + 384 * q;              // 0 .. 1535
                const int st = item >> 9, t = (item >> 3) & 63, ps = item & 7;
                const int c0 = st * 64 + ((ps ^ (t & 7)) << 3);
                f16x8* cell = (f16x8*)(smem + w * XS_WIN + st * XS_STAGE + t * 128 + ps * 16);
                f16x8 v = *cell;
                const f32x4 a0 = *(const f32x4*)(sc + c0), a1 = *(const f32x4*)(sc + c0 + 4);
                const f32x4 d0 = *(const f32x4*)(sc + E + c0), d1 = *(const f32x4*)(sc + E + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (f16)fmaf((float)v[e], a0[e], d0[e]); v[4 + e] = (f16)fmaf((float)v[4 + e], a1[e], d1[e]); }
                *cell = v;
            }
        __syncthreads();
    }
    RS_ATTN_STAMP(2);
    f16x8 kf[NW][4], qf[NW][4];
    {
        f16x8 wf[2][KS];
        f32x4 acc[2][4];
        // (scheduling fences between the passes: two windows' passes interleaved by the compiler do not fit 256 registers)
        load_w(wq, h * HD, wf);            // q_h: lane (lr,lg) holds d = {4lg+r, 16+4lg+r} of token 16fi+lr
#pragma unroll
        for (int w = 0; w < NW; ++w) { project(w, wf, p.bqkv, h * HD, acc); pack(acc, qf[w]); __builtin_amdgcn_sched_barrier(0); }
        RS_ATTN_STAMP(3);
        load_w(wq, E + h * HD, wf);        // k_h: the same d set per lane -> a consistent contraction order for S^T
#pragma unroll
        for (int w = 0; w < NW; ++w) { project(w, wf, p.bqkv, E + h * HD, acc); pack(acc, kf[w]); __builtin_amdgcn_sched_barrier(0); }
        RS_ATTN_STAMP(4);
        load_w(wq, 2 * E + h * HD, wf);    // v_h -> V^T[d][token] in LDS
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            __builtin_amdgcn_sched_barrier(0);
            project(w, wf, p.bqkv, 2 * E + h * HD, acc);
            f16* vt = vt_of(w);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[(16 * f + 4 * lg + r) * VP + 16 * ft + lr] = (f16)acc[f][ft][r];
        }
    }
    if (p.wproj && p.res) {
        // the shortcut's rows -> the residual / output tiles by LDS-DMA, requested HERE: memory operations complete in order, so the next
        // wait (the projection weights, behind the attention) is the first that includes them - they travel during the whole attention
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, res_bytes, 0x00020000);
#pragma unroll
        for (int w = 0; w < NW; ++w) dma_tile(rr, p.ldres, rt_of(w), w);
    }
    RS_ATTN_STAMP(5);
    __syncthreads();  // V^T of every head is in LDS; every wave is done with the token tiles (they are overwritten below)
    RS_ATTN_STAMP(6);
    // bias of (i = 16 fi + lr, j = 16 fj + 4 lg + r) = tb[30 (fi - fj) - r]
    const float* tb = btab + h * 256 + ((lr >> 3) - (lg >> 1) + 7) * 15 + (lr & 7) - 4 * (lg & 1) + 7;
    f16* out = (f16*)p.out;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        __builtin_amdgcn_sched_barrier(0);   // window after window: interleaving two windows' score tiles would need > 256 registers
        f32x4 s[4][4];  // [fj][fi]
#pragma unroll
        for (int fj = 0; fj < 4; ++fj)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
                s[fj][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[w][fj], qf[w][fi], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // softmax over the keys (per query column i = 16 fi + lr) in base 2: the scores arrive as s * (scale log2 e) + (bias log2 e) (the
        // table is stored pre-multiplied), so the exponential is the bare v_exp_f32.  The row sums come out of the P V matrix product (a
        // row of ones appended to V^T, below).  Only windows of the LAST window row can carry the shift mask (the mask regions are bands
        // of window_row * 8 + token_column, see win_attn_kernel): every other window takes the mask-free path (a wave-uniform branch).
        const float c2 = p.scale * 1.44269504088896f;
        const bool masked = p.shift > 0 && win_y(w) == p.H / WS - 1;
        auto softmax = [&](auto MK) {
            constexpr bool MASK = decltype(MK)::value;
            int rid_i = 0, rid_j[4] = {0, 0, 0, 0};
            if constexpr (MASK) {
                const int wy = win_y(w);
                auto band = [&](int c) { const int yq = wy * WS + c; return yq < p.H - WS ? 0 : (yq < p.H - p.shift ? 1 : 2); };
                rid_i = band(lr & 7);
#pragma unroll
                for (int r = 0; r < 4; ++r) rid_j[r] = band(4 * (lg & 1) + r);
            }
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
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
        if (masked) softmax(std::true_type{}); else softmax(std::false_type{});
        f32x4 o[2][4], ol[4];    // [fd][fi]; ol: the row of ones -> sum over the keys of the fp16 probabilities, for every lane of column i
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            ol[fi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int fd = 0; fd < 2; ++fd) o[fd][fi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f16x8 ones = f16x8{(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
        const f16* vt = vt_of(w);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 va[2], pb[4];
#pragma unroll
            for (int fd = 0; fd < 2; ++fd) {
                const f16* row = vt + (16 * fd + lr) * VP + 32 * ks + 4 * lg;
                const f16x4 lo = *(const f16x4*)row, hi = *(const f16x4*)(row + 16);
                va[fd] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const f32x4 a = s[2 * ks][fi], c = s[2 * ks + 1][fi];
                pb[fi] = f16x8{(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
            }
#pragma unroll
            for (int fd = 0; fd < 2; ++fd)
#pragma unroll
                for (int fi = 0; fi < 4; ++fi) o[fd][fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[fd], pb[fi], o[fd][fi], 0, 0, 0);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) ol[fi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pb[fi], ol[fi], 0, 0, 0);
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const float inv = 1.0f / ol[fi][0];
#pragma unroll
            for (int fd = 0; fd < 2; ++fd) {
                f16x4 hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (f16)(o[fd][fi][r] * inv);
                const int t = 16 * fi + lr, c = h * HD + 16 * fd + 4 * lg;     // token row, feature
                if (!p.wproj) *(f16x4*)(out + pixel(w, t) * p.ldo + c) = hv;
                // fused output projection: the heads' results meet in LDS (the token tile's space, same row / swizzle format)
                else *(f16x4*)(smem + w * XS_WIN + (c >> 6) * XS_STAGE + t * 128 + ((((c & 63) >> 3) ^ (t & 7)) << 4) + (c & 7) * 2) = hv;
            }
        }
        RS_ATTN_STAMP(7 + w);
    }
    if (!p.wproj) return;
    // ---- fused output projection: wave h produces output features 32h .. 32h+31 for all tokens: 48 MFMAs per window, weights
    // straight from L2 like the qkv passes
    f16x8 wf2[2][KS];
    load_w((const f16*)p.wproj, h * HD, wf2);
    const bool has_res = p.res != nullptr;
    __syncthreads();   // all heads' attention results are in LDS
    RS_ATTN_STAMP(9);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc2[2][4];
        project(w, wf2, p.bproj, h * HD, acc2);
        float s1[2][4], s2[2][4];   // per-channel sums of the STORED values over this lane's four tokens
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[f][r] = 0.f; s2[f][r] = 0.f; }
        char* const rt = rt_of(w);
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                // cell of (token t, features c .. c + 3) in the residual / output tile: read and rewritten by this lane only
                const int t = 16 * fi + lr, c = h * HD + 16 * f + 4 * lg;
                f16x4* cell = (f16x4*)(rt + (c >> 6) * XS_STAGE + t * 128 + ((((c & 63) >> 3) ^ (t & 7)) << 4) + (c & 7) * 2);
                f16x4 hv, rv = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                if (has_res) rv = *cell;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hv[r] = (f16)(acc2[f][fi][r] + (float)rv[r]);
                    if (p.ystats) { const float sv = (float)hv[r]; s1[f][r] += sv; s2[f][r] = fmaf(sv, sv, s2[f][r]); }
                }
                *cell = hv;
            }
        if (p.ystats) {   // statistics for norm2: the wave holds its 32 features of all 64 tokens of the window
            const int wi = blockIdx.x * NW + w;
            float* dst = p.ystats + (((long long)b * (nwx * (p.H / WS)) + wi) * p.ystats_ld + h * HD) * 2;
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = rs_sum16(s1[f][r]), q = rs_sum16(s2[f][r]);   // (DPP adds: same bits as the xor-shuffle butterfly)
                    if (lr == 0) { dst[(16 * f + 4 * lg + r) * 2] = a; dst[(16 * f + 4 * lg + r) * 2 + 1] = q; }
                }
        }
    }
    __syncthreads();   // the output tiles are complete
    // whole rows out: per window 24 pieces of 8 token rows x 128 B (8 full cache lines per wave instruction), 4 per wave
    {
        const int rsub = lane >> 3, ps = lane & 7, chunk = ps ^ (rsub & 7);
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int it = h * 4 + q, st = it >> 3, grp = it & 7;
                const f16x8 v = *(const f16x8*)(rt_of(w) + st * XS_STAGE + (grp * 8 + rsub) * 128 + ps * 16);
                *(f16x8*)(out + pixel(w, grp * 8 + rsub) * p.ldo + st * 64 + chunk * 8) = v;
            }
    }
    RS_ATTN_STAMP(10);
}


}  // namespace

#ifdef RS_SPLIT_ABLATE
// ablate builds: mean cycles between consecutive stamps (0 .. nst-1) of wave 0 over the first `nwg` workgroups of the last launch
extern "C" int rs_attn_phase_cycles(int nwg, int nst, double* out) {
    static long long h[16 * 4096];
    if (nwg < 1 || nwg > 4096 || nst < 2 || nst > 16) return -1;
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_clk), sizeof(long long) * 16 * nwg) != hipSuccess) return -1;
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

// fused qkv projection + window attention (fp16, 6 heads of 32): x, wqkv, bqkv, ldx of the parameter block are used, qkv is not
extern "C" int rs_win_attn_qkv_supported(int heads, int E) { return heads == 6 && E == 192; }
extern "C" int rs_win_attn_qkv_launch(const WinAttnParams* pp, hipStream_t st) {
    const WinAttnParams& p = *pp;
    if ((p.H % 8) || (p.W % 8) || !rs_win_attn_qkv_supported(p.heads, 32 * p.heads) || (p.ldx % 8) || (p.ldo % 8) || !p.bias_c) return -2;
    if (p.shift != 0 && p.shift != 4) return -2;
    if (p.wproj && (!p.bproj || (p.res && (p.ldres % 4)))) return -2;
    const size_t xb = (size_t)p.B * p.H * p.W * p.ldx * 2;
    if (xb >= 0xF0000000ull) return -2;
    // two windows per workgroup wherever an image has an even number of them (RS_ATTN_NW=1: one, for A/B runs)
    static const int nw_max = []() { const char* v = getenv("RS_ATTN_NW"); return v ? atoi(v) : 2; }();
    const int nwin = (p.H / 8) * (p.W / 8);
    const int NW = (nw_max >= 2 && nwin % 2 == 0) ? 2 : 1;
    const size_t rb = p.res ? (size_t)p.B * p.H * p.W * p.ldres * 2 : 0;
    if (rb >= 0xF0000000ull || (p.wproj && p.res && (p.ldres % 8))) return -2;
    // token tiles + V^T + the bias table (+ with the fused projection one residual / output tile per window)
    const size_t lds = (size_t)NW * (3 * 64 * 128 + (size_t)p.heads * 32 * (64 + 8) * sizeof(f16)) + 6144 + (p.wproj ? (size_t)NW * 3 * 64 * 128 : 0);
#ifdef RS_SPLIT_ABLATE
    {
        static const int abl = []() { const char* v = getenv("RS_ATTN_ABL"); const int a = v ? atoi(v) : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_abl), &a, sizeof(int)); return a; }();
        (void)abl;
    }
#endif
    {   // dynamic LDS above 64 KB needs the attribute: set once per device to the largest layout (fused projection)
        static RsAttrFlags attr_flags;
        if (attr_flags.need()) {
            const int per_win = 3 * 64 * 128 + 6 * 32 * (64 + 8) * 2 + 3 * 64 * 128;
            (void)hipFuncSetAttribute((const void*)win_attn_qkv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * per_win + 6144);
            (void)hipFuncSetAttribute((const void*)win_attn_qkv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, per_win + 6144);
        }
    }
    if (NW == 2) hipLaunchKernelGGL(win_attn_qkv_kernel<2>, dim3(nwin / 2, p.B), dim3(64 * p.heads), lds, st, p, (unsigned)xb, (unsigned)rb);
    else hipLaunchKernelGGL(win_attn_qkv_kernel<1>, dim3(nwin, p.B), dim3(64 * p.heads), lds, st, p, (unsigned)xb, (unsigned)rb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int rs_win_attn_launch(const WinAttnParams* pp, int dt, hipStream_t st) {
    const WinAttnParams& p = *pp;
    if ((p.H % 8) || (p.W % 8) || p.heads < 1 || p.heads > 8 || (p.ldq % 8) || (p.ldo % 8)) return -2;
    if (p.shift != 0 && p.shift != 4) return -2;
    dim3 grid((p.H / 8) * (p.W / 8), p.B), block(64 * p.heads);
    if (dt == RS_F16 && p.bias_n) {
        const size_t lds_m = (size_t)p.heads * 32 * (64 + 8) * sizeof(f16);
        hipLaunchKernelGGL(win_attn_mfma_kernel, grid, block, lds_m, st, p);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const size_t lds = (size_t)2 * p.heads * 64 * 32 * sizeof(float) + 64 * sizeof(int);
    if (dt == RS_F16) {
        (void)hipFuncSetAttribute((const void*)win_attn_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((win_attn_kernel<f16>), grid, block, lds, st, p);
    } else if (dt == RS_F16S) {
        static const bool valu = []() { const char* e = getenv("RS_ATTN_SPLIT_VALU"); return e && e[0] == '1'; }();   // A/B knob: fp32 VALU kernel
        if (p.bias_n && !valu) {
            const size_t lds_m = (size_t)p.heads * 2 * 32 * (64 + 8) * sizeof(f16);   // 72 KB at 8 heads: over the 64 KB default
            (void)hipFuncSetAttribute((const void*)win_attn_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(win_attn_split_kernel, grid, block, lds_m, st, p);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
        (void)hipFuncSetAttribute((const void*)win_attn_kernel<h2s>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((win_attn_kernel<h2s>), grid, block, lds, st, p);
    } else {
        (void)hipFuncSetAttribute((const void*)win_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((win_attn_kernel<float>), grid, block, lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------- row softmax (fp32 in)
namespace {

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = is_max ? fmaxf(r, sh[k]) : r + sh[k];
    return r;
}

template <typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, TO* out, int ncols, long long lds_, long long ldo) {
    __shared__ float sh[4];
    const float* row = s + (long long)blockIdx.x * lds_;
    TO* orow = out + (long long)blockIdx.x * ldo * Store<TO>::PM;
    float mx = -3.0e38f;
    for (int c = threadIdx.x * 4; c < ncols; c += 1024) {
        const f32x4 v = *(const f32x4*)(row + c);
        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    mx = block_reduce(mx, sh, true);
    float sum = 0.f;
    for (int c = threadIdx.x * 4; c < ncols; c += 1024) {
        const f32x4 v = *(const f32x4*)(row + c);
        sum += expf(v[0] - mx) + expf(v[1] - mx) + expf(v[2] - mx) + expf(v[3] - mx);
    }
    sum = block_reduce(sum, sh, false);
    const float inv = 1.0f / sum;
    for (int c = threadIdx.x * 4; c < ncols; c += 1024) {
        const f32x4 v = *(const f32x4*)(row + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) rs_st<TO>(orow + c + r, (int)ldo, expf(v[r] - mx) * inv);
    }
}

}  // namespace

extern "C" int rs_softmax_rows_launch(const float* s, void* out, int out_dt, long long nrows, int ncols, long long lds_, long long ldo,
                                      hipStream_t st) {
    if (ncols % 4) return -2;
    if (out_dt == RS_F16)
        hipLaunchKernelGGL((softmax_rows_kernel<f16>), dim3((unsigned)nrows), dim3(256), 0, st, s, (f16*)out, ncols, lds_, ldo);
    else if (out_dt == RS_F16S)
        hipLaunchKernelGGL((softmax_rows_kernel<h2s>), dim3((unsigned)nrows), dim3(256), 0, st, s, (h2s*)out, ncols, lds_, ldo);
    else
        hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3((unsigned)nrows), dim3(256), 0, st, s, (float*)out, ncols, lds_, ldo);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
