// GroupNorm coefficients without a launch of their own ("tail"): the kernel that produces the LAST piece of a GroupNorm's statistics
// also turns them into the per-(image, channel) affine [B][2][C] that the consuming kernel applies while it loads its input
// (igemm4_kernel.h, the fused Swin kernels).  Every workgroup that contributes statistics of image b publishes its partial sums and draws
// a ticket; the workgroup that draws the last ticket of the image reduces ALL partials in a FIXED order - the same order, the same
// expressions as the stand-alone coefficient kernel (gn_apply_kernel in its coefficient-only mode), so the two paths agree bit for bit -
// and writes the coefficients.  What disappears: one 5 - 10 us launch and one dependent kernel boundary per GroupNorm (645
// gn_fused / coefficient launches per pass).
//
// Reference arithmetic: GroupNorm32 = F.group_norm(x.float(), 32, w, b, eps) (models/basic_ops.py:15-17,89-96; eps 1e-5;
// ldm/modules/diffusionmodules/model.py:46-47: eps 1e-6), FiLM y = norm(h) * (1 + scale) + shift (models/unet.py:198-202).
//
// Inter-workgroup visibility follows /opt/skills/guides/cdna_hip_programming.md §6 Guideline 16, recipe R1 (the per-XCD L2s are not
// coherent with each other and a CU's L1 is never refreshed by other CUs' stores):
//   producer : payload = 8-byte WRITE-THROUGH stores (relaxed agent-scope __hip_atomic_store -> `global_store_dwordx2 ... sc1`) by ONE
//              wave, which drains them (`s_waitcnt vmcnt(0)`) before its lane 0 draws the ticket with a relaxed agent-scope fetch_add;
//   consumer : (the last arriver) the SAME lane issues ONE agent-scope acquire fence behind the ticket, then the same wave reads the
//              partials - with sc1 loads on top of the acquire (both forms are valid on their own for sc1 payloads).
// Tickets live in a pool that the engine zeroes with ONE hipMemsetAsync at the start of every call (never reset in-kernel: a poisoned
// word from an aborted launch cannot survive into the next call); every tail of a call owns its own B words of the pool.
#pragma once
#include "common.h"

// (struct GNTail: common.h, next to the launch parameter blocks that carry it)

// ---- the arithmetic shared by every path that turns sums into coefficients (explicit fmaf: no contraction freedom, so that the
// stand-alone kernel and the tails agree bit for bit)
__device__ __forceinline__ void rs_gn_group(float a, float q, float n, float eps, float& mean, float& rstd) {
    mean = a / n;
    const float var = fmaxf(fmaf(-mean, mean, q / n), 0.f);
    rstd = 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ void rs_gn_channel(float gamma, float beta, float mean, float rstd, const float* film, int c, int C, float& a, float& b) {
    a = gamma * rstd;
    b = fmaf(-mean, a, beta);
    if (film) {
        const float sc = 1.0f + film[c];
        a *= sc;
        b = fmaf(b, sc, film[C + c]);
    }
}

// 8-byte write-through store / L1-bypassing load of one (sum, sum of squares) pair
__device__ __forceinline__ void rs_pub_pair(float* dst, float a, float q) {
    const unsigned long long bits = ((unsigned long long)__float_as_uint(q) << 32) | (unsigned long long)__float_as_uint(a);
    __hip_atomic_store((unsigned long long*)dst, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 rs_get_pair(const float* src) {
    const unsigned long long bits = __hip_atomic_load((const unsigned long long*)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return float2{__uint_as_float((unsigned)(bits & 0xffffffffull)), __uint_as_float((unsigned)(bits >> 32))};
}

// ARRIVE - by ONE wave of a contributing workgroup (all 64 lanes; the wave that stored ALL of the workgroup's partial sums of image `img`
// with rs_pub_pair): the other waves never wait for the write-through round trip (1 - 3 us under load), it hides behind their output
// stores.  Returns (wave-uniform) whether this workgroup drew the image's last ticket; the caller parks that in an LDS word for FINISH.
__device__ __forceinline__ bool rs_gn_tail_arrive(const GNTail& t, int img) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have reached memory
    unsigned last = 0u;
    if ((threadIdx.x & 63) == 0) {
        const unsigned old = __hip_atomic_fetch_add(t.ticket + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old + 1u == (unsigned)t.expected ? 1u : 0u;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (covers the CU's L1: every wave of the workgroup, behind a barrier)
    }
    return __builtin_amdgcn_readfirstlane((int)last) != 0;
}

// FINISH - by ALL NT threads of the image's last workgroup, behind a workgroup barrier at the very end of the producing kernel (the LDS is
// free: `lds` needs 2 * C + 2 * groups floats): partials -> per-channel totals -> group statistics -> coefficients.  One thread per channel,
// its S partials requested in batches of 16 before the first is added (a loop that waits for each load in turn is S dependent round trips
// to memory at the tail of the launch - measured: it cost more than the coefficient launch the tail replaces); the additions keep the
// order s = 0, 1, ... of gn_apply_kernel's coefficient mode, so the two paths agree bit for bit (a padded slot adds +0.0f).
template <int NT>
__device__ __forceinline__ void rs_gn_tail_finish(const GNTail& t, int img, float* lds) {
    const int tid = threadIdx.x, C = t.C, cpg = C / t.groups;
    float* chs = lds;                 // [C][2] per-channel totals (group-partial mode: [slices][groups][2])
    float* gm = lds + 2 * C;          // [groups] mean
    float* gr = gm + t.groups;        // [groups] 1 / sqrt(var + eps)
    const float n = (float)cpg * (float)t.HW;
    constexpr int NB = 16;
    if (t.stg) {
        // per-group partials of a statistics pass: the reduction tree of gn_apply_kernel (256 / groups slices, then a fixed-order sum)
        const int nsl = 256 / t.groups;
        float* ps = chs;
        for (int item = tid; item < nsl * t.groups; item += NT) {
            const int g = item % t.groups, sl = item / t.groups;
            const float* in = t.stg + (((long long)img * t.Sg) * t.groups + g) * 2;
            float a = 0.f, q = 0.f;
            for (int s0 = sl; s0 < t.Sg; s0 += NB * nsl) {
                float2 v[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) { const int s = s0 + u * nsl; v[u] = s < t.Sg ? rs_get_pair(in + (long long)s * t.groups * 2) : float2{0.f, 0.f}; }
#pragma unroll
                for (int u = 0; u < NB; ++u) { a += v[u].x; q += v[u].y; }
            }
            ps[(sl * t.groups + g) * 2] = a; ps[(sl * t.groups + g) * 2 + 1] = q;
        }
        __syncthreads();
        if (tid < t.groups) {
            float a = 0.f, q = 0.f;
            for (int sl = 0; sl < nsl; ++sl) { a += ps[(sl * t.groups + tid) * 2]; q += ps[(sl * t.groups + tid) * 2 + 1]; }
            rs_gn_group(a, q, n, t.eps, gm[tid], gr[tid]);
        }
    } else {
        for (int c = tid; c < C; c += NT) {
            const bool s0_ = c < t.n0;
            const float* in = s0_ ? t.st0 + (((long long)img * t.S0) * t.ld0 + c) * 2 : t.st1 + (((long long)img * t.S1) * t.ld1 + (c - t.n0)) * 2;
            const int S = s0_ ? t.S0 : t.S1;
            const long long step = 2ll * (s0_ ? t.ld0 : t.ld1);
            float a = 0.f, q = 0.f;
            for (int s0 = 0; s0 < S; s0 += NB) {
                float2 v[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) v[u] = s0 + u < S ? rs_get_pair(in + (s0 + u) * step) : float2{0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NB; ++u) { a += v[u].x; q += v[u].y; }
            }
            chs[2 * c] = a; chs[2 * c + 1] = q;
        }
        __syncthreads();
        if (tid < t.groups) {
            float a = 0.f, q = 0.f;
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += chs[2 * c]; q += chs[2 * c + 1]; }
            rs_gn_group(a, q, n, t.eps, gm[tid], gr[tid]);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
        const int g = c / cpg;
        float a, b;
        rs_gn_channel(t.gamma[c], t.beta[c], gm[g], gr[g], t.film, c, C, a, b);
        t.coef[((long long)img * 2) * C + c] = a;
        t.coef[((long long)img * 2 + 1) * C + c] = b;
    }
}
