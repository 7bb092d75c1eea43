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
//   producer : payload = 8-byte WRITE-THROUGH stores (relaxed agent-scope __hip_atomic_store -> `global_store_dwordx2 ... sc1`), every
//              storing wave drains them (`s_waitcnt vmcnt(0)`), __syncthreads(), ONE lane draws the ticket with a relaxed agent-scope
//              fetch_add;
//   consumer : (the last arriver) the SAME lane issues ONE agent-scope acquire fence behind the ticket, __syncthreads(), then every
//              wave reads the partials - with sc1 loads on top of the acquire (both forms are valid on their own for sc1 payloads).
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

// Called by EVERY thread of a contributing workgroup once its partial sums of image `img` have been stored with rs_pub_pair.  `flag`:
// one LDS word nobody else uses at that moment.  True in exactly one workgroup per image: the one that arrived last.
__device__ __forceinline__ bool rs_gn_tail_arrive(const GNTail& t, int img, unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(t.ticket + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + 1u == (unsigned)t.expected;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *flag != 0u;
}

// The last arriver's work: partials -> per-channel totals -> group statistics -> coefficients.  `lds`: 2 * C + 2 * groups floats of LDS
// that are free now (called at the very end of the producing kernel).  NT = threads of the workgroup.
template <int NT>
__device__ __forceinline__ void rs_gn_tail_finish(const GNTail& t, int img, float* lds) {
    const int tid = threadIdx.x, C = t.C, cpg = C / t.groups;
    float* chs = lds;                 // [C][2] per-channel totals, later scale row | shift row
    float* gm = lds + 2 * C;          // [groups] mean
    float* gr = gm + t.groups;        // [groups] 1 / sqrt(var + eps)
    const float n = (float)cpg * (float)t.HW;
    if (t.stg) {
        // per-group partials of a statistics pass: the reduction tree of gn_apply_kernel (256 / groups slices, then a fixed-order sum)
        const int nsl = 256 / t.groups;
        float* ps = chs;              // [nsl][groups][2]
        if (tid < nsl * t.groups) {
            const int g = tid % t.groups, sl = tid / t.groups;
            float a = 0.f, q = 0.f;
            for (int s = sl; s < t.Sg; s += nsl) {
                const float2 v = rs_get_pair(t.stg + (((long long)img * t.Sg + s) * t.groups + g) * 2);
                a += v.x; q += v.y;
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
            const bool s0 = c < t.n0;
            const float* in = s0 ? t.st0 + (((long long)img * t.S0) * t.ld0 + c) * 2 : t.st1 + (((long long)img * t.S1) * t.ld1 + (c - t.n0)) * 2;
            const int S = s0 ? t.S0 : t.S1;
            const long long step = 2ll * (s0 ? t.ld0 : t.ld1);
            float a = 0.f, q = 0.f;
            for (int s = 0; s < S; ++s) { const float2 v = rs_get_pair(in + s * step); a += v.x; q += v.y; }
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
