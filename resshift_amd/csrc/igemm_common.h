// Device helpers shared by the implicit-GEMM kernels (igemm.hip: register-staged double buffer + split-K;
// igemm2.hip: direct-to-LDS 3-stage ring).  See igemm.hip for the formulation.
#pragma once
#include "common.h"

namespace igemm_detail {

template <typename T> struct MfmaOps;

template <> struct MfmaOps<f16> {
    static constexpr int CH = 8;  // elements per 16-byte chunk
    // one K stage = 8 chunks = 64 halfs = 2 MFMA k-steps of 32
    template <int FC, int FP>
    static __device__ __forceinline__ void stage(const char* ws, const char* xs, int wrow0, int xrow0, int lr, int lg,
                                                 f32x4 (&acc)[FC][FP]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + lg;
            f16x8 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int r = wrow0 + i * 16 + lr;
                a[i] = *(const f16x8*)(ws + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int r = xrow0 + j * 16 + lr;
                b[j] = *(const f16x8*)(xs + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <> struct MfmaOps<float> {
    static constexpr int CH = 4;
    // one K stage = 8 chunks = 32 floats.  Lane group lg reads chunk ss*4+lg (4 floats) and feeds
    // element s to MFMA step s: the k permutation is identical for both operands, so the sum is exact.
    template <int FC, int FP>
    static __device__ __forceinline__ void stage(const char* ws, const char* xs, int wrow0, int xrow0, int lr, int lg,
                                                 f32x4 (&acc)[FC][FP]) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int chunk = ss * 4 + lg;
            f32x4 a[FC], b[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int r = wrow0 + i * 16 + lr;
                a[i] = *(const f32x4*)(ws + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int r = xrow0 + j * 16 + lr;
                b[j] = *(const f32x4*)(xs + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename TO> struct Out4;   // 4 consecutive output elements (`lo`: see h2s in common.h; ignored by the plain types)
template <> struct Out4<h2s> {
    static __device__ __forceinline__ void load(const h2s* p, float (&v)[4], int lo) {
        const f16x4 h = *(const f16x4*)p, l = *(const f16x4*)((const f16*)p + lo);
        v[0] = rs_join(h[0], l[0]); v[1] = rs_join(h[1], l[1]); v[2] = rs_join(h[2], l[2]); v[3] = rs_join(h[3], l[3]);
    }
    static __device__ __forceinline__ void store(h2s* p, const float (&v)[4], int lo) {
        f16x4 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) { f16 a, b; rs_split(v[r], a, b); h[r] = a; l[r] = b; }
        *(f16x4*)p = h; *(f16x4*)((f16*)p + lo) = l;
    }
};
template <> struct Out4<f16> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[4], int lo = 0) {
        f16x4 t = *(const f16x4*)p;
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[4], int lo = 0) {
        f16x4 t; t[0] = (f16)v[0]; t[1] = (f16)v[1]; t[2] = (f16)v[2]; t[3] = (f16)v[3];
        *(f16x4*)p = t;
    }
};
template <> struct Out4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4], int lo = 0) {
        f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4], int lo = 0) {
        f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        *(f32x4*)p = t;
    }
};

// fp16-storage outputs use the v_rcp / v_exp formulations of common.h; the fp32 path keeps libm's erff and IEEE division
template <typename TO> __device__ __forceinline__ float epi_act(float x, int act) {
    if (act == RS_ACT_GELU) return Store<TO>::FAST ? rs_gelu_fast(x) : rs_gelu(x);
    if (act == RS_ACT_SILU) return Store<TO>::FAST ? rs_silu_fast(x) : rs_silu(x);
    return x;
}

// XCD-aware bijective remap of the linear workgroup id: hardware places id % 8 on XCD (id % 8); give every XCD a
// contiguous run of tile indices so that neighbouring tiles (which share operand panels) share an L2.
__device__ __forceinline__ int xcd_remap(int id, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}


}  // namespace igemm_detail
