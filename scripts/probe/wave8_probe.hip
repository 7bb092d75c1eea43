// Micro-probe for the 8-wave (2 per SIMD) igemm4 inner loop with TWO fragment register sets: per half stage 9 ds_read_b128
// (5 weight + 4 pixel fragments) feed 20 v_mfma_f32_16x16x32_f16 per wave; LDS-DMA refills with one stage of lead; one barrier
// per stage, placed between the two half stages (igemm3's schedule) or at the stage end.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
// MODE 0: load, wait, MFMA per half stage (single set), barrier at the stage end
// MODE 1: two sets, loads of the next half stage issued in front of the MFMAs of this one, barrier at the stage end
// MODE 2: two sets, barrier between the half stages (after MFMA set0, before loading set0 of the next stage)
// bit 2 (4): LDS-DMA refills (3 KB per wave and stage) with one stage of lead
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe8(float* out, const f16* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 3, wc = wave >> 2;
    for (int i = tid; i < 36 * 1024; i += 512) ((float*)smem)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64u << 20, 0x00020000);
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int abase = 110 * 1024 + (wc * 80 + lr) * 128 + ((lg ^ (lr & 7)) << 4);
    const int bbase = (wp * 64 + lr) * 128 + ((lg ^ (lr & 7)) << 4);
    f16x8 a0[5], b0[4], a1[5], b1[4];
    auto load = [&](f16x8 (&a)[5], f16x8 (&b)[4], int off, int ks) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *(const f16x8*)(smem + ((abase + i * 2048 + (off & 1) * 20480) ^ (ks * 64)));
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const f16x8*)(smem + ((bbase + j * 2048 + (off % 9) * 128 + (off & 1) * 55296) ^ (ks * 64)));
    };
    auto mma = [&](const f16x8 (&a)[5], const f16x8 (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto dma = [&](int it) {
#pragma unroll
        for (int q = 0; q < 3; ++q) lds_dma16(rs, smem + 130 * 1024 + (wave * 3 + q) * 1024, (unsigned)(((it * 24 + wave * 3 + q) & 16383) * 1024 + lane * 16));
    };
    constexpr int M = MODE & 3;
    if (M >= 1) load(a0, b0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (M == 0) {
            if (MODE & 4) dma(it);
            load(a0, b0, it, 0);
            mma(a0, b0);
            load(a1, b1, it, 1);
            mma(a1, b1);
            if (MODE & 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if (M == 1) {
            if (MODE & 4) dma(it);
            load(a1, b1, it, 1);
            mma(a0, b0);
            load(a0, b0, it + 1, 0);
            mma(a1, b1);
            if (MODE & 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            load(a1, b1, it, 1);
            mma(a0, b0);
            if (MODE & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (MODE & 4) dma(it);
            load(a0, b0, it + 1, 0);
            mma(a1, b1);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out, const f16* src, int iters) {
    (void)hipFuncSetAttribute((const void*)probe8<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe8<MODE>), dim3(256), dim3(512), 160 * 1024, 0, out, src, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe8<MODE>), dim3(256), dim3(512), 160 * 1024, 0, out, src, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * 8 * 40 * 16384.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-72s %8.3f ms  %7.1f ns/stage  %7.1f TFLOP/s\n", name, ms, ms * 1e6 / iters, tf);
}

int main() {
    float* out; f16* src;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, 64u << 20);
    (void)hipMemset(src, 0, 64u << 20);
    const int it = 20000;
    run<0>("one set: load, MFMA, load, MFMA, barrier", out, src, it);
    run<1>("two sets, next half stage's loads in front of the MFMAs, barrier at the end", out, src, it);
    run<2>("two sets, barrier between the half stages", out, src, it);
    run<4>("one set + LDS-DMA (one stage of lead)", out, src, it);
    run<5>("two sets, barrier at the end + LDS-DMA", out, src, it);
    run<6>("two sets, barrier between the half stages + LDS-DMA (igemm3 schedule)", out, src, it);
    return 0;
}
