// Micro-probe for a 4-wave / 512-register implicit-GEMM inner loop (one wave per SIMD, 128 px x 80 ch wave tile, accumulators
// in AGPRs): per half stage 13 ds_read_b128 (5 weight + 8 pixel fragments) feed 40 v_mfma_f32_16x16x32_f16; optional LDS-DMA
// refill traffic (26 KB per stage per CU, as igemm4's weight tile + halo piece) and a barrier per stage.
//   hipcc --offload-arch=gfx950 -O3 -o wave4_probe wave4_probe.hip && ./wave4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

// MODE bit 0: double-buffered fragments (loads of the next half stage behind the MFMAs of this one), bit 1: LDS-DMA refills,
// bit 2: barrier per stage
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe4(float* out, const f16* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    const int wp = wave & 1, wc = wave >> 1;
    for (int i = tid; i < 36 * 1024; i += 256) ((float*)smem)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64u << 20, 0x00020000);
    f32x4 acc[5][8];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int abase = 110 * 1024 + (wc * 80 + lr) * 128 + ((lg ^ (lr & 7)) << 4);       // weight slot rows
    const int bbase = (wp * 128 + lr) * 128 + ((lg ^ (lr & 7)) << 4);                    // pixel rows
    f16x8 a0[5], b0[8], a1[5], b1[8];
    auto load = [&](f16x8 (&a)[5], f16x8 (&b)[8], int off, int ks) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *(const f16x8*)(smem + ((abase + i * 2048 + (off & 1) * 20480) ^ (ks * 64)));
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = *(const f16x8*)(smem + ((bbase + j * 2048 + (off % 9) * 128 + (off & 1) * 55296) ^ (ks * 64)));
    };
    auto mma = [&](const f16x8 (&a)[5], const f16x8 (&b)[8]) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    if (MODE & 1) load(a0, b0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {   // 26 one-KB pieces per stage per CU: 7 (waves 0, 1) or 6 per wave
            const int np = wave < 2 ? 7 : 6;
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (q < np) lds_dma16(rs, smem + 130 * 1024 + (wave * 7 + q) * 1024, (unsigned)(((it * 28 + wave * 7 + q) & 16383) * 1024 + lane * 16));
        }
        if (MODE & 1) {
            load(a1, b1, it, 1);
            mma(a0, b0);
            load(a0, b0, it + 1, 0);
            mma(a1, b1);
        } else {
            load(a0, b0, it, 0);
            mma(a0, b0);
            load(a1, b1, it, 1);
            mma(a1, b1);
        }
        if (MODE & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 4) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out, const f16* src, int iters) {
    (void)hipFuncSetAttribute((const void*)probe4<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe4<MODE>), dim3(256), dim3(256), 160 * 1024, 0, out, src, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe4<MODE>), dim3(256), dim3(256), 160 * 1024, 0, out, src, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // 80 MFMAs per wave and stage = 4 waves x 80 x 16384 x 2 FLOP per CU
    const double tf = 256.0 * 4 * 80 * 16384.0 * iters / (ms * 1e-3) / 1e12;   // 16x16x32 MACs = 16384 FLOP per MFMA
    printf("%-60s %8.3f ms  %7.1f ns/stage  %7.1f TFLOP/s  (%4.1f %% of 2500)\n", name, ms, ms * 1e6 / iters, tf, tf / 25.0);
}

int main() {
    float* out; f16* src;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, 64u << 20);
    (void)hipMemset(src, 0, 64u << 20);
    const int it = 20000;
    run<0>("4 waves: load, MFMA, load, MFMA", out, src, it);
    run<1>("4 waves: double-buffered fragments", out, src, it);
    run<5>("4 waves: double-buffered + barrier", out, src, it);
    run<7>("4 waves: double-buffered + LDS-DMA refills + barrier", out, src, it);
    run<6>("4 waves: single-buffered + LDS-DMA refills + barrier", out, src, it);
    return 0;
}
