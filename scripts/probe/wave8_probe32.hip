// A/B of the MFMA shape in igemm4's inner loop: the 8-wave (2 per SIMD) loop of wave8_probe.hip with
// v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_16x16x32_f16.  Wave tile 64 px x 96 ch (2 x 3 fragments of 32 x 32; the channel tile
// of the BC = 192 kernel) or 64 x 64 (BC = 128): per 16-wide k-step 3 (2) weight + 2 pixel fragment reads (ds_read_b128: row l % 32,
// 16 bytes at k-half l / 32) feed 6 (4) MFMAs of 32 768 FLOP.  Fragment reads per FLOP: 0.42 / 0.5 of a ds_read_b128 per 16 384 FLOP
// against 0.45 for the 64 x 80 tile of 16 x 16 x 32 fragments - about the same by construction; what changes is the number of MFMA
// issue slots (half) and the time each MFMA leaves for the LDS returns (32 cycles instead of 16).  LDS rows of 128 bytes; the
// 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 (conflict-free for the 32-row fragments' lane groups).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
// FA weight fragments (32 channels each) per wave; MODE bit 0: second fragment set (next k-step's reads in front of this k-step's MFMAs),
// bit 2: LDS-DMA refills (3 KB per wave and stage) with one stage of lead.  A stage = 4 k-steps of 16 (one 128-byte row), as in igemm4.
template <int FA, int MODE>
__global__ __launch_bounds__(512, 2) void probe32(float* out, const f16* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, kh = lane >> 5;
    const int wp = wave & 3, wc = wave >> 2;
    for (int i = tid; i < 36 * 1024; i += 512) ((float*)smem)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64u << 20, 0x00020000);
    f32x16 acc[FA][2];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int arow = wc * (32 * FA) + l32, brow = wp * 64 + l32;
    auto addr = [&](int base, int row, int t) { return base + row * 128 + ((((t * 2 + kh)) ^ ((row >> 1) & 7)) << 4); };
    f16x8 a0[FA], b0[2], a1[FA], b1[2];
    auto load = [&](f16x8 (&a)[FA], f16x8 (&b)[2], int off, int t) {
#pragma unroll
        for (int i = 0; i < FA; ++i) a[i] = *(const f16x8*)(smem + addr(110 * 1024 + (off & 1) * 24576, arow + 32 * i, t));
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *(const f16x8*)(smem + addr((off & 1) * 55296, brow + 32 * j + (off % 9), t));
    };
    auto mma = [&](const f16x8 (&a)[FA], const f16x8 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto dma = [&](int it) {
#pragma unroll
        for (int q = 0; q < 3; ++q) lds_dma16(rs, smem + 130 * 1024 + (wave * 3 + q) * 1024, (unsigned)(((it * 24 + wave * 3 + q) & 16383) * 1024 + lane * 16));
    };
    if (MODE & 1) load(a0, b0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 4) dma(it);
        if (!(MODE & 1)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { load(a0, b0, it, t); mma(a0, b0); }
        } else {
            load(a1, b1, it, 1); mma(a0, b0);
            load(a0, b0, it, 2); mma(a1, b1);
            load(a1, b1, it, 3); mma(a0, b0);
            load(a0, b0, it + 1, 0); mma(a1, b1);
        }
        if (MODE & 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int FA, int MODE>
void run(const char* name, float* out, const f16* src, int iters) {
    (void)hipFuncSetAttribute((const void*)probe32<FA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe32<FA, MODE>), dim3(256), dim3(512), 160 * 1024, 0, out, src, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe32<FA, MODE>), dim3(256), dim3(512), 160 * 1024, 0, out, src, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * 8 * (4.0 * FA * 2) * 32768.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-86s %8.3f ms  %7.1f ns/stage  %7.1f TFLOP/s\n", name, ms, ms * 1e6 / iters, tf);
}

int main() {
    float* out; f16* src;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&src, 64u << 20);
    (void)hipMemset(src, 0, 64u << 20);
    const int it = 20000;
    run<3, 0>("32x32x16, wave tile 64 x 96: load, MFMA per k-step, barrier per stage", out, src, it);
    run<3, 1>("32x32x16, wave tile 64 x 96: two fragment sets", out, src, it);
    run<3, 4>("32x32x16, wave tile 64 x 96: one set + LDS-DMA (one stage of lead)", out, src, it);
    run<3, 5>("32x32x16, wave tile 64 x 96: two sets + LDS-DMA", out, src, it);
    run<2, 0>("32x32x16, wave tile 64 x 64: load, MFMA per k-step, barrier per stage", out, src, it);
    run<2, 5>("32x32x16, wave tile 64 x 64: two sets + LDS-DMA", out, src, it);
    return 0;
}
