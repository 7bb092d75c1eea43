// Micro-probe: how do ds_read_b128 fragment reads and MFMAs share a CU?  One workgroup of 8 waves per CU (the igemm4 shape):
// per iteration a wave issues 18 ds_read_b128 (1 KB each, swizzled conflict-free rows like the kernels) and / or 40
// v_mfma_f32_16x16x32_f16.  Prints cycles per iteration per mode (wall clock x the measured shader clock).
//   hipcc --offload-arch=gfx950 -O3 -o lds_mfma_probe lds_mfma_probe.hip && ./lds_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

template <int MODE, int NRD, int NMM>   // MODE bit 0: reads, bit 1: MFMAs, bit 2: software-pipelined (reads of i+1 behind MFMAs of i)
// MODE bit 4: LDS-DMA refills (3 one-KB pieces per wave and iteration = 24 KB per CU, igemm4's weight tile + halo piece), waited
// for in front of the barrier (bit 6: only the PREVIOUS iteration's pieces are waited for: one iteration of lead); bit 5: the same source lines for every CU (as the weight tile is) instead of distinct ones
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, long long* clk, const f16* src) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
    for (int i = tid; i < 36 * 1024; i += 512) ((float*)smem)[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    f32x4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 fr[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) fr[r] = f16x8{1, 2, 3, 4, 5, 6, 7, 8};
    const int base = (wave * 16 + lr) * 128 + ((lg ^ (lr & 7)) << 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64u << 20, 0x00020000);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int off = (it & 7) * 2048;
        if (MODE & 16) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned piece = (unsigned)(it * 24 + wave * 3 + q) + ((MODE & 32) ? 0u : blockIdx.x * 4099u);
                lds_dma16(rs, smem + 80 * 1024 + (wave * 3 + q) * 1024, (piece & 32767u) * 1024u + lane * 16u);
            }
        }
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) fr[r] = *(const f16x8*)(smem + base + ((off + r * 8192) & 0xFFFF));
        }
        if (!(MODE & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE & 2) {
#pragma unroll
            for (int m = 0; m < NMM; ++m) acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[m % NRD], fr[(m + 1) % NRD], acc[m % 10], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NRD; ++r) acc[r % 10][0] += (float)fr[r][0];
        }
        if (MODE & 16) { if (MODE & 64) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (MODE & 8) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int MODE, int NRD, int NMM>
void run(const char* name, float* out, long long* clk, int iters) {
    (void)hipFuncSetAttribute((const void*)probe<MODE, NRD, NMM>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static f16* src = nullptr;
    if (!src) { (void)hipMalloc(&src, 64u << 20); (void)hipMemset(src, 0, 64u << 20); }
    hipLaunchKernelGGL((probe<MODE, NRD, NMM>), dim3(256), dim3(512), 144 * 1024, 0, out, iters, clk, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, NRD, NMM>), dim3(256), dim3(512), 144 * 1024, 0, out, iters, clk, src);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    (void)hipMemcpy(&c, clk, sizeof c, hipMemcpyDeviceToHost);
    // clock64 counts at a fixed 100 MHz reference on this part: report wall time per iteration and cycles at 2.1 GHz
    printf("%-44s %8.3f ms  %8.1f ns/iter  ~%7.0f clk/iter @2.1GHz   LDS %6.1f B/clk/CU   MFMA %5.1f %% of peak\n", name, ms, ms * 1e6 / iters,
           ms * 1e6 / iters * 2.1, (MODE & 1) ? 8.0 * NRD * 1024 / (ms * 1e6 / iters * 2.1) : 0.0,
           (MODE & 2) ? 100.0 * (2.0 * NMM * 16 * 16 * 32 * 2) / (ms * 1e6 / iters * 2.1) / 1024.0 : 0.0);
}

int main() {
    float* out; long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 8);
    const int it = 20000;
    run<1, 18, 40>("reads only (18 x 1 KB per wave)", out, clk, it);
    run<9, 18, 40>("reads only + barrier", out, clk, it);
    run<2, 18, 40>("MFMA only (40 per wave)", out, clk, it);
    run<3, 18, 40>("reads, wait, MFMAs (serial in the wave)", out, clk, it);
    run<11, 18, 40>("reads, wait, MFMAs, barrier", out, clk, it);
    run<7, 18, 40>("reads of i+1 behind MFMAs of i (no wait)", out, clk, it);
    run<15, 18, 40>("reads behind MFMAs + barrier", out, clk, it);
    run<7, 9, 40>("9 reads behind 40 MFMAs", out, clk, it);
    run<7, 13, 80>("13 reads behind 80 MFMAs (128x80 wave tile)", out, clk, it);
    run<3, 9, 40>("9 reads, wait, 40 MFMAs", out, clk, it);
    run<11 + 16, 18, 40>("reads, wait, MFMAs, barrier + LDS-DMA (distinct lines per CU)", out, clk, it);
    run<11 + 48, 18, 40>("reads, wait, MFMAs, barrier + LDS-DMA (same lines for all CUs)", out, clk, it);
    run<10 + 48, 18, 40>("MFMAs + barrier + LDS-DMA (same lines), no fragment reads", out, clk, it);
    run<9 + 48, 18, 40>("reads + barrier + LDS-DMA (same lines), no MFMAs", out, clk, it);
    run<11 + 48 + 64, 18, 40>("reads, wait, MFMAs, barrier + LDS-DMA with one iteration of lead", out, clk, it);
    run<15 + 48 + 64, 18, 40>("reads behind MFMAs, barrier + LDS-DMA with one iteration of lead", out, clk, it);
    run<10 + 48 + 64, 18, 40>("MFMAs + barrier + LDS-DMA with lead, no fragment reads", out, clk, it);
    return 0;
}
