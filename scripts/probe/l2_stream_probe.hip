// Probe: what does the chip sustain when EVERY wave of EVERY CU streams its own slice of an L2-resident weight
// matrix - 1 KB contiguous per wave instruction (fragment-major order), straight into registers (MODE 0/1) or by LDS-DMA into a wave-private
// ring (MODE 2/3) - alone (MODE 0/2) or with 6 MFMAs issued per KB (MODE 1/3, the wino kernel's ratio at CF = 10)?
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream_probe scripts/probe/l2_stream_probe.hip && ./l2_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* w, unsigned wbytes, int kb_per_wave, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every workgroup walks the WHOLE matrix (like the conv's workgroups), wave w its eighth of it, starting at a block-dependent phase
    const unsigned slice = wbytes / 8;
    unsigned off = (unsigned)wave * slice + ((blockIdx.x * 7919u) % (slice / 1024u)) * 1024u;
    const unsigned end = (unsigned)(wave + 1) * slice, beg = (unsigned)wave * slice;
    f32x4 acc[6] = {};
    f16x8 a = {}, bsum = {};
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, wbytes, 0x00020000);
    char* ring = smem + wave * 8192;
    for (int i = 0; i < kb_per_wave; i += 4) {
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE < 2) v[u] = *(const f16x8*)(w + off + lane * 16);
            else dma16(rw, ring + ((i + u) & 7) * 1024, lane * 16, off);
            off += 1024; if (off >= end) off = beg;
        }
        if (MODE >= 2) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); 
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f16x8*)(ring + ((i + u + 4) & 7) * 1024 + lane * 16); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE & 1) {
#pragma unroll
                for (int m = 0; m < 6; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[u], a, acc[m], 0, 0, 0);
            } else bsum += v[u];
        }
    }
    float s = (float)bsum[0] + (float)bsum[3];
    for (int m = 0; m < 6; ++m) s += acc[m][0];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE> void run(const char* w, unsigned wbytes, float* out, const char* name) {
    const int kb = 2048, grid = 1024;
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(512), 65536, 0, w, wbytes, kb, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(512), 65536, 0, w, wbytes, kb, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 8 * kb * 1024.0;
    printf("%-46s %8.1f us  %7.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz)%s\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3 * 1e3 / 1e3,
           (MODE & 1) ? "  + 6 MFMA per KB" : "");
}
int main() {
    for (unsigned mb : {1u, 2u, 3u}) {
        const unsigned wbytes = mb * 1638400u / 1u;   // 1.64 / 3.3 / 4.9 MB (160->160, 320->160, 480->160 Winograd weights)
        char* w; float* out;
        hipMalloc(&w, wbytes); hipMemset(w, 0, wbytes); hipMalloc(&out, 4096);
        printf("weight matrix %.2f MB, every workgroup (1024 x 8 waves) streams 16 MB of it:\n", wbytes / 1e6);
        run<0>(w, wbytes, out, "  global_load_dwordx4 -> VGPR");
        run<1>(w, wbytes, out, "  global_load_dwordx4 -> VGPR");
        run<2>(w, wbytes, out, "  buffer_load ... lds (wave-private ring)");
        run<3>(w, wbytes, out, "  buffer_load ... lds (wave-private ring)");
        hipFree(w); hipFree(out);
    }
    return 0;
}
