// Halo-tile implicit GEMM, the two-workgroups-per-CU variant (igemm4_kernel.h, NWV = 4): 128-pixel tiles (4 x 32) on four waves, one halo
// buffer, <= 80 KB of LDS.  Separate translation unit: the instantiations compile in parallel with igemm4.hip's.
#include "igemm4_kernel.h"

extern "C" int rs_igemm4_w4_launch(const IGemmParams* pp, int in_dt, int BC, hipStream_t st) {
    const IGemmParams& p = *pp;
    hipError_t e;
    if (in_dt == RS_F16S) e = BC == 160 ? launch4_cfg<32, 160, true, 0, 4>(p, st) : launch4_cfg<32, 128, true, 0, 4>(p, st);
    else e = BC == 160 ? launch4_cfg<32, 160, false, 0, 4>(p, st) : (BC == 192 ? launch4_cfg<32, 192, false, 0, 4>(p, st) : launch4_cfg<32, 128, false, 0, 4>(p, st));
    return e == hipSuccess ? 0 : -1;
}
