// Halo-tile implicit GEMM, small-plane geometries (igemm4_kernel.h, SEG > 0): four images of an 8 x 8 plane or one 16 x 16 image per
// 256-pixel tile, split-K over stages.  Serves every 3x3 / stride-1 conv of the 16 x 16 and 8 x 8 UNet levels
// (models/unet.py:147,173 at ds = 16, 8).  Separate translation unit: the instantiations compile in parallel with igemm4.hip's.
#include "igemm4_kernel.h"

extern "C" int rs_igemm4_seg_launch(const IGemmParams* pp, int in_dt, int SEG, int BC, hipStream_t st) {
    const IGemmParams& p = *pp;
    hipError_t e;
    if (in_dt == RS_F16S) {
        if (SEG == 8) e = BC == 160 ? launch4_cfg<32, 160, true, 8>(p, st) : launch4_cfg<32, 128, true, 8>(p, st);
        else e = BC == 160 ? launch4_cfg<32, 160, true, 16>(p, st) : launch4_cfg<32, 128, true, 16>(p, st);
    } else {
        if (SEG == 8) e = BC == 160 ? launch4_cfg<32, 160, false, 8>(p, st) : launch4_cfg<32, 128, false, 8>(p, st);
        else e = BC == 160 ? launch4_cfg<32, 160, false, 16>(p, st) : launch4_cfg<32, 128, false, 16>(p, st);
    }
    return e == hipSuccess ? 0 : -1;
}
