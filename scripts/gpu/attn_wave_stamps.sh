#!/bin/bash
# phase stamps of the fused split attention per WAVE (0: first on a shared SIMD, 2: alone on its SIMD, 4: second on a shared SIMD).  The three ablate libraries are
# built with RS_ATTN_STAMP's `tid == 0` test in win_attn_split.hip changed to `tid == 64 * wave` (a one-line local edit, not in the tree):
#   RS_BUILD_DEFS="RS_SPLIT_ABLATE" RS_BUILD_OUT=resshift_amd/lib_ab_abl_w$w.so python -m resshift_amd.build     -> profiles/r6_attn_split_ab.txt
R=$(pwd); O=$R/gpurun_out/r6w; mkdir -p $O
for w in 0 2 4; do RESSHIFT_HIP_LIB=$R/resshift_amd/lib_ab_abl_w$w.so timeout 300 python scripts/attn_phases.py split > $O/phases_w$w.txt 2>&1; echo "== wave $w"; grep -A1 "64x64 shift 0" $O/phases_w$w.txt | tail -1 | cut -c1-400; done
