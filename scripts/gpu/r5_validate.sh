#!/bin/bash
# Next round's first GPU call for the patches of scripts/probe/r5 (apply them and rebuild in the build container first:
#   for p in scripts/probe/r5/0*.patch; do git apply $p; done; python -m resshift_amd.build
# then: gpurun --timeout 900 -- bash scripts/gpu/r5_validate.sh).  Outputs under gpurun_out/r5v.  Every knob defaults to the patched behaviour;
# the A/B legs switch one of them off at a time.  Nothing here touches profiles/ - re-collect with scripts/gpu/final_measure.sh once the
# set of patches to keep is known.
R=$(pwd); O=$R/gpurun_out/r5v; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "patch_unembed or swin_mlp or halo" > $O/pytest_ops.log 2>&1; echo "op tests rc=$?"; tail -2 $O/pytest_ops.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "patch_unembed_fold or shortcut_fold or groupnorm_tails or large_weights or unet_forward_vs_oracle" > $O/pytest_eng.log 2>&1; echo "engine tests rc=$?"; grep -E "fold|passed|failed" $O/pytest_eng.log | tail -8
B="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-torch-baseline"
ab() {   # ab <tag> [ENV=VALUE ...]
  local tag=$1; shift
  env "$@" timeout 200 python bench.py $B > $O/ab_$tag.json 2> $O/ab_$tag.err; echo "ab $tag rc=$?"
  python - <<PY
import json
d=json.load(open("$O/ab_$tag.json")); r=d["roofline"]
print("$tag", d["ms_per_step"], d["config"].get("kernel_launches_per_step"), [(k["kernel"][:22], k["ms_per_step"], k["launches_per_step"]) for k in r["per_kernel"]], r["groupnorm"].get("ms_per_step"))
PY
}
ab all_on_1 RS_NOP=1
ab no_unembed_fold RS_UNEMBED_FOLD=0
ab no_f16_fold RS_SKIP_FOLD_F16=0
ab bp128_sk RS_SPLIT_BP128_SK=64
ab all_on_2 RS_NOP=1
timeout 400 python bench.py --steps 5 --warmup 2 --parity-images 8 --cpu-seconds 60 --no-torch-baseline > $O/bench_parity8.json 2> $O/bench_parity8.err; echo "parity bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_parity8.json')); print(d['value'], d['ms_per_step'], d.get('parity_vs_cpu_oracle')[0], d['value_fp16_unqualified']['value'], d['value_fp16_unqualified']['image_psnr_db'])"
