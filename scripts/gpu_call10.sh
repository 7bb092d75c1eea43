#!/bin/bash
# round 3, call 10: streaming AE attention: op test, AE / engine parity, chop_size 512 tile A/B, bench A/B
O=gpurun_out/r3c10; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "ae_flash" > $O/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -6 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "autoencoder_vs or realsr_full or 128_tile or ae_attention or batch32 or decoder_with" > $O/pytest_eng.log 2>&1; echo "eng rc=$?"; tail -6 $O/pytest_eng.log
for fl in 1 0; do
  RS_AE_FLASH=$fl timeout 600 python scripts/tiled_chop512.py 512 fp16 > $O/tiled_chop512_flash$fl.txt 2> $O/tiled_chop512_flash$fl.err; echo "chop512 flash=$fl rc=$?"; head -6 $O/tiled_chop512_flash$fl.txt; grep shapes $O/tiled_chop512_flash$fl.err | head -4
done
for fl in 1 0 1 0; do
  for pol in fp16 parity; do
    RS_AE_FLASH=$fl timeout 300 python bench.py --precision $pol --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass > $O/b.json 2> $O/b.err; echo "$pol flash=$fl rc=$? $(python -c "import json;d=json.load(open('$O/b.json'));print(d['ms_per_step'])")"
  done
done
