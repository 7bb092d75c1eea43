#!/bin/bash
# the whole GPU suite + smoke() (gpurun --timeout 1900 -- bash scripts/gpu/full_suite.sh)
R=$(pwd); O=$R/gpurun_out/r6t; mkdir -p $O; export TMPDIR=/tmp
timeout 2600 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -25 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
timeout 300 python scripts/tiled_chop512.py 512 parity > $O/tiled_chop512_parity.txt 2>&1; echo "tiled rc=$?"; tail -5 $O/tiled_chop512_parity.txt
