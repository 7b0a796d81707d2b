#!/bin/bash
# parity of the rewritten flat kernels, then same-box A/B against the previous library and one SQ counter pass
set -o pipefail
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_bench.sh $O/ab "" base A
bash tools/profile_sq.sh r2j > $O/sq.txt 2>&1; cat $O/sq.txt
