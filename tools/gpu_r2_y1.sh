#!/bin/bash
set -o pipefail
bash tools/gpu_r2_check.sh || exit 1
O=gpurun_out/r2check
timeout -k 10 400 python tools/run_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
