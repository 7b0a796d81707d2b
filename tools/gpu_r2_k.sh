#!/bin/bash
set -o pipefail
O=gpurun_out/r2r; mkdir -p $O
for e in 0x900 0x500 0x600; do echo "== F ablation $e"; bash tools/ab_bench.sh $O/abl_$e "--quad --adj-flags $e" F; done
