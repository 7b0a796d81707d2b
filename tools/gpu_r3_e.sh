#!/bin/bash
set -o pipefail
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --variant-steps 3 --no-cpu-baseline --variants cube6_rotated --debug-counters --adj-flags 0x1000000 > $O/ring.json 2> $O/ring.err
grep debug $O/ring.err
