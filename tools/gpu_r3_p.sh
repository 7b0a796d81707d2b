#!/bin/bash
set -o pipefail
O=gpurun_out/r3p; mkdir -p $O
for ex in 0 4; do
  F="--experiment $ex"; [ $ex = 0 ] && F="--debug-counters"
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --workload cube6_rotated $F --adj-flags 0x1000000 > $O/ring_$ex.json 2> $O/ring_$ex.err || echo "$ex failed"
done
python - <<'PY'
import json
for ex in (0,4):
    d=json.load(open(f'gpurun_out/r3p/ring_{ex}.json')); print('ring ABL experiment',ex,'cube6 adj %.2f'%d['phase_ms']['backtrace'])
PY
