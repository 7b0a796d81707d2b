#!/bin/bash
# round-3 session D: ablations of the adjoint on the six rotated views (ABL instantiation; differences matter)
set -o pipefail
O=gpurun_out/r3d; mkdir -p $O
for kern in box ring; do
  K=""; [ $kern = ring ] && K="0x1000000"
  for ex in 0 1 2 3 5 6; do
    F="--experiment $ex"; [ $ex = 0 ] && F="--debug-counters"
    A=""; [ -n "$K" ] && A="--adj-flags $K"
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --variant-steps 3 --no-cpu-baseline --variants cube6_rotated $F $A > $O/${kern}_$ex.json 2> $O/${kern}_$ex.err || echo "$kern $ex failed"
  done
done
python - <<'PY'
import json
for kern in ('box','ring'):
    for ex in (0,1,2,3,5,6):
        try: d=json.load(open(f'gpurun_out/r3d/{kern}_{ex}.json'))
        except Exception as e: print(kern,ex,'unreadable'); continue
        v=d['variants']['cube6_rotated']
        print(kern,'experiment',ex,'headline adj %.2f'%d['phase_ms']['backtrace'],'cube6 adj %.2f'%v['backtrace'])
PY
