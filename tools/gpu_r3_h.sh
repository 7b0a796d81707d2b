#!/bin/bash
# ablations of the ring kernel on the six rotated views
set -o pipefail
O=gpurun_out/r3h; mkdir -p $O
for ex in 0 1 2 3; do
  F="--experiment $ex"; [ $ex = 0 ] && F="--debug-counters"
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --variant-steps 3 --no-cpu-baseline --variants cube6_rotated $F --adj-flags 0x1000000 > $O/ring_$ex.json 2> $O/ring_$ex.err || echo "$ex failed"
done
python - <<'PY'
import json
for ex in (0,1,2,3):
    try: d=json.load(open(f'gpurun_out/r3h/ring_{ex}.json'))
    except Exception as e: print(ex,'unreadable'); continue
    v=d['variants']['cube6_rotated']
    print('ring experiment',ex,'headline adj %.2f'%d['phase_ms']['backtrace'],'cube6 adj %.2f fwd %.2f'%(v['backtrace'],v['trace']))
PY
