#!/bin/bash
set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 300 python tools/check_flat.py > $O/check_flat.log 2>&1; echo "check_flat rc=$?"; tail -7 $O/check_flat.log
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for e in 0 1 3 4; do
  $B --adj-flags 0x80000 --experiment $e > $O/flat_exp$e.json 2> $O/flat_exp$e.err; echo "flat exp$e rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
