#!/bin/bash
set -o pipefail
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for rep in 1 2; do
  $B > $O/new_$rep.json 2> $O/new_$rep.err; echo "new rc=$?"
  $B --fwd-flags 0x100000 --adj-flags 0x80000 > $O/legacy_$rep.json 2> $O/legacy_$rep.err; echo "legacy rc=$?"
done
$B --shard-of 8 > $O/s8.json 2> $O/s8.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2i/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
