#!/bin/bash
set -o pipefail
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 300 python tools/check_flat.py > $O/check_flat.log 2>&1; echo "check_flat rc=$?"; tail -8 $O/check_flat.log
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for v in win:0 flat:0x80000; do
  name=${v%%:*}; fl=${v##*:}
  $B --adj-flags $fl > $O/full_$name.json 2> $O/full_$name.err; echo "full $name rc=$?"
  $B --adj-flags $fl --shard-of 8 > $O/s8_$name.json 2> $O/s8_$name.err; echo "s8 $name rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
