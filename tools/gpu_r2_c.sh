#!/bin/bash
# round-2 GPU session C: where does a lone wave's step latency go?  adjoint ablations at 1/16 and full size
set -o pipefail
O=gpurun_out/r2c; mkdir -p $O
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for e in 0 1 2 3; do
  $B --shard-of 16 --experiment $e > $O/s16_exp$e.json 2> $O/s16_exp$e.err; echo "s16 exp$e rc=$?"
  $B --experiment $e > $O/full_exp$e.json 2> $O/full_exp$e.err; echo "full exp$e rc=$?"
done
$B --shard-of 16 --direct-atomics > $O/s16_direct.json 2> $O/s16_direct.err; echo "s16 direct rc=$?"
$B --shard-of 16 --no-sort > $O/s16_nosort.json 2> $O/s16_nosort.err; echo "s16 nosort rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
