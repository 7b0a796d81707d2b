#!/bin/bash
# round-2 GPU session B: shard timings with / without the LDS spreading of small grids; reuse default check
set -o pipefail
O=gpurun_out/r2b; mkdir -p $O
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
for g in 2 4 8 16; do
  $B --shard-of $g > $O/shard_spread_$g.json 2> $O/shard_spread_$g.err; echo "spread $g rc=$?"
  DRRT_DEV_NO_SPREAD=1 $B --shard-of $g > $O/shard_nospread_$g.json 2> $O/shard_nospread_$g.err; echo "nospread $g rc=$?"
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_sensor.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2b/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
