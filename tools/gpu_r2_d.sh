#!/bin/bash
# round-2 GPU session D: pipelined adjoint vs unpipelined (full size and 1/8, 1/16 shards) + parity tests
set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_baseline_configs.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for v in pipe:0 nopipe:0x40000; do
  name=${v%%:*}; fl=${v##*:}
  $B --adj-flags $fl > $O/full_$name.json 2> $O/full_$name.err; echo "full $name rc=$?"
  $B --adj-flags $fl --shard-of 8 > $O/s8_$name.json 2> $O/s8_$name.err; echo "s8 $name rc=$?"
  $B --adj-flags $fl --shard-of 16 > $O/s16_$name.json 2> $O/s16_$name.err; echo "s16 $name rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2d/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()})
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
