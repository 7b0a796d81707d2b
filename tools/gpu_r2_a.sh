#!/bin/bash
# round-2 GPU session A: smoke, full gpu tests, bench A-B of the forward tap reuse (SLP on / off builds),
# per-shard timings, multi-rank launcher rehearsal.  Everything goes to gpurun_out/.
set -o pipefail
O=gpurun_out/r2a; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke_rc=$?" | tee -a $O/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest_rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
NOSLP=$PWD/adjointnonlinearraytracing_amd/csrc/_build_noslp/libdrrt_noslp.so
for v in default:0 off:0x10000 cell:0x20000; do
  name=${v%%:*}; fl=${v##*:}
  $B --fwd-flags $fl > $O/bench_slp_$name.json 2> $O/bench_slp_$name.err; echo "slp $name rc=$?"
  DRRT_HIP_LIB=$NOSLP $B --fwd-flags $fl > $O/bench_noslp_$name.json 2> $O/bench_noslp_$name.err; echo "noslp $name rc=$?"
done
for g in 2 4 8; do $B --shard-of $g > $O/shard_of_$g.json 2> $O/shard_of_$g.err; echo "shard $g rc=$?"; done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 60 python bench.py --gpus 8 > $O/bench_gpus8.json 2> $O/bench_gpus8.err; echo "gpus8 rc=$? (expected non-zero)"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a/*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['phase_ms'].items()}, 'n_gpus',d['n_gpus'], d.get('weak_scaling',{}).get('ms_per_step'))
    except Exception as e:
        print(f.split('/')[-1], 'unreadable', e)
PY
