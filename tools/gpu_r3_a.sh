#!/bin/bash
# round-3 session A: smoke, bench with variants + parity_check, event counters of the three ray sets
set -o pipefail
O=gpurun_out/r3a; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke_rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench_rc=$?" >> $O/bench.err
tail -3 $O/bench.err
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --variant-steps 2 --no-cpu-baseline --debug-counters > $O/bench_dbg.json 2> $O/bench_dbg.err; echo "rc=$?" >> $O/bench_dbg.err
grep debug $O/bench_dbg.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a/bench.json'))
print('value %.4g ms/step %.3f'%(d['value'],d['ms_per_step']), d['phase_ms'])
print('parity', d.get('parity_check'))
for k,v in d.get('variants',{}).items(): print(k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a not in('rays','rays_per_view')})
PY
