#!/bin/bash
# round-3 session B: light-field sort key vs chord key (existing kernels), counters
set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke_rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
for tag in lf chord; do
  F=""; [ $tag = chord ] && F="--fwd-flags 0x800000 --adj-flags 0x800000"
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $F > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?" >> $O/bench_$tag.err
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --variant-steps 2 --no-cpu-baseline --debug-counters $F > $O/dbg_$tag.json 2> $O/dbg_$tag.err
  echo "== $tag"; grep debug $O/dbg_$tag.err
done
python - <<'PY'
import json
for tag in ('lf','chord'):
    d=json.load(open(f'gpurun_out/r3b/bench_{tag}.json'))
    print(tag,'ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['phase_ms'].items() if v})
    for k,v in d.get('variants',{}).items():
        if isinstance(v,dict): print('   ',k,'step %.2f fwd %.2f adj %.2f sort %.2f ratio %.2f relL2 %.1e'%(v['ms_per_step'],v['trace'],v['backtrace'],v['sort'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics']))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
