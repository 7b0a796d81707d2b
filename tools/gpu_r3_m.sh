#!/bin/bash
set -o pipefail
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 3 > $O/auto.json 2> $O/auto.err; echo "rc=$?" >> $O/auto.err
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --variant-steps 2 --no-cpu-baseline --variants cube6_rotated --debug-counters --adj-flags 0x1000000 > $O/dbg.json 2> $O/dbg.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3m/auto.json'))
pc=d.get('parity_check') or {}
print('ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['phase_ms'].items() if v},'parity',pc.get('ok'),pc.get('rel_l2'))
for k,v in d.get('variants',{}).items():
    if isinstance(v,dict): print('    ',k,'step %.2f fwd %.2f adj %.2f ratio %.2f relL2 %.1e'%(v['ms_per_step'],v['trace'],v['backtrace'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics']))
PY
grep -h "debug" $O/dbg.err | tail -1 | cut -c1-900
timeout -k 10 600 python -m pytest tests/test_sensor.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_optimizer.py -m gpu -q -x 2>&1 | tail -3
