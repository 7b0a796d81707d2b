#!/bin/bash
set -o pipefail
O=gpurun_out/r3v; mkdir -p $O
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 3 > $O/auto.json 2> $O/auto.err; echo "rc=$?" >> $O/auto.err
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --adj-flags 0x1000000 > $O/ring.json 2> $O/ring.err
python - <<'PY'
import json
for tag in ('auto','ring'):
    d=json.load(open(f'gpurun_out/r3v/{tag}.json'))
    pc=d.get('parity_check') or {}
    print(tag,'ms/step %.3f adj %.3f'%(d['ms_per_step'],d['phase_ms']['backtrace']),'parity',pc.get('ok'),pc.get('rel_l2'))
    for k,v in d.get('variants',{}).items():
        if isinstance(v,dict): print('    ',k,'fwd %.2f adj %.2f ratio %.2f relL2 %.1e'%(v['trace'],v['backtrace'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics']))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -3
