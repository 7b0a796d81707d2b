#!/bin/bash
# round-3 session F: step hint on/off x box/ring window
set -o pipefail
O=gpurun_out/r3f; mkdir -p $O
for kern in box ring; do for hint in on off; do
  A=""; [ $kern = ring ] && A="--adj-flags 0x1000000"
  H=""; [ $hint = off ] && H="--no-step-hint"
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-seconds 3 $A $H > $O/${kern}_$hint.json 2> $O/${kern}_$hint.err; echo "rc=$?" >> $O/${kern}_$hint.err
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --variant-steps 2 --no-cpu-baseline --variants cube6_rotated --debug-counters $A $H > $O/dbg_${kern}_$hint.json 2> $O/dbg_${kern}_$hint.err
done; done
python - <<'PY'
import json
for kern in ('box','ring'):
  for hint in ('on','off'):
    try: d=json.load(open(f'gpurun_out/r3f/{kern}_{hint}.json'))
    except Exception as e: print(kern,hint,'unreadable'); continue
    pc=d.get('parity_check') or {}
    print(kern,'hint',hint,'ms/step %.3f adj %.3f'%(d['ms_per_step'],d['phase_ms']['backtrace']),'parity',pc.get('ok'),'%.2e'%pc.get('rel_l2',-1), pc.get('adj_ray_steps_gpu')==pc.get('adj_ray_steps_oracle'))
    for k,v in d.get('variants',{}).items():
        if isinstance(v,dict): print('    ',k,'adj %.2f ratio %.2f relL2 %.1e'%(v['backtrace'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics']))
PY
grep -h debug $O/dbg_*_on.err | cut -c1-900
