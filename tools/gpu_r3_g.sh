#!/bin/bash
# round-3 session G: classifier-selected kernels (box / ring v2 + step hint); ring forced; counters
set -o pipefail
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 3 > $O/auto.json 2> $O/auto.err; echo "rc=$?" >> $O/auto.err
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --adj-flags 0x1000000 > $O/ring.json 2> $O/ring.err; echo "rc=$?" >> $O/ring.err
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --adj-flags 0x400000 > $O/box.json 2> $O/box.err; echo "rc=$?" >> $O/box.err
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --variant-steps 2 --no-cpu-baseline --variants cube6_rotated --debug-counters --adj-flags 0x1000000 > $O/dbg.json 2> $O/dbg.err
python - <<'PY'
import json
for tag in ('auto','ring','box'):
    try: d=json.load(open(f'gpurun_out/r3g/{tag}.json'))
    except Exception as e: print(tag,'unreadable'); continue
    pc=d.get('parity_check') or {}
    print(tag,'ms/step %.3f adj %.3f'%(d['ms_per_step'],d['phase_ms']['backtrace']),'parity',pc.get('ok'),pc.get('rel_l2'))
    for k,v in d.get('variants',{}).items():
        if isinstance(v,dict): print('    ',k,'adj %.2f ratio %.2f relL2 %.1e steps %d'%(v['backtrace'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics'],v['adj_ray_steps']))
PY
grep -h "debug\|rror" $O/*.err | cut -c1-1100
