#!/bin/bash
# round-3 session C: ring-window adjoint vs box-window adjoint
set -o pipefail
O=gpurun_out/r3c; mkdir -p $O
for tag in ring box; do
  F=""; [ $tag = ring ] && F="--adj-flags 0x1000000"
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 4 $F > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?" >> $O/bench_$tag.err
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --variant-steps 2 --no-cpu-baseline --debug-counters $F > $O/dbg_$tag.json 2> $O/dbg_$tag.err
  echo "== $tag"; grep -h "debug\|PARITY\|Error\|error" $O/dbg_$tag.err $O/bench_$tag.err | cut -c1-400
done
python - <<'PY'
import json
for tag in ('ring','box'):
    try: d=json.load(open(f'gpurun_out/r3c/bench_{tag}.json'))
    except Exception as e: print(tag,'unreadable',e); continue
    print(tag,'ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['phase_ms'].items() if v})
    pc=d.get('parity_check') or {}
    print('   parity ok',pc.get('ok'),'rel_l2',pc.get('rel_l2'),'xt',pc.get('xt_bit_exact'),'adj steps',pc.get('adj_ray_steps_gpu'),pc.get('adj_ray_steps_oracle'))
    for k,v in d.get('variants',{}).items():
        if isinstance(v,dict): print('   ',k,'step %.2f fwd %.2f adj %.2f ratio %.2f relL2 %.1e'%(v['ms_per_step'],v['trace'],v['backtrace'],v['adj_ns_ratio_to_headline'],v['grad_rel_l2_vs_direct_atomics']))
PY
