#!/bin/bash
# Per-shard timings on ONE GPU for the strong-scaling projection: EVERY shard k of G of the ray set --
# a strong-scaling step is the max over ranks, and shard 0 of a pixel-ordered source is the slab at the rim of the lens,
# not the slowest one -- for the metric workload, the six rotated views (sharded view by view, dist.shard_views) and
# BASELINE configs 3 and 4; no all-reduce.  -> gpurun_out/shards_<tag>/*.json -> profiles/<tag>_shards.json with per-k rows
# and the max per (workload, G).
set -o pipefail
TAG=${1:-r4}
O=gpurun_out/shards_$TAG; mkdir -p $O
run() { # name grid rays G k extra
  local f=$O/$1_G$4_k$5.json
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --grid $2 --rays $3 --shard-of $4 --shard-index $5 $6 > $f 2> ${f%.json}.err || echo "$1 G=$4 k=$5 failed"
}
for G in 1 2 4 8; do for ((k=0;k<G;k++)); do run metric 256 1048576 $G $k; done; done
for G in 1 2 4 8; do for ((k=0;k<G;k++)); do run cube6 256 1048576 $G $k "--workload cube6_rotated"; done; done
for G in 1 4; do for ((k=0;k<G;k++)); do run config3 65 1048576 $G $k; done; done
for G in 1 8; do for ((k=0;k<G;k++)); do run config4 256 4194304 $G $k; done; done
echo progress: shards done
python - "$TAG" <<'PY'
import json,glob,sys,os,collections
tag=sys.argv[1]
rows={}; worst=collections.defaultdict(lambda: None)
for f in sorted(glob.glob(f'gpurun_out/shards_{tag}/*.json')):
    try: d=json.load(open(f))
    except Exception: continue
    k=os.path.basename(f)[:-5]
    name,G,idx=k.rsplit('_',2); G=int(G[1:]); idx=int(idx[1:])
    p=d['phase_ms']
    r={'workload':name,'G':G,'k':idx,'rays':d['config']['rays_rank0'],'grid':d['config']['grid'],'ms_per_step':d['ms_per_step'],'sort':p['sort_avg'],
       'pair_copy':p['pair_copy'],'trace':p['trace'],'backtrace':p['backtrace'],'fwd_ray_steps':d['config']['fwd_ray_steps_rank0'],
       'pair_grid':d['config']['pair_grid'],'adjoint_kernel':(d['config'].get('adjoint_kernel') or {}).get('kernel'),'lib_version':d.get('lib_version')}
    rows[k]=r
    w=worst[(name,G)]
    if w is None or r['ms_per_step']>w['ms_per_step']: worst[(name,G)]=r
out={'per_shard':rows,'max_over_shards':{f'{n}_G{G}':{'k':r['k'],'ms_per_step':r['ms_per_step'],'sort':r['sort'],'trace':r['trace'],'backtrace':r['backtrace'],
     'min_ms_per_step':min(x['ms_per_step'] for x in rows.values() if x['workload']==n and x['G']==G)} for (n,G),r in sorted(worst.items())}}
for k,v in out['max_over_shards'].items(): print(k.ljust(14),'slowest shard k=%d step %.3f (sort %.3f fwd %.3f adj %.3f); fastest %.3f'%(v['k'],v['ms_per_step'],v['sort'],v['trace'],v['backtrace'],v['min_ms_per_step']))
json.dump(out,open(f'profiles/{tag}_shards.json','w'),indent=1,sort_keys=True)
PY
cp profiles/${TAG}_shards.json gpurun_out/${TAG}_shards.json
