#!/bin/bash
# Per-shard timings on ONE GPU for the strong-scaling projection: shard 0 of G of the ray set of the
# metric workload and of BASELINE configs 3 and 4, no all-reduce.  -> gpurun_out/shards/*.json -> profiles/<tag>_shards.json
set -o pipefail
TAG=${1:-r3}
O=gpurun_out/shards; mkdir -p $O
run() { # name grid rays G extra
  local f=$O/$1_G$4$6.json
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-variants --grid $2 --rays $3 --shard-of $4 $5 > $f 2> ${f%.json}.err || echo "$1 G=$4 failed"
}
for G in 1 2 4 8 16; do run metric 256 1048576 $G; done
for G in 1 2 4 8; do run config3 65 1048576 $G; done
for G in 1 2 4 8; do run config4 256 4194304 $G; done
for G in 8 16; do run metric 256 1048576 $G --lds-bricks _bricks; done
python - "$TAG" <<'PY'
import json,glob,sys,os
out={}
for f in sorted(glob.glob('gpurun_out/shards/*.json')):
    try: d=json.load(open(f))
    except Exception: continue
    k=os.path.basename(f)[:-5]
    p=d['phase_ms']
    out[k]={'rays':d['config']['rays_rank0'],'grid':d['config']['grid'],'ms_per_step':d['ms_per_step'],'sort':p['sort_avg'],'pair_copy':p['pair_copy'],
            'trace':p['trace'],'backtrace':p['backtrace'],'fwd_ray_steps':d['config']['fwd_ray_steps_rank0'],'pair_grid':d['config']['pair_grid'],'lib_version':d.get('lib_version')}
    print(k.ljust(22),'rays %8d step %.3f sort %.3f fwd %.3f adj %.3f'%(out[k]['rays'],out[k]['ms_per_step'],out[k]['sort'],out[k]['trace'],out[k]['backtrace']))
json.dump(out,open(f'profiles/{sys.argv[1]}_shards.json','w'),indent=1,sort_keys=True)
PY
cp profiles/${TAG}_shards.json gpurun_out/${TAG}_shards.json
