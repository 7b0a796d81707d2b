#!/bin/bash
# ab_flags.sh OUTDIR NAME=ADJFLAGS...  : same library, same box: the bench (headline + variants) under different adjoint flag bits
set -o pipefail
O=$1; shift
mkdir -p $O
for rep in 1 2; do
  for kv in "$@"; do
    n=${kv%%=*}; fl=${kv#*=}
    timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --adj-flags $fl > $O/${n}_$rep.json 2> $O/${n}_$rep.err || echo "$n rep $rep FAILED"
  done
done
python - "$O" <<'PY'
import json,glob,sys,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.load(open(f)); n=f.split('/')[-1].rsplit('_',1)[0]
        v=d.get('variants',{})
        r[n].append('step %.3f adj %.3f | '%(d['ms_per_step'],d['phase_ms']['backtrace']) + ' | '.join('%s adj %.2f (%.1e)'%(k[:5],x['backtrace'],x['grad_rel_l2_vs_direct_atomics']) for k,x in v.items() if isinstance(x,dict)))
    except Exception as e: print(f,'unreadable',e)
for n,v in r.items(): print(n.ljust(8), ' || '.join(v))
PY
