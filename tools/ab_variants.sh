#!/bin/bash
# ab_variants.sh OUTDIR "bench args" NAME...  : like ab_bench.sh, but reports the `variants` leg (six rotated views, shifted plane)
set -o pipefail
O=$1; ARGS=$2; shift 2
mkdir -p $O
V=$PWD/adjointnonlinearraytracing_amd/csrc/_variants
for rep in 1 2; do
  for n in "$@"; do
    DRRT_HIP_LIB=$V/$n.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $ARGS > $O/${n}_$rep.json 2> $O/${n}_$rep.err || echo "$n rep $rep FAILED"
  done
done
python - "$O" <<'PY'
import json,glob,sys,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.load(open(f)); n=f.split('/')[-1].rsplit('_',1)[0]
        v=d.get('variants',{})
        r[n].append('head %.3f+%.3f | '%(d['phase_ms']['trace'],d['phase_ms']['backtrace']) + ' | '.join('%s %.2f+%.2f (%.0e)'%(k[:5],x['trace'],x['backtrace'],x['grad_rel_l2_vs_direct_atomics']) for k,x in v.items() if isinstance(x,dict)))
    except Exception as e: print(f,'unreadable',e)
for n,v in r.items(): print(n.ljust(8), ' || '.join(v))
PY
