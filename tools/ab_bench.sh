#!/bin/bash
# ab_bench.sh OUTDIR "bench args" NAME...  : runs bench.py with each variant library twice, interleaved (same box, same session)
set -o pipefail
O=$1; ARGS=$2; shift 2
mkdir -p $O
V=$PWD/adjointnonlinearraytracing_amd/csrc/_variants
for rep in 1 2; do
  for n in "$@"; do
    DRRT_HIP_LIB=$V/$n.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $ARGS > $O/${n}_$rep.json 2> $O/${n}_$rep.err || echo "$n rep $rep FAILED"
  done
done
python - "$O" <<'PY'
import json,glob,sys,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.load(open(f)); n=f.split('/')[-1].rsplit('_',1)[0]
        r[n].append((d['phase_ms']['trace'], d['phase_ms']['backtrace'], d['ms_per_step']))
    except Exception as e: print(f,'unreadable')
for n,v in r.items(): print(n.ljust(14), ' | '.join('fwd %.3f adj %.3f step %.3f'%x for x in v))
PY
