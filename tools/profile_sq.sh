#!/bin/bash
# one rocprofv3 PMC pass (SQ counters) of bench.py: VALU instructions per launch etc.  usage: profile_sq.sh TAG [bench args]
tag=${1:-run}; shift
out=gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $out/bench.json 2> $out/sq.err
python3 - "$out" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/pmc_sq/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'drrt::' not in r['Kernel_Name']: continue
        kn=r['Kernel_Name'].split('(')[0]
        agg[kn][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,c in agg.items():
    print(kn, {k: '%.4g'%(sum(v)/len(v)) for k,v in c.items() })
PY
