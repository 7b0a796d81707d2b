#!/bin/bash
# PMC of the ring variants on the six rotated views
set -o pipefail
O=gpurun_out/r3r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/adjointnonlinearraytracing_amd/csrc/_variants
for n in seq simple; do
  export DRRT_HIP_LIB=$V/$n.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --workload cube6_rotated --adj-flags 0x1000000 > $O/b_$n.json 2> $O/e_$n.err
  rocprofv3 --pmc TCC_EA0_ATOMIC_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/tcc_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --workload cube6_rotated --adj-flags 0x1000000 > $O/b2_$n.json 2> $O/e2_$n.err
done
python - <<'PY'
import csv,glob,collections
for n in ('seq','simple'):
    agg=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/r3r/*_{n}/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_backtrace_ring' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(n, {k:'%.3g'%(sum(v)/len(v)) for k,v in agg.items()})
PY
