#!/bin/bash
# Profiles `bench.py` on the GPU box: (1) kernel trace + stats, (2)-(4) PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Outputs under
# gpurun_out/prof_$1/ ; condense with tools/condense_profile.py into profiles/.
# usage: profile_bench.sh TAG [extra bench.py args]   (e.g. `profile_bench.sh r3_cube6 --workload cube6_rotated`)
tag=${1:-run}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# --no-variants: one ray set per profile, so that a kernel's average duration in the stats is that of ONE workload
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- $B > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- $B > $out/bench_write.json 2> $out/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_sq -- $B > $out/bench_sq.json 2> $out/sq.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out/pmc_tcc -- $B > $out/bench_tcc.json 2> $out/tcc.err
# effective shader clock of every dispatch: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration (MI355X_MICROARCH.md, DVFS)
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_clk -- $B > $out/bench_clk.json 2> $out/clk.err
find $out -name "*.csv" | head -30
for f in $out/*.err; do tail -n 2 "$f" | cut -c1-200; done
