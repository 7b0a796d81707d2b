#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/bench_iteration.py (a whole optimisation iteration, 4 views): per-kernel time of the
# full flow.  Output: gpurun_out/prof_iter/ ; the summary is copied to profiles/rN_iteration_kernel_stats.csv.
set -o pipefail
out=gpurun_out/prof_iter; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_iteration.py --iters 10 > $out/iteration.json 2> $out/trace.err
echo "rc=$?"; cat $out/iteration.json | cut -c1-400
f=$(find $out -name "*_kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-200
