#!/bin/bash
# round-3 evidence run with the final library: profiles (kernel stats + PMC) of the metric workload and of the six rotated
# views, plain bench line, per-config timings, iteration benchmark, view probe, sweeps, 2-rank gloo rehearsal.
set -o pipefail
O=gpurun_out/r3final; mkdir -p $O
bash tools/profile_bench.sh r3 > $O/profile.log 2>&1; echo "profile rc=$?"
bash tools/profile_bench.sh r3_cube6 --workload cube6_rotated > $O/profile_cube6.log 2>&1; echo "profile cube6 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 400 python tools/run_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 > $O/iteration_4views.json 2> $O/iteration.err; echo "iteration rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 --views 1 --nbins 512 > $O/iteration_1view.json 2>> $O/iteration.err; echo "iteration1 rc=$?"
timeout -k 10 300 python tools/probe_views.py > $O/probe_views.txt 2> $O/probe_views.err; echo "probe rc=$?"
timeout -k 10 300 python tools/profile_sweeps.py > $O/sweeps.txt 2> $O/sweeps.err; echo "sweeps rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 300 python bench.py --rays 16777216 --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_16M.json 2> $O/bench_16M.err; echo "16M rc=$?"
for f in $O/*.err; do echo "== $f"; tail -n 2 $f | cut -c1-300; done
