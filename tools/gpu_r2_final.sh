#!/bin/bash
# round-2 evidence run: profiles (kernel stats + PMC), bench line, per-shard timings, per-config timings, sweeps,
# 2-rank gloo rehearsal of the self-launching bench.  Outputs under gpurun_out/r2final (copied to profiles/ afterwards).
set -o pipefail
O=gpurun_out/r2final; mkdir -p $O
bash tools/profile_bench.sh r2 > $O/profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for g in 2 4 8 16; do timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shard-of $g > $O/shard_of_$g.json 2> $O/shard_of_$g.err; echo "shard $g rc=$?"; done
timeout -k 10 400 python tools/run_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
timeout -k 10 300 python tools/profile_sweeps.py > $O/sweeps.txt 2> $O/sweeps.err; echo "sweeps rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 60 python bench.py --gpus 8 > $O/bench_gpus8.json 2> $O/bench_gpus8.err; echo "gpus8 rc=$? (expected 2)"
timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --fwd-flags 0x100000 --adj-flags 0x80000 > $O/bench_legacy_kernels.json 2> $O/bench_legacy.err; echo "legacy rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 > $O/iteration_4views.json 2> $O/iteration.err; echo "iteration rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 --views 1 --nbins 512 > $O/iteration_1view.json 2>> $O/iteration.err; echo "iteration1 rc=$?"
timeout -k 10 300 python tools/probe_views.py > $O/probe_views.txt 2> $O/probe_views.err; echo "probe rc=$?"
timeout -k 10 300 python tools/bench_optim.py 256 20 > $O/optim_256.json 2> $O/optim.err; echo "optim rc=$?"

