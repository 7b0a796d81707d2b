#!/bin/bash
# Round-4 evidence run, ONCE, with the final library (the round-3 review asked for one collection at the end instead of one
# per micro-change).  Two gpurun calls (the limit per call is 20 minutes):
#   gpurun -- 'bash tools/gpu_r4_final.sh A'   profiles (kernel stats + PMC incl. the effective clock) of the metric workload and of
#                                             the six rotated views, plain bench line
#   gpurun -- 'bash tools/gpu_r4_final.sh B'   per-shard timings of every shard index, chunking cost, per-config timings, iteration
#                                             benchmark, sweeps, 2-rank gloo rehearsals (plain, slab-wise reduce, view-wise shards)
# Then locally: bash tools/collect_r4_evidence.sh
set -o pipefail
O=gpurun_out/r4final; mkdir -p $O
if [ "$1" = "A" ]; then
bash tools/profile_bench.sh r4 > $O/profile.log 2>&1; echo "profile rc=$?"
bash tools/profile_bench.sh r4_cube6 --workload cube6_rotated > $O/profile_cube6.log 2>&1; echo "profile cube6 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
else
bash tools/run_shards.sh r4 > $O/shards.log 2>&1; echo "shards rc=$?"
for K in 0 4; do timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --shard-of 8 --shard-index 3 --adjoint-chunks $K > $O/chunk_G8_k3_K$K.json 2> $O/chunk_K$K.err; done; echo "chunks rc=$?"
timeout -k 10 400 python tools/run_configs.py > $O/configs.json 2> $O/configs.err; echo "configs rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 > $O/iteration_4views.json 2> $O/iteration.err; echo "iteration rc=$?"
timeout -k 10 300 python tools/bench_iteration.py --iters 10 --views 1 --nbins 512 > $O/iteration_1view.json 2>> $O/iteration.err; echo "iteration1 rc=$?"
timeout -k 10 300 python tools/profile_sweeps.py > $O/sweeps.txt 2> $O/sweeps.err; echo "sweeps rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --scaling strong --adjoint-chunks 4 --overlap-reduce > $O/bench_gloo2_overlap.json 2> $O/bench_gloo2_overlap.err; echo "gloo2 overlap rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --scaling strong --workload cube6_rotated > $O/bench_gloo2_cube6.json 2> $O/bench_gloo2_cube6.err; echo "gloo2 cube6 rc=$?"
fi
for f in $O/*.err; do echo "== $f"; tail -n 2 $f | cut -c1-300; done
