#!/bin/bash
# Adjoint time of ONE shard of G of the metric ray set (low occupancy: 1-2 waves per SIMD) under the development
# ablations of BackArgs::experiment: where does a lone wave's step go?  -> gpurun_out/lone/*.json
set -o pipefail
O=gpurun_out/lone; mkdir -p $O
for G in 16 8 1; do
  for X in 0 1 2 3 5; do
    f=$O/G${G}_x$X.json
    timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-variants --shard-of $G --experiment $X > $f 2> ${f%.json}.err || echo "G=$G x=$X failed"
  done
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/lone/*.json')):
    try: d=json.load(open(f))
    except Exception: continue
    p=d['phase_ms']; print(os.path.basename(f)[:-5].ljust(10),'rays %8d fwd %.3f adj %.3f'%(d['config']['rays_rank0'],p['trace'],p['backtrace']))
PY
