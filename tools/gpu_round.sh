#!/bin/bash
# one GPU-box session: smoke, gpu tests, bench (+ variants)
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke_rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest_rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench_rc=$?" >> gpurun_out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print('value %.4g ray-steps/s  ms/step %.3f'%(d['value'],d['ms_per_step']), d['phase_ms'], 'adj frac %.3f fwd frac %.3f'%(d['roofline']['frac'], d['roofline_fwd']['frac']))
print(d.get('cpu_baseline'))
PY
tail -2 gpurun_out/bench.err
