#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest_rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for args in "" "--no-lds-bricks"; do
  echo "== bench $args"
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args 2> gpurun_out/exp.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.4g  ms/step %.3f'%(d['value'],d['ms_per_step']), {k:round(v,3) for k,v in d['phase_ms'].items()})"
  grep debug gpurun_out/exp.err
done
