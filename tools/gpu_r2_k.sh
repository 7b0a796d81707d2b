#!/bin/bash
set -o pipefail
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_optimizer.py tests/test_sensor.py tests/test_abi.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_optim.py 256 20 > $O/optim_256.json 2> $O/optim.err; cat $O/optim_256.json
timeout -k 10 300 python tools/bench_optim.py 128 20 > $O/optim_128.json 2>> $O/optim.err; cat $O/optim_128.json
