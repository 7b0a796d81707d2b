#!/bin/bash
# round-2 confirmation run: smoke, full GPU test tier, plain bench line with the final library.
set -o pipefail
O=gpurun_out/r2check; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
