#!/bin/bash
# GPU box: the inflate kernel's tests, then its throughput on the end-to-end leg's BAM file
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/inflate
mkdir -p $O
timeout -k 10 300 python3 -u -m pytest tests/test_inflate.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python3 -u scripts/bench_inflate.py > $O/r03_gpu_inflate.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/r03_gpu_inflate.json
