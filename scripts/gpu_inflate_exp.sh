#!/bin/bash
# GPU box: the inflate tests, then its throughput (diagnostic while working on the decoder)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/inflate_exp
mkdir -p $O
timeout -k 10 300 python3 -u -m pytest tests/test_inflate.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python3 -u scripts/bench_inflate.py --reps 5 > $O/base.json 2> $O/base.err || { tail -5 $O/base.err; exit 1; }
cat $O/base.json
