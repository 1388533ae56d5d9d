#!/bin/bash
# GPU box: parity tests (golden, IR, fullsize), then the C2 and C5 step times
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/quick
mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_einverted.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -u bench.py --quick --steps 20 --warmup 3 --dump-profile $O/c2_events.txt > $O/c2_bench.json 2> $O/c2_bench.err || { tail -20 $O/c2_bench.err; exit 1; }
head -8 $O/c2_events.txt; tail -1 $O/c2_events.txt
timeout -k 10 400 python3 -u bench.py --quick --config C5_meta_50mb --steps 10 --warmup 3 --dump-profile $O/c5_events.txt > $O/c5_bench.json 2> $O/c5_bench.err || { tail -20 $O/c5_bench.err; exit 1; }
head -8 $O/c5_events.txt; tail -1 $O/c5_events.txt
