#!/bin/bash
# development loop on the GPU box: parity tests, then the per-kernel tables of the C5-tenth and C2 workloads
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/iter
mkdir -p $O
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -u bench.py --config C5_tenth --steps 5 --warmup 2 --no-cpu-baseline --dump-profile $O/c5t_events.txt > $O/c5t_bench.json 2> $O/c5t_bench.err || { tail -20 $O/c5t_bench.err; exit 1; }
head -28 $O/c5t_events.txt; tail -1 $O/c5t_events.txt
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile $O/c2_events.txt > $O/c2_bench.json 2> $O/c2_bench.err || { tail -20 $O/c2_bench.err; exit 1; }
head -14 $O/c2_events.txt; tail -1 $O/c2_events.txt
