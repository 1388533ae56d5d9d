#!/bin/bash
# C4 (isolate batch) on the GPU box: cohort parity tests, then the batch of 12 isolates in one call vs one isolate per call
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c4
mkdir -p $O
timeout -k 10 300 python3 -u -m pytest tests/test_cohort.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -u bench.py --config C1_ecoli_100x --steps 20 --warmup 3 --no-cpu-baseline --dump-profile $O/c1_events.txt > $O/c1_bench.json 2> $O/c1_bench.err || { tail -20 $O/c1_bench.err; exit 1; }
cat $O/c1_bench.json
timeout -k 10 400 python3 -u bench.py --config C4_ecoli_batch --steps 10 --warmup 3 --dump-profile $O/c4_events.txt > $O/c4_bench.json 2> $O/c4_bench.err || { tail -20 $O/c4_bench.err; exit 1; }
cat $O/c4_bench.json
head -16 $O/c4_events.txt; tail -1 $O/c4_events.txt
