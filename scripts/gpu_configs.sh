#!/bin/bash
# every BASELINE config at N=1: bench line + per-kernel table (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/configs
mkdir -p $O
for c in C1_ecoli_100x C3_klebsiella C5_meta_50mb; do
  timeout -k 10 500 python3 -u bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --dump-profile $O/${c}_events.txt > $O/${c}_bench.json 2> $O/${c}_bench.err || { tail -20 $O/${c}_bench.err; exit 1; }
  cat $O/${c}_bench.json
done
head -24 $O/C5_meta_50mb_events.txt; tail -1 $O/C5_meta_50mb_events.txt
