#!/bin/bash
# GPU box, end of a round: the driver's bench line (CPU legs included), host-side section timing of C2 and C5, the parity fuzz.
#   bash scripts/gpu_final.sh r03
R=${1:-r03}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -u bench.py > $O/${R}_bench_default.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench done"
MFK_HOST_TIMING=1 timeout -k 10 200 python3 -u scripts/host_timing.py C2_saureus_1000x > $O/${R}_host_timing_c2.txt 2>&1 || { tail -20 $O/${R}_host_timing_c2.txt; exit 1; }
MFK_HOST_TIMING=1 timeout -k 10 300 python3 -u scripts/host_timing.py C5_meta_50mb > $O/${R}_host_timing_c5.txt 2>&1 || { tail -20 $O/${R}_host_timing_c5.txt; exit 1; }
echo "host timing done"
timeout -k 10 330 python3 -u tests/fuzz_gpu.py --budget-s 240 > $O/${R}_fuzz.txt 2>&1 || { tail -20 $O/${R}_fuzz.txt; exit 1; }
tail -2 $O/${R}_fuzz.txt
