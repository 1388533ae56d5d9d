#!/bin/bash
# Round profile of one config (run on the GPU box): HIP-event table, rocprofv3 kernel stats and the three PMC passes.
#   bash scripts/profile_round.sh r03 C2_saureus_1000x c2      (round tag, bench config, short tag)
set -e
R=${1:-r03}; CFG=${2:-C2_saureus_1000x}; T=${3:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round_$T
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 -u bench.py --quick --config $CFG --steps 20 --warmup 3 --dump-profile $O/${R}_bench_${T}_hip_events.txt > $O/${R}_bench_${T}_quick.json
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --quick --config $CFG --steps 20 --warmup 3 > $O/stats.log 2>&1
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --quick --config $CFG --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --quick --config $CFG --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
echo "write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -- python3 bench.py --quick --config $CFG --steps 3 --warmup 1 > $O/pmc_valu.log 2>&1
echo "valu done"
python3 scripts/pmc_table.py $O/pmc_fetch $O/pmc_write $CFG $O/${R}_pmc_traffic.json $O/pmc_valu $O/${R}_pmc_valu.json > $O/pmc_table.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${R}_rocprofv3_kernel_stats_${T}.csv
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_valu     # raw traces are large; the tables above are what is kept
tail -3 $O/${R}_bench_${T}_hip_events.txt
