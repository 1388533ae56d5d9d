#!/bin/bash
# Round profile: bench line, HIP-event table, rocprofv3 kernel stats and the two PMC passes (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round
rm -rf $O && mkdir -p $O
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 3 --dump-profile $O/hip_events.txt > $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 scripts/pmc_table.py $O/pmc_fetch $O/pmc_write C2_saureus_1000x $O/pmc_traffic.json > $O/pmc_table.log 2>&1
cat $O/bench.json
