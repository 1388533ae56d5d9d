#!/bin/bash
# GPU box: HIP API calls of the quick bench (how many calls a step makes and what they cost the host)
#   bash scripts/gpu_hip_api.sh r03 C2_saureus_1000x c2
set -e
R=${1:-r03}; CFG=${2:-C2_saureus_1000x}; T=${3:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/hip_api_$T
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --hip-trace --stats --output-format csv -d $O/stats -- python3 bench.py --quick --config $CFG --steps 100 --warmup 3 > $O/stats.log 2>&1
cp $(find $O/stats -name '*hip_api_stats.csv' | head -1) $O/${R}_hip_api_stats_${T}.csv
rm -rf $O/stats
head -30 $O/${R}_hip_api_stats_${T}.csv | cut -c1-140
