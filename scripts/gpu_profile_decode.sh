#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the device decode (three calls on the end-to-end leg's file)
#   bash scripts/gpu_profile_decode.sh r03
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_decode
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 -u scripts/profile_decode.py /tmp/mfk_prof_decode.bam --make 4000000 > $O/make.json
timeout -k 10 200 python3 -u scripts/profile_decode.py /tmp/mfk_prof_decode.bam > $O/${R}_decode_calls.txt 2> $O/plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 scripts/profile_decode.py /tmp/mfk_prof_decode.bam > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${R}_rocprofv3_kernel_stats_decode.csv
rm -rf $O/stats /tmp/mfk_prof_decode.bam
cat $O/make.json $O/${R}_decode_calls.txt
head -14 $O/${R}_rocprofv3_kernel_stats_decode.csv | cut -c1-150
