#!/bin/bash
# C5 iteration loop on the GPU box: parity of both consensus paths + the full-size digests, then the per-kernel table.
#   gpurun --timeout 900 -- 'bash scripts/gpu_c5.sh TAG'
tag=${1:-x}
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread -m gpu -k "small_job or digest" 2>&1 | tail -4 &&
timeout -k 10 400 python bench.py --config C5_meta_50mb --steps 10 --no-cpu-baseline --dump-profile gpurun_out/r02_c5_${tag}_hip_events.txt > gpurun_out/r02_c5_${tag}.json 2> gpurun_out/r02_c5_${tag}.err &&
head -14 gpurun_out/r02_c5_${tag}_hip_events.txt && tail -1 gpurun_out/r02_c5_${tag}_hip_events.txt
