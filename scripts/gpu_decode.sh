#!/bin/bash
# GPU box: the device decoder's tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/decode
mkdir -p $O
timeout -k 10 500 python3 -u -m pytest tests/test_gpu_decode.py tests/test_inflate.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
