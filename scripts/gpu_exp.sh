#!/bin/bash
# one-off experiment runner on the GPU box: C2 per-kernel table under a few environment settings
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp
mkdir -p $O
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
run() {
    tag=$1; shift
    env "$@" timeout -k 10 200 python3 -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --dump-profile $O/$tag.txt > $O/$tag.json 2> $O/$tag.err || { tail -20 $O/$tag.err; exit 1; }
    echo "== $tag"; grep -E "k_bucket_sort|wall ms" $O/$tag.txt | head -4
}
run ds512 MFK_X=0
run ds1024 MFK_BSORT_T=1024
