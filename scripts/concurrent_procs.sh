#!/bin/bash
# GPU box: P processes on one GPU, each with S library contexts in S threads, all starting their timed leg at the same wall-clock
# time.  bash scripts/concurrent_procs.sh P S STEPS   (aggregate = sum of the processes' `together.value` while they overlap)
P=${1:-2}; S=${2:-3}; STEPS=${3:-2000}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/procs
mkdir -p $O
AT=$(python3 -c "import time; print(time.time() + 75)")
pids=()
for i in $(seq 1 $P); do
  timeout -k 10 300 python3 -u scripts/concurrent_ctx.py --contexts $S --steps $STEPS --start-at $AT > $O/p${P}_s${S}_$i.json 2> $O/p${P}_s${S}_$i.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cat $O/p${P}_s${S}_*.json
exit $rc
