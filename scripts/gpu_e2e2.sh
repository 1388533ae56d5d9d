#!/bin/bash
# GPU box: the end-to-end leg of bench.py with both decoders
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/e2e
mkdir -p $O
timeout -k 10 500 python3 -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --table-configs "" --end-to-end-reads 4000000 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['end_to_end'], indent=1))"
