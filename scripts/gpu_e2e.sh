#!/bin/bash
# GPU box: the tests touched by the record index / deep fixtures / communicator reset, then the end-to-end leg of bench.py
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/e2e
mkdir -p $O
timeout -k 10 400 python3 -u -m pytest tests/test_cli.py tests/test_gpu_comm.py tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "cli or comm or golden_tsvs" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python3 -u bench.py --steps 5 --warmup 2 --no-cpu-baseline --table-configs "" --end-to-end-reads 4000000 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['end_to_end'], indent=1))"
