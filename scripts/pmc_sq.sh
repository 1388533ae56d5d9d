#!/bin/bash
# SQ-level counters of one config (diagnostic): wave cycles, waits, LDS conflicts, instruction cache
#   bash scripts/pmc_sq.sh C5_tenth tag
set -e
CFG=${1:-C5_tenth}; T=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq_$T
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/a -- python3 bench.py --quick --config $CFG --steps 3 --warmup 1 > $O/a.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/b -- python3 bench.py --quick --config $CFG --steps 3 --warmup 1 > $O/b.log 2>&1
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for sub in ("a", "b"):
    for path in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): cnt[k] += 1
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
         "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_IFETCH", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES"]
with open(os.path.join(sys.argv[1], "sq_table.txt"), "w") as fh:
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0))[:12]:
        n = max(cnt[k], 1)
        fh.write(k + "  (launches %d)\n" % n)
        for c in names:
            if c in tot[k]: fh.write("   %-22s %16.0f per launch\n" % (c, tot[k][c] / n))
print(open(os.path.join(sys.argv[1], "sq_table.txt")).read()[:3000])
PY
rm -rf $O/a $O/b
