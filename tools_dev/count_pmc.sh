#!/bin/bash
# SQ counters of k_vote_count (separate --pmc passes, kernel trace only)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU"; do
    i=$((i+1))
    cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -o p -- python3 /root/repo/tools_dev/vote_loop.py --hn 1000 --iters 40 > $OUT/p$i.log 2>&1
    cd /root/repo
    python - <<PY
import csv, glob, collections, statistics
path = glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True)
if not path:
    print("no counters for pass $i:", open("$OUT/p$i.log").read()[-400:])
else:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path[0])):
        if "k_vote_count" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: statistics.median(v) for k, v in d.items()})
PY
done
