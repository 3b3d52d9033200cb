#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1 1000 b1_hn1000" "32 128 b32_hn128"; do
    set -- $cfg; B=$1; HN=$2; TAG=$3
    D=$R/gpurun_out/prof_$TAG
    mkdir -p $D
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 > $D/stats.log 2>&1
    cd $R
    python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 8
    grep per-call $D/stats.log
    python tools_dev/vote_loop.py --hn $HN --frames $B --iters 300 | grep per-call
done
