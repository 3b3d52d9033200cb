#!/bin/bash
# profiles of the vote sequence: PMC traffic / VALU passes and kernel-trace stats, B=1 hn=1000 and B=32 hn=128
set -u
export TMPDIR=/tmp
R=/root/repo
for cfg in "1 1000 b1_hn1000" "32 128 b32_hn128"; do
    set -- $cfg; B=$1; HN=$2; TAG=$3
    D=$R/gpurun_out/prof_$TAG
    mkdir -p $D
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        first=${c%% *}
        cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$first -o p -- python3 $R/tools_dev/vote_time.py $B $HN 20 > $D/$first.log 2>&1
    done
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 > $D/stats.log 2>&1
    cd $R
    python tools_dev/vote_traffic.py $D $R/gpurun_out/r02_vote_traffic_$TAG.json $B $HN | tail -3
    cp $(ls $D/stats/*/*kernel_stats.csv | tail -1) $R/gpurun_out/r02_vote_${TAG}_kernel_stats.csv
    grep per-call $D/stats.log
done
