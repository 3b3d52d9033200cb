#!/bin/bash
# Round-3 profiles of the vote sequence at both bench settings, from the library in this tree:
#   kernel-trace stats (rocprofv3 --kernel-trace --stats) of a 200-call launch loop  -> gpurun_out/r03_vote_<tag>_kernel_stats.csv
#   PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU ...; separate runs, never with a system trace) -> gpurun_out/r03_vote_traffic_<tag>.json
# Usage on the GPU box:  bash tools_dev/profile_vote.sh <commit> [bits]
#   bits: the vote as the model's pipeline calls it, with the aggregation layer's mask bit words (the scan skips the f32
#         planes) -> files named r03_vote_bits_*; without: the stand-alone entry reading f32 masks
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
BITS=${2:-}
TAGP=${BITS:+bits_}
LOOPF=${BITS:+--bits}
for cfg in "1 1000 b1_hn1000" "32 128 b32_hn128"; do
    set -- $cfg; B=$1; HN=$2; TAG=$3
    D=$R/gpurun_out/prof_${TAGP}$TAG
    rm -rf $D; mkdir -p $D
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        first=${c%% *}
        cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$first -o p -- python3 $R/tools_dev/vote_time.py $B $HN 20 $BITS > $D/$first.log 2>&1
    done
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 --sets 8 $LOOPF > $D/stats.log 2>&1
    cd $R
    python tools_dev/vote_traffic.py $D $R/gpurun_out/r03_vote_${TAGP}traffic_$TAG.json $B $HN $COMMIT | tail -4
    python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 6 --out $R/gpurun_out/r03_vote_${TAGP}${TAG}_kernel_stats.csv | grep "k_vote"
    grep per-call $D/stats.log
    python tools_dev/vote_loop.py --hn $HN --frames $B --iters 300 --sets 8 $LOOPF | grep per-call
    python tools_dev/vote_loop.py --hn $HN --frames $B --iters 300 --sets 1 $LOOPF | grep per-call
done
