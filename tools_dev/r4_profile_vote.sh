#!/bin/bash
# Round-4 profiles of the vote sequence at the three bench settings (B=1 hn=1000; B=32 hn=128; B=32 hn=1000 = config 3), from the
# library in this tree, with the bit words (the model's pipeline) and on f32 masks (the stand-alone entry):
#   kernel-trace stats (rocprofv3 --kernel-trace --stats) of a 200-call launch loop over 8 input sets -> gpurun_out/r04_vote_[bits_]<tag>_kernel_stats.csv
#   PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU ...; separate runs, never with a system trace)  -> gpurun_out/r04_vote_[bits_]traffic_<tag>.json
# Usage on the GPU box:  bash tools_dev/r4_profile_vote.sh <commit>
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
COMMIT=${1:-unknown}
for BITS in bits ""; do
  TAGP=${BITS:+bits_}
  LOOPF=${BITS:+--bits}
  for cfg in "1 1000 b1_hn1000" "32 128 b32_hn128" "32 1000 b32_hn1000"; do
    set -- $cfg; B=$1; HN=$2; TAG=$3
    D=$R/gpurun_out/prof_${TAGP}$TAG
    rm -rf $D; mkdir -p $D
    for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        first=${c%% *}
        cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$first -o p -- python3 $R/tools_dev/vote_time.py $B $HN 20 $BITS > $D/$first.log 2>&1
    done
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 --sets 8 $LOOPF > $D/stats.log 2>&1
    cd $R
    python tools_dev/vote_traffic.py $D $R/gpurun_out/r04_vote_${TAGP}traffic_$TAG.json $B $HN $COMMIT | grep "traffic_bytes_per_launch\|traffic_over\|valu_wave"
    python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 6 --out $R/gpurun_out/r04_vote_${TAGP}${TAG}_kernel_stats.csv | grep "k_vote"
    python tools_dev/vote_loop.py --hn $HN --frames $B --iters 300 --sets 8 $LOOPF | grep per-call
  done
done
