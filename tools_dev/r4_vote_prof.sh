#!/bin/bash
# per-kernel profile of the vote at the three bench settings (f32 masks and bit words): rocprofv3 kernel stats of a launch loop
# Usage on the GPU box: bash tools_dev/r4_vote_prof.sh [tag]
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAGX=${1:-base}
for cfg in "1 1000 b1_hn1000" "32 128 b32_hn128" "32 1000 b32_hn1000"; do
  set -- $cfg; B=$1; HN=$2; TAG=$3
  for BITS in "" "--bits"; do
    D=$R/gpurun_out/r4prof_${TAGX}_${TAG}${BITS:+_bits}
    rm -rf $D; mkdir -p $D
    cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools_dev/vote_loop.py --hn $HN --frames $B --iters 200 --sets 8 $BITS > $D/stats.log 2>&1
    cd $R
    echo "== $TAG $BITS"
    python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 8 --out $R/gpurun_out/r4_${TAGX}_vote_${TAG}${BITS:+_bits}_kernel_stats.csv | grep "k_vote"
    python tools_dev/vote_loop.py --hn $HN --frames $B --iters 300 --sets 8 $BITS | grep per-call
  done
done
