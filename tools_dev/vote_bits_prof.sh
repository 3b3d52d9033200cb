#!/bin/bash
# per-kernel times of the vote with the mask bit words (cold rotating inputs), both bench settings
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "1 1000" "32 128"; do set -- $cfg
  D=$R/gpurun_out/prof_bits_b$1; rm -rf $D; mkdir -p $D
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools_dev/vote_loop.py --hn $2 --frames $1 --iters 200 --sets 8 --bits > $D.log 2>&1
  cd $R; python tools_dev/kstats.py $(ls $D/*/*kernel_stats.csv | tail -1) --top 6 | grep k_vote; grep per-call $D.log
done
