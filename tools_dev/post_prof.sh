#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 1 32; do
  D=$R/gpurun_out/prof_post_b$B; rm -rf $D; mkdir -p $D
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools_dev/post_loop.py --frames $B --iters 200 > $D/log.txt 2>&1
  cd $R; python tools_dev/kstats.py $(ls $D/*/*kernel_stats.csv | tail -1) --top 9 | grep -v "^total"; grep "per call" $D/log.txt
done
