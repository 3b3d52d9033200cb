#!/bin/bash
# k_vote_plan time against the instance count (tight launch loop, rocprofv3 stats)
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/plan
mkdir -p $OUT
for fr in ${FRS:-1 4 8 16 32}; do
  rm -rf $OUT/f$fr; cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f$fr -- python3 /root/repo/tools_dev/vote_loop.py --hn ${HN:-128} --frames $fr --iters 100 > $OUT/f$fr.log 2>&1
  cd /root/repo
  f=$(ls $OUT/f$fr/*/*kernel_stats.csv | tail -1)
  echo "== frames $fr"; python tools_dev/kstats.py $f --top 4 | tail -4 | cut -c1-110
done
