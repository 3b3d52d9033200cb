#!/bin/bash
# round 5: parity of the progressive count, then A/B timings of the vote at config 3 (bit words), then kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-progressive or v3 or vote}" > gpurun_out/r5_t.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 gpurun_out/r5_t.log | cut -c1-120)"
if grep -q "Memory access fault" gpurun_out/r5_t.log; then echo FAULT; tail -30 gpurun_out/r5_t.log | cut -c1-200; exit 1; fi
[ $rc = 0 ] || { tail -60 gpurun_out/r5_t.log | cut -c1-240; exit 1; }
[ "$1" = noprof ] && exit 0
for args in "--prune 0" "--prune 1 --cum 5,10" "--prune 1 --cum 4,9" "--prune 1 --cum 6,11" "--prune 1 --cum 8" "--prune 1 --cum 4,8,12"; do
  echo "== $args"; timeout 200 python tools_dev/vote_loop.py --hn 1000 --frames 32 --iters 200 --sets 8 --bits --info $args 2>&1 | grep "per-call\|alive"
done
for tag in "0" "1"; do
  D=gpurun_out/r5prof_prune$tag; rm -rf $D; mkdir -p $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/stats -- python3 $GRAFT_REPO_ROOT/tools_dev/vote_loop.py --hn 1000 --frames 32 --iters 200 --sets 8 --bits --prune $tag > $GRAFT_REPO_ROOT/$D/stats.log 2>&1)
  echo "== kernel stats prune=$tag"
  python tools_dev/kstats.py $(ls $D/stats/*/*kernel_stats.csv | tail -1) --top 10 --out gpurun_out/r5_vote_b32_hn1000_bits_prune${tag}_kernel_stats.csv | grep "k_vote"
done
