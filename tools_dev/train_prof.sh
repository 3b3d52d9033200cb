#!/bin/bash
# one steady-state training step per kernel (rocprofv3 kernel trace of bench.py --train)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_train; rm -rf $D; mkdir -p $D
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --train --steps 6 --warmup 3 > $D/log.txt 2>&1
cd $R; python tools_dev/train_step_stats.py $(ls $D/*/*kernel_trace.csv | tail -1) 40 > $D/step.txt; cat $D/step.txt | head -24
