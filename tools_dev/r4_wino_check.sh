#!/bin/bash
# split-precision Winograd change check: conv parity cases, K-loop stamps, then network tests + bench + frame timeline
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -k "conv2d" > gpurun_out/wino_pytest.log 2>&1; tail -3 gpurun_out/wino_pytest.log | cut -c1-300
python tools_dev/wino_stamps.py -5 2>&1 | tail -21
bash tools_dev/r4_net_check.sh
