#!/bin/bash
# round 5: the pixel-resident lateral kernel — parity cases, stand-alone times, and the RCCL one-rank test's own output
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_net.py -x -q -k "conv2d" 2>&1 | tail -5
timeout 300 python tools_dev/lateral_time.py 2>&1 | tail -30
timeout 300 python tests/_rccl_one_rank_child.py 2>&1 | tail -30
