#!/bin/bash
# round 5: network parity tests + per-dispatch timelines (no stream stats, no bench line)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_rccl_one_rank.py -x -q 2>&1 | tail -5
NOSTREAM=1 NOBENCH=1 TAG=${TAG:-r05b} bash tools_dev/r5_profile_bench.sh 2>&1 | tail -40
