#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "from fastposecnn_amd import build; build.build(extra=['-DFPC_STAMP_VOTE'])" > gpurun_out/tb.log 2>&1 || { tail -20 gpurun_out/tb.log; exit 1; }
python tools_dev/vote_stamps.py --hn 1000 --frames 32 --bits --prune 1 2>&1 | tail -12
python tools_dev/vote_stamps.py --hn 1000 --frames 32 --bits --prune 0 2>&1 | tail -5
