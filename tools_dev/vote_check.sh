#!/bin/bash
# parity tests of the vote, then the per-kernel profile at both bench settings, then the phase stamps
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/m_t.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 gpurun_out/m_t.log | cut -c1-80)"
if grep -q "Memory access fault" gpurun_out/m_t.log; then echo FAULT; exit 1; fi
[ $rc = 0 ] || { tail -30 gpurun_out/m_t.log | cut -c1-200; exit 1; }
bash tools_dev/vote_prof_quick.sh 2>&1 | grep -v "^$" | grep -v amdgpu.ids | grep "k_vote\|per-call\|fault\|fillBuffer"
if [ "$1" = stamps ]; then
  python -c "from fastposecnn_amd import build; build.build(extra=['-DFPC_STAMP_VOTE'])" > gpurun_out/tb.log 2>&1
  python tools_dev/vote_stamps.py --hn 1000 --frames 1 2>&1 | tail -4; python tools_dev/vote_stamps.py --hn 128 --frames 32 2>&1 | tail -4
fi
