#!/bin/bash
# parity tests of the vote, then the per-kernel profile at the three bench settings
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/m_t.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 gpurun_out/m_t.log | cut -c1-100)"
if grep -q "Memory access fault" gpurun_out/m_t.log; then echo FAULT; tail -30 gpurun_out/m_t.log | cut -c1-200; exit 1; fi
[ $rc = 0 ] || { tail -40 gpurun_out/m_t.log | cut -c1-220; exit 1; }
[ "$1" = noprof ] && exit 0
bash tools_dev/r4_vote_prof.sh ${1:-new} 2>&1 | grep -v amdgpu.ids | grep "==\|k_vote\|per-call\|fault"
