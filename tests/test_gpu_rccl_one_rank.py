"""The RCCL branches of the multi-GPU code, executed for real on the one GPU of the box (Next 5): a fresh child
process initialises backend "nccl" with world_size 1 and drives PoseGatherer (add from four streams / flush / finish /
latest) and ShardedLookaheadRAdam (reduce-scatter in backward, statistics all-reduce, parameter all-gather) through
`all_gather_into_tensor` / `reduce_scatter_tensor` on device memory; the gathered pose records must equal the collective-free
single-process path bit for bit, the trained parameters to rounding (tests/_rccl_one_rank_child.py).  The gloo tests cover world_size 2 on the CPU."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_device_branches_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_one_rank_child.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["ok"] is True and rec["backend"] == "nccl" and rec["world"] == 1
    assert rec["gatherer"]["collectives"] == 5 and rec["optimiser"]["buckets"] >= 3
