"""The inference side's multi-rank path on the GPU box (Missing 3): `bench.py --gpus 2` with two ranks sharing
the one GPU over gloo (RCCL needs a device per rank) — native pack of the pose records, PoseGatherer on its side stream, the
collective, and the check that what every rank received equals every rank's records rebuilt locally, in rank order, with
sample ids offset by the shard's first image."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_gathered_records_match_single_rank():
    env = dict(os.environ, FPC_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                          "--min-seconds", "0", "--check-gather", "--gather-every", "3", "--encoder", "resnet18", "--batch", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling_measured"] is True and rec["rccl_ranks_seen"] == 2
    assert rec["collective_backend"] == "gloo" and len(rec["per_rank_img_per_s"]) == 2
    pg = rec["config"]["pose_gather"]
    assert pg["verified"] is True and pg["frames_per_collective"] == 3 and pg["us_per_collective"] > 0
    # 8 timed steps of 3 frames per collective: 2 full rounds + one flushed partial round per timed region, on both ranks alike
    assert pg["collectives"] >= 3
    assert rec["value"] > 0 and rec["config"]["global_batch"] == 2
