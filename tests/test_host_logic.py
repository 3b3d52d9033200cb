"""CPU-side checks (no GPU, no compute calls into the HIP library):
  * libfpc_hip.so loads and exports every symbol include/fpc.h declares;
  * the engine's parameter table matches the drop-in model's state dict (smp naming);
  * the convolution planner returns valid tilings;
  * host logic: eps-threshold equivalence used by the kernels, pose-record packing,
    image sharding, and the world-size-2 all-gather of pose records over gloo.
"""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from fastposecnn_amd import build, _native
    build.build()
    return _native.lib()


def test_abi_exports_every_declared_symbol(hiplib):
    from fastposecnn_amd import _native
    hdr = open(os.path.join(REPO, "include", "fpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fpc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    raw = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} is declared in include/fpc.h but not exported"
    assert declared == set(_native.EXPORTED), declared ^ set(_native.EXPORTED)
    assert hiplib.fpc_abi_version() == 11
    assert hiplib.fpc_error_string(-2).decode().startswith("workspace")


def test_missing_library_fails_loudly(monkeypatch):
    from fastposecnn_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libfpc_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.lib()


def test_cpu_tensors_are_refused():
    import fastposecnn_amd.lib  # noqa: F401
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    with pytest.raises(RuntimeError):
        rvg.ransac_voting_layer_v3(torch.zeros((1, 8, 8)), torch.zeros((1, 8, 8, 1, 2)), 16)
    with pytest.raises(RuntimeError):
        al.AggregationLayer(None, 7).forward({"mask": torch.zeros((1, 8, 8), dtype=torch.int64)})


def test_eps_threshold_equivalence():
    """(double)x < 1e-6  <=>  x <= float32(1e-6) for every float x (common.hpp: below_eps)."""
    f = np.float32(1e-6)
    assert float(f) < 1e-6 < float(np.nextafter(f, np.float32(1)))
    for x in (np.nextafter(f, np.float32(0)), f, np.nextafter(f, np.float32(1)), np.float32(0), np.float32(1e-7)):
        assert (float(x) < 1e-6) == bool(x <= f)


@pytest.mark.parametrize("encoder", ["resnet18", "resnet34"])
def test_engine_param_table_matches_state_dict(hiplib, encoder):
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config
    hp = config.INFERENCE()
    hp.ENCODER = encoder
    m = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp)
    sd = dict(m.named_parameters())
    sd.update(dict(m.named_buffers()))
    h = ctypes.c_void_p()
    assert hiplib.fpc_net_create(encoder.encode(), 7, 2, 480, 640, ctypes.byref(h)) == 0
    try:
        n = hiplib.fpc_net_param_count(h)
        names = [hiplib.fpc_net_param_name(h, i).decode() for i in range(n)]
        assert len(set(names)) == n
        for i, name in enumerate(names):
            assert name in sd, name
            assert sd[name].numel() == hiplib.fpc_net_param_numel(h, i), name
        unused = [k for k in sd if k not in set(names) and not k.endswith("num_batches_tracked")]
        assert unused == []
        assert hiplib.fpc_net_workspace_bytes(h) > 0
        # forward before load_params is refused, not a crash
        assert hiplib.fpc_net_forward(h, *([None] * 11), None) == -1
    finally:
        hiplib.fpc_net_destroy(h)
    assert hiplib.fpc_net_create(b"resnet50", 7, 1, 480, 640, ctypes.byref(h)) == -1
    assert hiplib.fpc_net_create(b"resnet18", 7, 1, 481, 640, ctypes.byref(h)) == -1


def test_conv_planner_is_valid(hiplib):
    out = (ctypes.c_int * 4)()
    for B, Ho, Wo, Cin, Cout, k in [(1, 120, 160, 256, 128, 3), (1, 15, 20, 512, 512, 3), (32, 15, 20, 512, 512, 3),
                                    (1, 240, 320, 3, 64, 7), (1, 120, 160, 128, 7, 1), (2, 30, 40, 256, 256, 1)]:
        assert hiplib.fpc_conv2d_plan(B, Ho, Wo, Cin, Cout, k, k, 0, 0, 0, out) == 0
        bm, bn, ns, p32 = out
        ksteps = -(-Cin * k * k // 32)
        assert bm in (64, 128) and bn in (64, 128) and 1 <= ns <= 32
        per = -(-ksteps // ns)
        assert (ns - 1) * per < ksteps                      # no empty split
        assert p32 == -(-Ho * Wo // bm) * bm // 32
        if Cout % 4:
            assert ns == 1


def test_pose_record_roundtrip_and_sharding():
    from fastposecnn_amd import parallel
    assert [list(parallel.shard_indices(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    g = torch.Generator().manual_seed(0)
    n = 5
    agg = {"sample_ids": torch.arange(n), "class_ids": torch.randint(1, 7, (n,), generator=g),
           "quaternion": torch.randn(n, 4, generator=g), "scales": torch.randn(n, 3, generator=g),
           "xy": torch.randn(n, 2, generator=g), "z": torch.randn(n, 1, generator=g),
           "R": torch.randn(n, 3, 3, generator=g), "T": torch.randn(n, 3, generator=g), "RT": torch.randn(n, 4, 4, generator=g)}
    buf = parallel.pack_pose_records(agg, sample_offset=32, capacity=8)
    assert tuple(buf.shape) == (9, parallel.RECORD_WIDTH)
    out = parallel.unpack_pose_records(buf[None])
    assert torch.equal(out["sample_ids"], agg["sample_ids"] + 32) and torch.equal(out["class_ids"], agg["class_ids"])
    for k in ("quaternion", "scales", "xy", "z", "R", "T", "RT"):
        assert torch.equal(out[k].reshape(agg[k].shape), agg[k]), k
    with pytest.raises(RuntimeError):
        parallel.pack_pose_records(agg, 0, capacity=4)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from fastposecnn_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 2 + rank                                             # ragged instance counts
    g = torch.Generator().manual_seed(100 + rank)
    imgs = parallel.shard_indices(6, rank, world)
    agg = {"sample_ids": torch.randint(0, len(imgs), (n,), generator=g), "class_ids": torch.full((n,), rank + 1),
           "quaternion": torch.randn(n, 4, generator=g), "scales": torch.randn(n, 3, generator=g),
           "xy": torch.randn(n, 2, generator=g), "z": torch.randn(n, 1, generator=g),
           "R": torch.randn(n, 3, 3, generator=g), "T": torch.randn(n, 3, generator=g), "RT": torch.randn(n, 4, 4, generator=g)}
    gathered = parallel.all_gather_pose_records(agg, imgs[0], capacity=4)
    out = parallel.unpack_pose_records(gathered)
    q.put((rank, out["class_ids"].tolist(), out["sample_ids"].tolist(), float(out["RT"].sum()),
           (agg["sample_ids"] + imgs[0]).tolist(), float(agg["RT"].sum())))
    dist.destroy_process_group()


def test_all_gather_pose_records_gloo_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks see the same concatenation, in rank order, with globally offset sample ids
    assert res[0][1] == res[1][1] == [1, 1, 2, 2, 2]
    assert res[0][2] == res[1][2] == res[0][4] + res[1][4]
    assert abs(res[0][3] - (res[0][5] + res[1][5])) < 1e-4 and abs(res[0][3] - res[1][3]) < 1e-6


def _gather_agg(n, seed, nimg):
    g = torch.Generator().manual_seed(seed)
    return {"sample_ids": torch.randint(0, nimg, (n,), generator=g), "class_ids": torch.randint(1, 7, (n,), generator=g),
            "quaternion": torch.randn(n, 4, generator=g), "scales": torch.randn(n, 3, generator=g),
            "xy": torch.randn(n, 2, generator=g), "z": torch.randn(n, 1, generator=g),
            "R": torch.randn(n, 3, 3, generator=g), "T": torch.randn(n, 3, generator=g), "RT": torch.randn(n, 4, 4, generator=g)}


def _gatherer_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from fastposecnn_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = parallel.PoseGatherer(capacity=6, every=3)
    seen = []
    for f in range(8):                                       # 8 frames, 3 per collective: 3 + 3 + a flushed 2
        G.add(_gather_agg((f + rank) % 5, 10 * f + rank, 2), sample_offset=2 * rank)
        if (f + 1) % 3 == 0:
            seen.append(G.latest().clone())
    G.flush()
    seen.append(G.latest().clone())
    rows = []
    for block in seen:                                       # [world, every, capacity + 1, 40]; unfilled slots are empty
        for fr in range(block.shape[1] if block is not seen[-1] else 2):
            u = parallel.unpack_pose_records(block[:, fr])
            rows.append((u["class_ids"].tolist(), u["sample_ids"].tolist(), round(float(u["RT"].sum()), 4)))
    q.put((rank, G.collectives, rows))
    dist.destroy_process_group()


def test_pose_gatherer_batches_frames_gloo_world2():
    """Several frames per collective: 8 frames -> 3 all-gathers; every rank sees, frame by frame, the
    concatenation of both ranks' records in rank order with globally offset sample ids."""
    import torch.multiprocessing as mp
    from fastposecnn_amd import parallel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 3 and res[0][2] == res[1][2] and len(res[0][2]) == 8
    for f, (cls, sid, rt) in enumerate(res[0][2]):
        a0, a1 = _gather_agg(f % 5, 10 * f, 2), _gather_agg((f + 1) % 5, 10 * f + 1, 2)
        assert cls == a0["class_ids"].tolist() + a1["class_ids"].tolist()
        assert sid == a0["sample_ids"].tolist() + (a1["sample_ids"] + 2).tolist()
        assert abs(rt - float(a0["RT"].sum() + a1["RT"].sum())) < 1e-3


def _uneven_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from fastposecnn_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    num_images, every = 7, 3                                  # shards of 4 and 3 frames: 1 full round + (1 | 0) pending
    imgs = parallel.shard_indices(num_images, rank, world)
    G = parallel.PoseGatherer(capacity=6, every=every)
    rows = []
    def take(block):
        for fr in range(block.shape[1]):
            u = parallel.unpack_pose_records(block[:, fr])
            rows.append((u["class_ids"].tolist(), u["sample_ids"].tolist()))
    for k, img in enumerate(imgs):
        before = G.collectives
        G.add(_gather_agg(1 + img % 3, 100 + img, 1), sample_offset=img)
        if G.collectives != before:
            take(G.latest().clone())
    G.finish(parallel.PoseGatherer.rounds_for(num_images, world, every), device="cpu")
    take(G.latest().clone())
    q.put((rank, G.collectives, rows))
    dist.destroy_process_group()


def test_pose_gatherer_uneven_shards_stay_in_lockstep_gloo_world2():
    """ranks with different frame counts must issue the same number of collectives (7 images on 2 ranks,
    3 frames per collective: 4 / 3 frames -> rank 1 has nothing pending at the end and still takes part in round 2)."""
    import torch.multiprocessing as mp
    from fastposecnn_amd import parallel
    assert parallel.PoseGatherer.rounds_for(17, 4, 4) == 2 and parallel.PoseGatherer.rounds_for(7, 2, 3) == 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2 and res[0][2] == res[1][2]
    # every image's records arrive exactly once, with its global sample id
    want = sorted((_gather_agg(1 + img % 3, 100 + img, 1)["class_ids"].tolist(), img) for img in range(7))
    got = []
    shards = [list(parallel.shard_indices(7, r, 2)) for r in range(2)]
    for slot, (cls, sid) in enumerate(res[0][2]):
        rnd, fr = divmod(slot, 3)
        pos = 0
        for r in range(2):
            k = rnd * 3 + fr
            if k < len(shards[r]):
                n = 1 + shards[r][k] % 3
                assert sid[pos:pos + n] == [shards[r][k]] * n
                got.append((cls[pos:pos + n], shards[r][k]))
                pos += n
        assert pos == len(cls)
    assert sorted(got) == want


def test_config1_cpu_mask_head_plumbing():
    """BASELINE.json configs[0]: one frame, mask head only, CPU tensors, voting not involved — the reference's
    own CPU-runnable case.  Checks the forward() schema (keys, dtypes, shapes) of the drop-in model."""
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth
    hp = config.MASK_TRAINING()
    torch.manual_seed(0)
    m = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).eval()
    x = synth.make_image(0, 64, 96)[None]
    with torch.no_grad():
        out = m(x)
    assert set(out) == {"logits", "categorical", "aggregated"} and out["aggregated"] is None
    C = len(hp.SELECTED_CLASSES)
    assert tuple(out["logits"]["mask"].shape) == (1, C, 64, 96)
    assert tuple(out["logits"]["quaternion"].shape) == (1, 4 * (C - 1), 64, 96)
    assert tuple(out["logits"]["xy"].shape) == (1, 2 * (C - 1), 64, 96) and tuple(out["logits"]["z"].shape) == (1, C - 1, 64, 96)
    cat = out["categorical"]
    assert cat["mask"].dtype == torch.int64 and tuple(cat["mask"].shape) == (1, 64, 96)
    assert tuple(cat["quaternion"].shape) == (1, 4, 64, 96) and tuple(cat["z"].shape) == (1, 64, 96)
    fg = cat["mask"] != 0
    assert torch.equal(cat["mask"], torch.argmax(torch.nn.LogSoftmax(dim=1)(out["logits"]["mask"]), dim=1))
    qn = cat["quaternion"].norm(dim=1)
    assert torch.allclose(qn[fg], torch.ones_like(qn[fg]), atol=1e-5) and (qn[~fg] == 0).all()
    # frozen branches of the mask-training preset
    assert not any(p.requires_grad for p in m.rotation_decoder.parameters())
    assert all(p.requires_grad for p in m.mask_decoder.parameters())
    # state-dict names follow segmentation_models_pytorch
    sd = m.state_dict()
    for k in ("encoder.layer1.0.conv1.weight", "mask_decoder.p4.skip_conv.bias",
              "rotation_decoder.seg_blocks.0.block.2.block.1.weight", "scales_head.0.bias"):
        assert k in sd, k


def test_load_from_ckpt_lightning_round_trip(tmp_path):
    """F/lib/pose_regressor.py:506-539 with a Lightning-style checkpoint: keys carry the 'model.' prefix, and
    MODEL / BACKBONE_ARCH / ENCODER / ENCODER_WEIGHTS / SELECTED_CLASSES (only those) come from the checkpoint's
    'hyper_parameters', overriding the caller's HPARAM."""
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth
    hp_src = config.HEAD_TRAINING()
    hp_src.ENCODER = 'resnet34'
    hp_src.SELECTED_CLASSES = ['bg', 'bottle', 'bowl', 'mug']           # 4 classes: head widths change too
    torch.manual_seed(3)
    src = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp_src).eval()
    ckpt = {
        'state_dict': {'model.' + k: v.clone() for k, v in src.state_dict().items()},
        'hyper_parameters': {'MODEL': 'PoseRegressor', 'BACKBONE_ARCH': 'FPN', 'ENCODER': 'resnet34',
                             'ENCODER_WEIGHTS': None, 'SELECTED_CLASSES': hp_src.SELECTED_CLASSES,
                             'BATCH_SIZE': 99, 'HV_NUM_OF_HYPOTHESES': 7},
        'epoch': 3, 'global_step': 1234,
    }
    path = tmp_path / "last.ckpt"
    torch.save(ckpt, path)

    hp = config.INFERENCE()                                             # resnet18, 7 classes, imagenet weights
    torch.manual_seed(99)
    m = L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(str(path), hp).eval()
    assert hp.ENCODER == 'resnet34' and hp.ENCODER_WEIGHTS is None and hp.SELECTED_CLASSES == hp_src.SELECTED_CLASSES
    assert hp.BATCH_SIZE == 1 and hp.HV_NUM_OF_HYPOTHESES == 1000       # not among the five copied fields
    assert m.encoder.name == 'resnet34' and m.classes == 4
    assert tuple(m.rotation_head[0].weight.shape) == (12, 128, 1, 1)
    sd, want = m.state_dict(), src.state_dict()
    assert list(sd) == list(want)
    for k in want:
        assert torch.equal(sd[k], want[k]), k
    x = synth.make_image(0, 64, 64)[None]
    hp.PERFORM_AGGREGATION = False
    hp_src.PERFORM_AGGREGATION = False
    with torch.no_grad():
        a, b = m(x), src(x)
    for k in a["logits"]:
        assert torch.equal(a["logits"][k], b["logits"][k]), k
    # a checkpoint of another architecture is refused by the strict load (as in the reference)
    bad = dict(ckpt)
    bad['hyper_parameters'] = dict(ckpt['hyper_parameters'], ENCODER='resnet18')
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        L.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(str(path), config.INFERENCE())


def test_encoder_weights_from_local_file(tmp_path, monkeypatch, caplog):
    """ENCODER_WEIGHTS='imagenet' (every preset): smp downloads the torchvision checkpoint; here it is read from a
    configured local file (fc.* dropped), and without one the random initialisation is announced, not silent."""
    import logging
    import backbone as bb
    torch.manual_seed(1)
    donor = bb.ResNetEncoder('resnet18')
    tv = {k: v.clone() for k, v in donor.state_dict().items()}
    tv['fc.weight'] = torch.zeros(1000, 512); tv['fc.bias'] = torch.zeros(1000)      # torchvision's classifier
    torch.save(tv, tmp_path / "resnet18.pth")
    monkeypatch.setenv("FPC_ENCODER_WEIGHTS_DIR", str(tmp_path))
    torch.manual_seed(2)
    enc = bb.get_encoder('resnet18', weights='imagenet')
    assert enc.loaded_weights and all(torch.equal(v, donor.state_dict()[k]) for k, v in enc.state_dict().items())
    monkeypatch.delenv("FPC_ENCODER_WEIGHTS_DIR")
    bb._WARNED_WEIGHTS.clear()
    with caplog.at_level(logging.WARNING, logger='fastposecnn'):
        enc = bb.get_encoder('resnet18', weights='imagenet')
    assert enc.loaded_weights is None and "RANDOMLY initialised" in caplog.text
    assert bb.get_encoder('resnet18', weights=None).requested_weights is None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (the way the driver starts it) must spawn its N ranks itself: fresh child processes
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's JSON line relayed, exit code propagated.  Dry run:
    gloo rendezvous and one all-reduce instead of the GPU work."""
    import json
    import subprocess
    env = dict(os.environ, FPC_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE line, from rank 0 only
    rec = json.loads(lines[0])
    base = {k: rec[k] for k in ("dryrun", "n_gpus", "rank_sum", "local_rank", "master")}
    assert base == {"dryrun": True, "n_gpus": 2, "rank_sum": 3.0, "local_rank": 0, "master": "127.0.0.1"}
    # an N > 1 line describes itself: measured, every rank's own rate, the ranks the backend's all-reduce summed over
    assert rec["scaling_measured"] is True and rec["rccl_ranks_seen"] == 2 and rec["collective_backend"] == "gloo"
    assert rec["per_rank_img_per_s"] == [200.0, 100.0]
    ncpu = len(os.sched_getaffinity(0))
    assert rec["cores_per_rank"] == (ncpu // 2 if ncpu >= 2 else None)
    # under an external launcher (WORLD_SIZE already set) the script is a rank, it does not spawn again
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"], env=env1, capture_output=True,
                         text=True, timeout=300)
    one = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0 and one["n_gpus"] == 1 and one["scaling_measured"] is False
    # a mismatch between --gpus and the launcher's world size is refused
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env1, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0


def test_training_conv_helpers_fall_back_to_torch_on_cpu():
    """lib/train_conv.py on CPU tensors: plain torch results and gradients (the native kernels are GPU-only and are never
    substituted silently: the GPU tests assert that they ran)."""
    import torch
    import fastposecnn_amd.lib  # noqa: F401
    from fastposecnn_amd.lib import train_conv, backbone
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 32, 9, 11), generator=g, requires_grad=True)
    conv = backbone.Conv2d(32, 8, 3, 1, 1).train()
    y = conv(x)
    assert torch.equal(y, torch.nn.functional.conv2d(x, conv.weight, conv.bias, 1, 1))
    before = dict(train_conv.counters)
    y.sum().backward()
    assert train_conv.counters == before and x.grad is not None
    up = train_conv.upsample_bilinear(x.detach(), 2)
    assert torch.equal(up, torch.nn.functional.interpolate(x.detach(), scale_factor=2, mode="bilinear", align_corners=True))
    gn = torch.nn.GroupNorm(8, 32)
    assert torch.equal(train_conv.groupnorm_relu(x.detach(), gn), torch.relu(gn(x.detach())))


def test_mask_bits_ride_on_the_very_tensor_only():
    import torch
    import fastposecnn_amd.lib  # noqa: F401
    import aggregation_layer as al
    m = torch.zeros((3, 4, 5))
    assert al.mask_bits_of(m) is None
    bits = torch.zeros((3, 64), dtype=torch.int64)
    m._fpc_mask_bits = (bits, m._version)
    assert al.mask_bits_of(m) is bits
    assert al.mask_bits_of(m[:2]) is None and al.mask_bits_of(m.clone()) is None
    m.add_(1.0)
    assert al.mask_bits_of(m) is None            # written since: the words may be stale


def test_isa_lint_flags_salu_vcc_before_div_fmas():
    """fastposecnn_amd/isa_lint.py: the pattern behind the streaming-runtime hypothesis mismatch."""
    from fastposecnn_amd import isa_lint
    paired = ["v_div_scale_f32 v7, vcc, v16, v8, v16", "v_div_scale_f32 v20, s[0:1], v17, v9, v17",
              "v_fma_f32 v4, -v4, v21, v7", "v_div_fmas_f32 v4, v4, v18, v21", "s_mov_b64 vcc, s[0:1]",
              "v_div_fixup_f32 v18, v4, v8, v16", "v_div_fmas_f32 v4, v7, v19, v22"]
    hits = isa_lint.scan(paired)
    assert [(h[0], h[1]) for h in hits] == [(2, "s_mov_b64 vcc, s[0:1]")]
    alone = ["v_div_scale_f32 v4, vcc, v8, v8, v16", "v_rcp_f32_e32 v18, v4", "v_div_scale_f32 v7, vcc, v16, v8, v16",
             "v_fma_f32 v21, -v4, v18, 1.0", "v_div_fmas_f32 v4, v4, v18, v21"]
    assert isa_lint.scan(alone) == []
    far = ["s_mov_b64 vcc, s[0:1]"] + ["v_mov_b32_e32 v1, v2"] * isa_lint.WINDOW + ["v_div_fmas_f32 v4, v7, v19, v22"]
    assert isa_lint.scan(far) == []


def test_isa_lint_built_library_is_clean():
    from fastposecnn_amd import isa_lint
    if not os.path.exists(isa_lint.LIB) or not os.path.exists(isa_lint.OBJDUMP):
        pytest.skip("libfpc_hip.so or llvm-objdump not present")
    bad, n_div = isa_lint.findings()
    assert n_div > 0 and bad == []


def test_cpu_baseline_leaves_thread_settings_alone_and_names_its_threads():
    """the oracle's thread count used to be set with omp_set_num_threads — process-wide, shared with
    torch's CPU kernels — and left at 1, so the SECOND cpu_baseline() of a bench run (config 3, the top-level record) timed a
    one-thread network under a "cores: 16" label.  Now: torch's count is the same before and after each of two calls, the
    oracle's setting is its own, both legs are timed on 1 thread and on the job's CPU share, and `cores` is a count the
    dominant leg really used."""
    import importlib
    import numpy as np
    import torch
    bench = importlib.import_module("bench")
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth
    from oracle import oracle as orc
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    torch.manual_seed(0)
    model = L.pose_regressor.MODELS["PoseRegressor"].load_from_ckpt(None, hp).eval()
    image = synth.make_image(0, 64, 96)[None]
    cat_cpu, _ = synth.make_vote_frame(0, K=2, H=64, W=96, rmin=6, rmax=14)
    inv_k = np.linalg.inv(hp.NUMPY_INTRINSICS).astype(np.float32)
    share = bench.cpu_share()
    assert 1 <= share <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    orc_before = orc.get_threads()
    for _ in range(2):
        rec = bench.cpu_baseline(model, image, cat_cpu, 32, inv_k, "resnet18")
        assert torch.get_num_threads() == before
        assert orc.get_threads() == orc_before
        assert rec["torch_threads_before_after"] == [before, before]
        assert rec["cpu_share"] == share
        assert rec["threads"]["net"] in (1, share) and rec["threads"]["post"] in (1, min(share, 16))
        assert rec["cores"] in (rec["threads"]["net"], rec["threads"]["post"])
        assert set(rec["net_ms_by_threads"]) == {str(k) for k in {1, share}}
        assert f"on {rec['threads']['net']} thread(s)" in rec["sample"]
    # the oracle's own setting never reaches the process-wide OpenMP count
    orc.set_threads(1)
    assert torch.get_num_threads() == before
    orc.set_threads(orc_before)


def test_zero_edit_overlay_answers_import_lib(tmp_path):
    """with fastposecnn_amd/overlay on PYTHONPATH a script that says `import lib` — from a directory
    that has its OWN lib/ package next to it, as the reference's scripts do (F/inference.py:19) — gets this repo's mirror, with
    the reference's module aliases; FPC_OVERLAY=0 gives the script's own package back."""
    (tmp_path / "lib").mkdir()
    (tmp_path / "lib" / "__init__.py").write_text("WHO = 'the script directory own lib'\n")
    script = tmp_path / "inference_like.py"
    script.write_text("import lib\n"
                      "print(getattr(lib, 'WHO', None) or lib.pose_regressor.__file__)\n"
                      "print(sorted(n for n in ('gtf', 'mg', 'pose_regressor', 'loss', 'metrics') if hasattr(lib, n)))\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "fastposecnn_amd", "overlay"))
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    first, second = out.stdout.strip().splitlines()[-2:]
    assert os.path.realpath(first) == os.path.realpath(os.path.join(REPO, "fastposecnn_amd", "lib", "pose_regressor.py")), out.stdout
    assert second == "['gtf', 'loss', 'metrics', 'mg', 'pose_regressor']"
    off = subprocess.run([sys.executable, str(script)], env=dict(env, FPC_OVERLAY="0"), capture_output=True, text=True, timeout=300,
                         cwd=str(tmp_path))
    assert off.returncode == 0 and "the script directory own lib" in off.stdout, off.stdout + off.stderr[-500:]
