"""`python bench.py --train [--gpus N]`: BASELINE.json configs[4] — one training step of ResNet18-FPN + all heads at
batch 8 per GPU (64 on 8 GPUs), forward + HEAD_TRAINING losses + backward + gradient reduction + optimiser step.

What runs where (stated in the JSON line as well):
  * encoder / decoder / head convolutions: lib/train_conv.py — forward on the engine's implicit-GEMM / Winograd kernels,
    data gradient on the same kernels (stride 1: flipped weights; stride 2: four parity convolutions of dy; heads of odd
    width: zero-padded to 32 channels), weight gradient on csrc/conv_wgrad.hip; torch keeps the 7x7 stem (Cin = 3);
    bilinear upsampling forward / backward on csrc/upsample.hip; BatchNorm / GroupNorm / ReLU / adds: torch
    (FPC_TRAIN_NATIVE_CONV=0 returns every convolution and upsampling to torch: the A/B this file's numbers come with);
  * everything after the logits, forward: the inference kernels (class compression, connected components, aggregation,
    RANSAC vote, RT); backward: csrc/train.hip through lib/train_functions.py;
  * matching (fpc_mask_iou) and the loss arithmetic of F/lib/pose_regressor.py:188-307 with lib/loss.py;
  * gradients: bucketed RCCL reduce-scatter overlapped with backward, sharded native Lookahead(RAdam) step with clipping
    and the inf/NaN guard on device scalars, all-gather of the parameters (fastposecnn_amd/train_parallel.py).

Synthetic data: the images are the reference's input recipe on uniform noise; ground truth is the vote-bench fixture
(6 instances per frame).  A random-weight network predicts no usable instances, which would leave the post-network
path, the matching and four of the five loss groups idle, so a CONSTANT logit offset derived from the fixture is added to
the network's logits: the arg-max mask, the votes and the regression planes then look like a trained model's (noise
included) while every gradient still flows through the addition into the network.  The step does all of its work.
"""
import json
import os
import time

import torch
import torch.distributed as dist


def _fixture_logit_offsets(cat, G=6, gain=12.0, mask_gain=40.0):
    """Constant logits that make class compression reproduce the fixture `cat` (lib/gpu_tensor_funcs.class_compress).
    The mask margin is wide enough that the random-weight network's own mask logits never flip a pixel: the predicted
    instances stay the fixture's while the weights move (with a margin of 12 the count drifted from 78 to 544 specks
    over 20 optimiser steps and the matching / loss stage with it)."""
    B, H, W = cat["mask"].shape
    onehot = torch.nn.functional.one_hot(cat["mask"], G + 1).permute(0, 3, 1, 2).float()
    off = {"mask": onehot * mask_gain}
    sel = torch.nn.functional.one_hot((cat["mask"] - 1).clamp(min=0), G).permute(0, 3, 1, 2).unsqueeze(2).float()
    sel = sel * (cat["mask"] != 0).float().view(B, 1, 1, H, W)
    for key, a in (("quaternion", 4), ("scales", 3), ("xy", 2), ("z", 1)):
        v = cat[key] if key != "z" else cat[key].unsqueeze(1)
        off[key] = (sel * v.unsqueeze(1) * (gain if key in ("quaternion", "xy") else 1.0)).reshape(B, G * a, H, W)
    return off


def _ground_truth(model, cat, symmetric_classes=(1, 2, 4)):
    """AggData of the fixture (the keys matching.batchwise_find_matches stacks) through the inference path."""
    with torch.no_grad():
        agg = model.agg_hough_and_generate_RT({k: v.clone() for k, v in cat.items()})
    gt = {k: agg[k].clone() for k in ("class_ids", "sample_ids", "instance_masks", "quaternion", "scales", "xy", "z", "R", "T", "RT")}
    sym = torch.zeros_like(gt["class_ids"])
    for c in symmetric_classes:             # bottle, bowl, can (F/tools/project.py: symmetric classes of the CAMERA set)
        sym |= (gt["class_ids"] == c).long()
    gt["symmetric_ids"] = sym
    return gt


def _conv_note():
    from fastposecnn_amd.lib import train_conv
    if not train_conv.ENABLED:
        return "torch modules (MIOpen / rocBLAS) forward and backward (FPC_TRAIN_NATIVE_CONV=0)"
    c = train_conv.counters
    return ("native: forward fpc_conv2d (implicit GEMM / Winograd), data gradient on the same kernels (stride 2: four parity "
            "convolutions of dy; odd-width heads zero-padded), weight gradient fpc_conv2d_wgrad, bilinear upsampling "
            "fpc_upsample_bilinear_fwd/bwd; torch for the 7x7 stem alone (Cin = 3: outside these counters) "
            f"(calls so far: {dict(c)})")


def main(args):
    run(args)


def run(args, quiet=False):
    """The config-4 step.  quiet: return rank 0's line (bench.py's `train` object) instead of printing it; the process
    group, if any, is left to the caller."""
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth, _native
    from fastposecnn_amd.train_parallel import ShardedLookaheadRAdam
    import loss as loss_lib
    import matching as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FPC_BENCH_BACKEND", "nccl")        # "gloo": lets two ranks share one GPU in a smoke test
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    _native.lib()

    Bq = args.train_batch
    # the step's host work is a few dozen tiny CPU tensor ops (matching's arg-max on an [n1,n2] matrix, scalar NaN tests):
    # torch's intra-op pool (one thread per core, 256 here) only adds wake-up stalls to them
    torch.set_num_threads(int(os.environ.get("FPC_TRAIN_CPU_THREADS", "4")))
    # MIOpen's find mode (FPC_TRAIN_MIOPEN_FIND=1) was measured: same step time (59.3 ms either way) after minutes of search
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("FPC_TRAIN_MIOPEN_FIND", "0")))
    hp = config.HEAD_TRAINING()
    hp.ENCODER = args.encoder
    hp.RUNTIME_TIMING = False
    torch.manual_seed(0)
    model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).to(dev).train()
    image = torch.stack([synth.make_image(rank * Bq + i) for i in range(Bq)]).to(dev)
    cat_cpu, _ = synth.make_vote_batch(range(rank * Bq, rank * Bq + Bq))
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    batch = {"image": image, "mask": cat["mask"], "agg_data": _ground_truth(model, cat)}
    offsets = {k: v.to(dev) for k, v in _fixture_logit_offsets(cat_cpu).items()}
    net_forward = model.pure_model_forward

    def forward_with_offsets(x):
        logits = net_forward(x)
        return {k: v + offsets[k] for k, v in logits.items()}

    model.pure_model_forward = forward_with_offsets
    criterion = loss_lib.head_training_criterion()
    opt = ShardedLookaheadRAdam(model, lr=1e-5, weight_decay=3e-4, clip_norm=0.15, bucket_mb=args.bucket_mb)
    n_param = sum(p.numel() for p in model.parameters() if p.requires_grad)

    ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(2)] for k in ("fwd", "loss", "bwd", "opt")}
    last = {}

    def step(timed=False):
        opt.zero_grad()
        if timed: ev["fwd"][0].record()
        out = model(batch["image"])
        if timed: ev["fwd"][1].record(); ev["loss"][0].record()
        matches = mg.batchwise_find_matches(out["aggregated"], batch["agg_data"])
        total, report = loss_lib.total_loss(criterion, out, batch, matches)
        if timed: ev["loss"][1].record(); ev["bwd"][0].record()
        total.backward()
        if timed: ev["bwd"][1].record(); ev["opt"][0].record()
        opt.step()
        if timed: ev["opt"][1].record()
        last["total"], last["report"], last["matches"] = total.detach(), report, matches
        last["n_pred"] = int(out["aggregated"]["class_ids"].shape[0])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(2, args.warmup)):
        step()
    barrier()
    t0 = time.perf_counter()
    trace = []
    for _ in range(args.steps):
        step()
        if os.environ.get("FPC_TRAIN_TRACE"):
            torch.cuda.synchronize()
            trace.append(round((time.perf_counter() - t0) * 1e3, 1))
    if trace and rank == 0:
        print("cumulative ms per step:", trace, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # stage breakdown of one more step (HIP events on the compute stream; the reduce-scatters overlap `bwd`)
    step(timed=True)
    torch.cuda.synchronize()
    stages = {k: round(a.elapsed_time(b), 3) for k, (a, b) in ev.items()}
    matched = 0 if last["matches"] is None else int(last["matches"]["class_ids"].shape[0])
    in_sync = True
    if world > 1:       # every rank must hold the same parameters after the all-gathers
        cs = opt.flat_p.double().sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool((hi == lo).item())
    if rank == 0:
        losses = {t: {k: (None if bool(torch.isnan(v)) else round(float(v.detach()), 6)) for k, v in d.items()}
                  for t, d in last["report"].items()}
        line = {
            "metric": "img/s train step (fwd + losses + bwd + gradient reduction + optimiser) 640x480",
            "value": round(world * Bq * args.steps / dt, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(2, args.warmup), "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.encoder}-FPN + all heads, HEAD_TRAINING losses, batch={Bq} 640x480 per GPU per step, "
                                   f"hn={hp.HV_NUM_OF_HYPOTHESES}, 6 instances per frame (vote-bench fixture as ground truth; a constant "
                                   f"logit offset from the fixture stands in for trained weights), random-init weights",
                       "global_batch": world * Bq,
                       "parallelism": (f"dp{world}: bucketed reduce-scatter (overlapped with backward) + sharded optimiser + "
                                       f"parameter all-gather over RCCL") if world > 1 else "single GPU",
                       "trainable_parameters": n_param, "gradient_bytes": 4 * opt.total, "buckets": len(opt.buckets),
                       "optimizer": "Lookahead(RAdam) k=5 alpha=0.5, lr 1e-5, weight decay 3e-4, clip 0.15; native shard kernel",
                       "optimizer_state_bytes_per_rank": opt.state_bytes(),
                       "convolutions": _conv_note(),
                       "post_network": "HIP kernels forward (inference path) and backward (csrc/train.hip)"},
            "stages_ms": stages,
            "step_check": {"total_loss": round(float(last["total"]), 6), "losses": losses, "predicted_instances": last["n_pred"],
                           "matched_instances": matched, "skipped_steps": int(opt.skipped), "replicas_in_sync": in_sync},
        }
        if quiet:
            return line
        print(json.dumps(line), flush=True)
    if world > 1 and not quiet:
        dist.destroy_process_group()
