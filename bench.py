#!/usr/bin/env python3
"""bench.py — end-to-end 640x480 inference throughput (img/s) + hough-vote kernel roofline.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py ...`)

Workload (BASELINE.json configs[1]): ResNet18-FPN + all four heads, batch = 1 frame of 640x480
per GPU per step, HV_NUM_OF_HYPOTHESES = 1000 (config.INFERENCE), random-init weights
(torch.manual_seed(0)), synthetic data, f32 throughout.  One step is one pass of the hot path over
one frame:

    image -> native engine (fpc_net_forward): encoder -> 4 FPN decoders -> 4 heads -> x4 upsample
             -> class compression                                           (on the synthetic image)
          -> aggregation (CC + per-instance means) -> RANSAC hough voting -> RT (on the synthetic
             post-network "vote bench" frame: 6 elliptical instances — random-init weights give no
             usable instances)
          -> [N > 1] RCCL all-gather of the per-instance pose records

All inputs are resident in HBM before the timed region.  Frames are streamed with five in flight
(fastposecnn_amd/streaming.py: consecutive frames go round-robin to four native plans on their own HIP
streams — one per hardware compute pipe — each frame's post-network stages follow on the same stream;
every frame completes all of its work, it is only collected four submissions later) — `--no-pipeline`
finishes each frame before starting the next and `config.ms_per_frame_one_in_flight` reports that latency.  Weak scaling: every rank runs its own
frame per step; `value` = N * K / max-over-ranks(time).

Extra objects on the JSON line:
  roofline      the hough-vote launch sequence (fpc_ransac_voting_v3): algorithmic bytes
                n_instances * 12*H*W per call / HIP-event time of the call on its stream,
                against the 8 TB/s HBM peak of MI355X_MICROARCH.md; `traffic` = PMC-measured HBM bytes
                per launch (profiles/r01_vote_traffic.json: FETCH_SIZE doubled per the guide + WRITE_SIZE)
  backbone      the network part: algorithmic f32 FLOP of the direct convolutions (100.1 GFLOP/frame,
                ResNet18) / HIP-event time, against the 157.3 TFLOP/s f32 matrix-core peak
  cpu_baseline  the same step on the host: torch-CPU backbone + the C oracle's post-network path
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # 4 frame streams + null stream, before HIP initialises (streaming.py)

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak (155 measured)
BACKBONE_GFLOP = {"resnet18": 2 * (11.10 + 4 * 9.70 + 0.16), "resnet34": 2 * (22.43 + 4 * 9.70 + 0.16)}
H, W = 480, 640


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--hn", type=int, default=1000, help="HV_NUM_OF_HYPOTHESES (1000 = config.INFERENCE)")
    ap.add_argument("--encoder", default="resnet18")
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step (1 = BASELINE.json configs[1]; 32 = configs[2]/[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vote-only", action="store_true", help="time only the post-network stages (profiling aid)")
    ap.add_argument("--tune-mode", type=int, default=0, help="conv autotune objective: 0 latency, 1 latency x sqrt(chip share)")
    ap.add_argument("--net-streams", type=int, default=4, help="frame streams: native plans on their own HIP streams that take consecutive frames")
    ap.add_argument("--post-stream", action="store_true", help="run the post-network stages of all frames on one extra stream instead of the frame's own")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every frame before starting the next (latency mode)")
    return ap.parse_args()


def cpu_baseline(model_cpu, image, cat_cpu, hn, inv_k):
    """One frame on the host: torch CPU backbone (+class compression) and the oracle's C
    restatement of aggregation / voting / RT.  Bounded: one frame, a few seconds."""
    from oracle import oracle as orc
    orc.build()
    threads = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.perf_counter()
        logits = model_cpu.pure_model_forward(image)
        model_cpu.class_compression(logits)
        t_net = time.perf_counter() - t0
    cat_np = {k: v.numpy() for k, v in cat_cpu.items()}
    t0 = time.perf_counter()
    agg = orc.aggregate(cat_np)
    vertex = agg["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :]
    xy = orc.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=1)
    orc.pose_rt(agg["quaternion"], xy[:, 0], agg["z"], inv_k)
    t_post = time.perf_counter() - t0
    return {"value": round(1.0 / (t_net + t_post), 4), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"1 frame: torch-CPU ResNet18-FPN forward + class compression on {threads} threads "
                      f"({t_net * 1e3:.0f} ms) + oracle/fpc_oracle.c aggregation, hn={hn} voting and RT on 1 thread "
                      f"({t_post * 1e3:.0f} ms); host has {os.cpu_count()} logical CPUs",
            "net_ms": round(t_net * 1e3, 1), "post_ms": round(t_post * 1e3, 1)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))    # (several ranks on one GPU only in tests)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FPC_BENCH_BACKEND", "nccl")        # "gloo": lets two ranks share one GPU in a smoke test
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth, parallel, _native
    _native.lib()      # fail loudly if the HIP library is missing

    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = args.hn
    hp.ENCODER = args.encoder
    hp.ENGINE_TUNE_MODE = args.tune_mode
    hp.ENGINE_SPLIT_PRECISION = bool(int(os.environ.get('FPC_SPLIT_PRECISION', '0')))      # opt-in experiment
    hp.ENGINE_GRAPH = bool(int(os.environ.get('FPC_ENGINE_GRAPH', '1')))      # HIP graph replay of the frame-invariant launches
    torch.manual_seed(0)
    model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval()
    model_gpu = model.to(dev)
    Bq = args.batch
    image = torch.stack([synth.make_image(rank * Bq + i) for i in range(Bq)])      # per-rank frames (weak scaling)
    cat_cpu, _ = synth.make_vote_batch(range(rank * Bq, rank * Bq + Bq))
    x = image.to(dev)
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    n_inst = 6 * Bq                                           # vote-bench fixture: 6 instances per frame
    cap = 64 * Bq

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    vote_ms = []

    # Frame-streaming runtime (fastposecnn_amd/streaming.py): consecutive frames go round-robin to
    # `--net-streams` native plans on their own HIP streams (network + post-network stages of a frame in order).
    # Every frame does all of its work and is complete (instance count read back, tensors trimmed) when it
    # is collected, `depth` submissions later.
    from fastposecnn_amd.streaming import FrameStreamer
    streamer = FrameStreamer(model_gpu, net_streams=1 if args.no_pipeline else args.net_streams,
                             post_inline=not args.post_stream)
    s_net = streamer.net_streams[0]
    depth = 0 if args.no_pipeline else len(streamer.models)
    pending = []
    gather_buf = [None]

    def finish(ticket):
        out = {"aggregated": model_gpu.post_network_finish(ticket)} if args.vote_only else streamer.collect(ticket)
        if world > 1:      # ONE fixed-capacity RCCL all-gather of pose records per step: pack = one native launch
            gather_buf[0] = parallel.all_gather_pose_records(out["aggregated"], rank * Bq, cap, out=gather_buf[0])
        return out

    def step(pipelined=True):
        """One step = one batch of frames through the whole hot path (network on the image, post-network on
        the vote-bench fixture)."""
        if args.vote_only:
            with torch.no_grad():
                pending.append(model_gpu.post_network_enqueue(cat))
        else:
            pending.append(streamer.submit(x, categorical_override=cat))
        if len(pending) > (depth if pipelined else 0):
            return finish(pending.pop(0))
        return None

    def drain():
        while pending:
            finish(pending.pop(0))

    def vote_probe():
        """HIP events around the hough-voting enqueue alone (its inputs produced just before)."""
        with torch.no_grad():
            agg = model_gpu.aggregate(cat)
            ev[0].record()
            agg = model_gpu.hough_voting(agg)
            ev[1].record()
        return agg

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: every stream's plan is built and autotuned here (seconds), whatever --warmup says
    if not args.vote_only:
        streamer.prepare(x, categorical_override=cat)
    for _ in range(2 * (depth + 1)):       # ... and every stream's allocator pool has seen a full pipeline of frames
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-frame latency with ONE frame in flight (not the headline number)
    nlat = max(5, min(args.steps, 20))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(nlat):
        step(pipelined=False)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / nlat * 1e3

    # backbone alone: HIP events around the engine call on its stream
    net_ms = []
    if not args.vote_only:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(max(5, min(args.steps, 30))):
            with torch.no_grad(), torch.cuda.stream(s_net):
                e0.record()
                logits = model_gpu.pure_model_forward(x)
                model_gpu.class_compression(logits)
                e1.record()
            e1.synchronize()
            net_ms.append(e0.elapsed_time(e1))
        net_ms.sort()

    # vote roofline: separate, untimed-for-throughput loop with HIP events around the vote call
    for _ in range(max(5, min(args.steps, 30))):
        vote_probe()
        ev[1].synchronize()
        vote_ms.append(ev[0].elapsed_time(ev[1]))
    vote_ms.sort()
    vote_t = vote_ms[len(vote_ms) // 2] * 1e-3
    alg_bytes = n_inst * 12 * H * W
    achieved = alg_bytes / vote_t / 1e9
    traffic = None
    tpath = os.path.join(REPO, "profiles", "r01_vote_traffic.json")
    if os.path.exists(tpath) and args.hn == 1000:
        with open(tpath) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")

    # measured device-to-device copy ceiling on this GPU (read + write bytes), beside the 8 TB/s spec
    copy_gbps = None
    try:
        src = torch.empty(256 << 20, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record(); c1.synchronize()
        copy_gbps = round(10 * 2 * src.numel() / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        del src, dst
    except Exception:
        pass

    if rank == 0:
        line = {
            "metric": "img/s end-to-end 640x480 inference; hough-vote kernel HBM GB/s vs roofline",
            "value": round(world * Bq * args.steps / dt, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.encoder}-FPN + all heads, batch={Bq} 640x480 per GPU, hn={args.hn}, "
                                   f"{n_inst} instances/frame (vote-bench fixture), random-init weights",
                       "global_batch": world * Bq, "parallelism": f"image-sharded dp{world}" if world > 1 else "single GPU",
                       "vote_only": bool(args.vote_only),
                       "frames_in_flight": 1 + depth, "net_streams": len(streamer.models),
                       "ms_per_frame_one_in_flight": round(latency_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": "fpc_ransac_voting_v3 launch sequence (k_chunk_count .. k_count_hi .. k_refine)",
                         "algorithmic_bytes_per_launch": alg_bytes, "measured_copy_GBps": copy_gbps, "launch_ms": round(vote_t * 1e3, 4),
                         "note": "HIP events on the launch stream around the whole call; at hn=1000 the count "
                                 "kernel is VALU-bound (DESIGN.md); traffic from profiles/r01_vote_traffic.json (PMC)"},
        }
        if net_ms:
            t_net = net_ms[len(net_ms) // 2] * 1e-3
            tf = Bq * BACKBONE_GFLOP.get(args.encoder, 0.0) / t_net / 1e3
            line["backbone"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "ms": round(t_net * 1e3, 4),
                                "algorithmic_gflop_per_frame": round(BACKBONE_GFLOP.get(args.encoder, 0.0), 2),
                                "note": "direct-convolution FLOP count; the engine runs the large 3x3 layers as "
                                        "Winograd F(2x2,3x3) on f32 MFMA (2.25x fewer multiply-adds)"}
        if world == 1 and not args.no_cpu_baseline:
            one = {k: v[:1] for k, v in cat_cpu.items()}
            line["cpu_baseline"] = cpu_baseline(model.to("cpu"), image[:1], one, args.hn,
                                                torch.inverse(torch.from_numpy(hp.NUMPY_INTRINSICS).float()).numpy())
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
