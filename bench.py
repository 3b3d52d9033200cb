#!/usr/bin/env python3
"""bench.py — end-to-end 640x480 inference throughput (img/s) + hough-vote kernel roofline.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started plainly (`python bench.py --gpus 8`) the script spawns its N ranks itself
(fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the parent never touches a GPU and relays
rank 0's JSON line); started under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is
a rank.  RCCL ("nccl") carries the one collective of the path: the all-gather of per-instance pose records.

Workload of the headline line (BASELINE.json configs[1]): ResNet18-FPN + all four heads, batch = 1 frame of 640x480
per GPU per step, HV_NUM_OF_HYPOTHESES = 1000 (config.INFERENCE), random-init weights (torch.manual_seed(0)),
synthetic data, f32 operands / accumulation / results (product forms: config.matrix_products).  One step is one pass of the hot path over one frame:

    image -> native engine (fpc_net_forward): encoder -> 4 FPN decoders -> 4 heads -> x4 upsample
             -> class compression                                           (on the synthetic image)
          -> aggregation (CC + per-instance means) -> RANSAC hough voting -> RT (on the synthetic
             post-network "vote bench" frame: 6 elliptical instances — random-init weights give no
             usable instances; SURVEY.md 8d sanctions the fixture)
          -> [N > 1] RCCL all-gather of the per-instance pose records

All inputs are resident in HBM before the timed region.  Frames are streamed with five in flight
(fastposecnn_amd/streaming.py: consecutive frames go round-robin to four native plans on their own HIP streams, each
frame's post-network stages follow on the same stream; every frame completes all of its work, it is only collected
four submissions later); `config.ms_per_frame_one_in_flight` is the latency with one frame in flight.  Weak scaling:
every rank runs its own frames; `value` = N * B * K / max-over-ranks(time).

Extra objects on the JSON line:
  roofline          the hough-vote launch sequence at the headline config, as the model's pipeline calls it (the aggregation
                    layer's mask bit words: fpc_ransac_voting_v3_bits): algorithmic bytes n_instances * 12*H*W per call /
                    HIP-event device time of 10 calls on cold inputs (graph replay), against the 8 TB/s HBM peak of MI355X_MICROARCH.md;
                    `traffic` and `valu` are PMC counters of a separately profiled run (`from_profile` names file and commit:
                    profiles/r04_vote_bits_traffic_*.json)
  roofline_hn128    the same sequence at the training value hn = 128 (F/config.py:93) on a 32-frame batch (192 instances)
  roofline_hn128_f32_masks  ... and through the reference's own interface: f32 mask planes, no bit words (every algorithmic byte read)
  post_network      connected components and aggregation at B = 1 and B = 32: time, algorithmic bytes, fraction of 8 TB/s
  backbone          the network part: executed f32-equivalent multiply-add FLOP of the engine's plans (Winograd sites count
                    1/2.25) / device time per network, against the 157.3 TFLOP/s f32 matrix peak (and, second figure,
                    against the bf16 x 3 equivalent peak: config.matrix_products says which product form the plans use)
  configs.config3   BASELINE.json configs[2]: ResNet34, batch 32 per step, same pipeline (shorter timed region)
  frames_per_launch  the headline stream (one frame per step) with FrameStreamer(coalesce=2 / 4): the runtime groups consecutive
                    frames into one engine launch (dynamic batching; informational, never `value`)
  plain_f32_products  the headline measurement once more with split-precision products switched off (1-GPU runs)
  train             BASELINE.json configs[4] at batch 8 on this GPU (fastposecnn_amd/train_bench.py; 1-GPU runs)
  cpu_baseline      the same step on the host: torch-CPU backbone + the C oracle's post-network path (1 thread and the
                    job's CPU share), three samples each
`value` is the MEDIAN of the repeated K-step timed regions (`repeats`, `ms_per_step_min_max`).  `scaling_measured` is true
exactly when this line comes from an N > 1 run; such a line also carries `per_rank_img_per_s`, `rccl_ranks_seen` (what an
all-reduce of ones over the job's backend summed to), `collective_backend`, `cores_per_rank` (each rank pins itself to its
own share of the host cores before touching the GPU) and `config.pose_gather.us_per_collective`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # 4 frame streams + null stream, before HIP initialises (streaming.py)

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak (155 measured)
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2      # 1024 SIMD-32 units, a wave64 instruction issues over 2 cycles (the guide)
H, W = 480, 640


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 300; 20 with --train)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 20; 3 with --train)")
    ap.add_argument("--hn", type=int, default=1000, help="HV_NUM_OF_HYPOTHESES (1000 = config.INFERENCE)")
    ap.add_argument("--encoder", default=None, help="default: resnet34 at the top level with ResNet18 / batch 1 under configs.config2 (see --batch)")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per GPU per step.  Neither --encoder nor --batch given: the top-level record is BASELINE.json configs[2] "
                         "(ResNet34, batch 32: the largest single-GPU configuration; per rank it is configs[3]'s share at N > 1) and, at "
                         "N = 1, configs[1] (ResNet18, batch 1) with its side measurements sits under configs.config2.  Either flag given: "
                         "that one configuration at the top level, as before round 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the ResNet34 batch-32 section (configs[2])")
    ap.add_argument("--no-batch-scan", action="store_true", help="skip the 2 / 4 frames-per-launch section")
    ap.add_argument("--no-hn128", action="store_true", help="skip the hn=128 / 32-frame vote roofline")
    ap.add_argument("--no-plain-f32", action="store_true", help="skip the second streamed measurement with plain f32 matrix products")
    ap.add_argument("--no-train-line", action="store_true", help="skip the `train` object (configs[4] at B=8 on this GPU, 1-GPU runs only)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="the K-step timed region is repeated until this much time is covered; `value` is the median repeat")
    ap.add_argument("--gather-every", type=int, default=0, help="frames per pose all-gather on the side stream (0 = frames in flight)")
    ap.add_argument("--check-gather", action="store_true", help="N > 1: verify the gathered pose records against every rank's records rebuilt locally")
    ap.add_argument("--vote-only", action="store_true", help="time only the post-network stages (profiling aid)")
    ap.add_argument("--tune-mode", type=int, default=0, help="conv autotune objective of the model's own plan (the `backbone` object): 0 latency, 1 latency x sqrt(chip share)")
    ap.add_argument("--tune-trials", type=int, default=3, help="headline stream: build the runtime's set of plans this many times and keep the set that streams fastest (set-up, untimed)")
    ap.add_argument("--stream-tune-mode", type=int, default=1, help="the same for the frame-streaming runtime's plans (several frames in flight); -1: the runtime's first plan is the model's own")
    ap.add_argument("--net-streams", type=int, default=4, help="frame streams: native plans on their own HIP streams that take consecutive frames")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frames submitted before the oldest is collected (0 = frame streams + 1)")
    ap.add_argument("--post-stream", action="store_true", help="run the post-network stages of all frames on one extra stream instead of the frame's own")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every frame before starting the next (latency mode)")
    ap.add_argument("--train", action="store_true", help="BASELINE.json configs[4]: train step (fwd + losses + bwd + gradient reduction + optimiser)")
    ap.add_argument("--train-batch", type=int, default=8, help="frames per GPU per training step (8 x 8 GPUs = configs[4]'s 64)")
    ap.add_argument("--bucket-mb", type=float, default=16.0, help="gradient bucket size of the training step")
    args = ap.parse_args(argv)
    args.promote = args.encoder is None and args.batch is None and not args.train and not args.vote_only
    if args.encoder is None:
        args.encoder = "resnet18"
    if args.batch is None:
        args.batch = 1
    if args.steps is None:
        args.steps = 20 if args.train else 300
    if args.warmup is None:
        args.warmup = 3 if args.train else 20
    return args


# --------------------------------------------------------------------------------------------------- launcher

def launch_ranks(n, argv):
    """Spawn n fresh child processes (one per GPU) of this script and relay rank 0's output.  The parent has not
    touched the GPU (no torch import yet), and nothing is exec'ed over an initialised process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    return rc


def cpu_share():
    """The CPUs this job may use: the smaller of the scheduler affinity mask and the cgroup's CPU bandwidth quota
    (v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  A GPU box reports 256 logical CPUs to `os.cpu_count()` while one
    GPU's job owns 16 of them: thread pools sized by the former burn the quota in spin-waits and get the whole process
    throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


def pin_rank_to_cores(local_rank, local_world):
    """Confine this rank to its share of the host cores (before torch starts its pools and before anything touches the GPU;
    no re-exec).  N > 1: a contiguous block of the affinity mask per rank — eight ranks x one enqueue thread at ~80 launches
    per frame plus their PNG / staging helpers otherwise migrate over all cores and onto each other.  N = 1 (round 6): the
    affinity mask is left alone (other tenants of the host would pick the same block) and the THREAD COUNTS are confined
    to `cpu_share()` — OMP_NUM_THREADS / MKL_NUM_THREADS before torch is imported, `torch.set_num_threads` in main().
    FPC_BENCH_NO_AFFINITY=1 leaves everything alone.  Returns (cores kept or None, thread count to use)."""
    if os.environ.get("FPC_BENCH_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None, None
    share = cpu_share()
    try:
        if local_world > 1:
            cores = sorted(os.sched_getaffinity(0))
            per = min(len(cores), share) // local_world
            if per < 1:
                return None, None
            mine = cores[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, mine)
            nthr = max(1, min(per, 16))
        else:
            mine, nthr = None, share
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ.setdefault(var, str(nthr))
        return mine, nthr
    except OSError:
        return None, None


def multi_rank_fields(dist, torch, world, rank, dev, local_s, units_per_rank, backend, cores):
    """What makes an N > 1 line self-describing: every rank's own rate, how many ranks the backend's all-reduce really
    summed over, the backend, the cores each rank was pinned to.  All ranks call this (collectives inside)."""
    t = torch.tensor([float(local_s)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one)
    ncores = torch.tensor([float(len(cores) if cores else 0)], dtype=torch.float32, device=dev)
    dist.all_reduce(ncores, op=dist.ReduceOp.MIN)
    return {"scaling_measured": True,
            "per_rank_img_per_s": [round(units_per_rank / max(float(x.item()), 1e-9), 2) for x in every],
            "rccl_ranks_seen": int(round(float(one.item()))), "collective_backend": backend,
            "cores_per_rank": int(ncores.item()) or None,
            # HIP hardware queues this process asked for (fastposecnn_amd/__init__.py sets 8 before HIP initialises: the
            # frame streams + the pose gather's side stream + the null stream of ONE rank; N ranks ask for N x that of the node)
            "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) or None}


# --------------------------------------------------------------------------------------------------- pieces

def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(model_cpu, image, cat_cpu, hn, inv_k, encoder):
    """One frame on the host, three samples per leg (medians): torch CPU backbone (+ class compression) and the oracle's C
    restatement of aggregation / voting / RT, each leg on 1 thread (the scalar port) AND on the job's CPU share
    (`cpu_share()`: affinity mask and cgroup quota, whichever is smaller; the oracle's OpenMP loop over the hypotheses is
    capped at FPCO_MAX_THREADS), the better of the two taken per leg — a baseline is the host's best.  `cores` is the
    count the DOMINANT leg used.  Process-wide thread settings are left as they were found (round 5's version left
    OpenMP on one thread: the second call timed a one-thread network under a "cores: 16" label).  Bounded: ~15 s."""
    import torch
    from oracle import oracle as orc
    orc.build()
    saved_torch = torch.get_num_threads()
    saved_orc = orc.get_threads()
    share = cpu_share()
    cat_np = {k: v.numpy() for k, v in cat_cpu.items()}

    def net():
        with torch.no_grad():
            logits = model_cpu.pure_model_forward(image)
            model_cpu.class_compression(logits)

    def post():
        agg = orc.aggregate(cat_np)
        vertex = agg["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :]
        xy = orc.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=1)
        orc.pose_rt(agg["quaternion"], xy[:, 0], agg["z"], inv_k)

    def samples(fn, n=3):
        fn()                                      # untimed: thread pool start-up, page faults of the first pass
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    net_t, post_t = {}, {}
    try:
        for nthr in sorted({1, share}):
            torch.set_num_threads(nthr)
            net_t[nthr] = samples(net)
        torch.set_num_threads(saved_torch)
        for nthr in sorted({1, share}):
            used = orc.set_threads(nthr)
            post_t[used] = samples(post)
    finally:
        torch.set_num_threads(saved_torch)
        orc.set_threads(saved_orc)
    net_thr = min(net_t, key=lambda k: median(net_t[k]))
    post_thr = min(post_t, key=lambda k: median(post_t[k]))
    t_net, t_post = median(net_t[net_thr]), median(post_t[post_thr])
    dominant = net_thr if t_net >= t_post else post_thr

    def ms(d):
        return {str(k): [round(t * 1e3, 1) for t in v] for k, v in d.items()}
    return {"value": round(1.0 / (t_net + t_post), 4), "unit": "img/s", "cores": dominant, "kind": "port",
            "sample": f"1 frame x 3 samples per leg (medians after one untimed pass): torch-CPU {encoder}-FPN forward + class compression on "
                      f"{net_thr} thread(s) ({t_net * 1e3:.0f} ms) + oracle/fpc_oracle.c aggregation, hn={hn} voting and RT on {post_thr} "
                      f"thread(s) ({t_post * 1e3:.0f} ms); each leg timed on 1 thread and on the job's CPU share ({share} of the host's "
                      f"{os.cpu_count()} logical CPUs: affinity mask / cgroup quota), the faster kept; `cores` = the dominant leg's threads",
            "cpu_share": share, "threads": {"net": net_thr, "post": post_thr},
            "net_ms_by_threads": ms(net_t), "post_ms_by_threads": ms(post_t),
            "torch_threads_before_after": [saved_torch, torch.get_num_threads()]}


def vote_roofline(model_gpu, cat, n_inst, reps, label, calls=30, use_bits=True):
    """Average DEVICE time of the hough-voting launch sequence: `calls` enqueues of the call — each on its own freshly
    aggregated (cold) inputs — are captured into a HIP graph and the replay is bracketed by HIP events on its stream;
    median over `reps` replays, divided by `calls`.  This is what the rocprofv3 kernel trace of the same call adds up to
    (profiles/).  30 calls per group: a replay carries ~0.1 ms of fixed launch / completion cost (10 calls: 155 us per call
    on the 32-frame f32 sequence, 30 calls: 147-148, two-point slope: 143.5; tools_dev/vote_bench_method.py).  (Until round 4 the calls were enqueued eagerly back to back: the Python wrapper costs ~150 us of host time
    per call, as much as the 32-frame sequence itself, so that figure measured the wrapper — 155 us where the launch loop
    of tools_dev/vote_loop.py and the kernel trace say 146.  Eager enqueue stays as the fallback if capture fails.)"""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    mode = "HIP-graph replay of the captured calls (device time)"
    with torch.no_grad():
        aggs = [model_gpu.aggregate(cat) for _ in range(calls)]
        if not use_bits:                                       # the stand-alone interface: f32 mask planes, no side channel
            for agg in aggs:
                agg["instance_masks"]._fpc_mask_bits = None
        inputs = [(agg["instance_masks"], agg["xy"]) for agg in aggs]      # hough_voting replaces agg['xy']: keep the planes alive
        graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model_gpu.hough_voting(dict(aggs[0]))                        # workspace and allocator pools warm
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    outs = [model_gpu.hough_voting(dict(agg)) for agg in aggs]
            torch.cuda.current_stream().wait_stream(side)
        except Exception as e:      # noqa: BLE001 - any capture problem: measure eagerly
            print(f"vote_roofline: graph capture unavailable ({type(e).__name__}: {e}); eager timing", file=sys.stderr)
            graph = None
            mode = "eager back-to-back enqueues (host-bound above ~150 us per call)"
        cover = torch.empty(256 << 20, dtype=torch.float32, device=cat["mask"].device)
        for _ in range(reps):
            # the device is busy (~0.4 ms of fills) while the host gets to the graph launch: the region does not start with
            # the device waiting for it (a replay's launch latency was ~100 us of a 10-call group otherwise)
            cover.fill_(0.0)
            cover.fill_(1.0)
            ev[0].record()
            if graph is not None:
                graph.replay()
            else:
                for agg in aggs:
                    model_gpu.hough_voting(dict(agg))
            ev[1].record()
            ev[1].synchronize()
            ms.append(ev[0].elapsed_time(ev[1]) / calls)
        del aggs, inputs
    t = median(ms) * 1e-3
    alg = n_inst * 12 * H * W
    ach = alg / t / 1e9
    return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
            "traffic": None, "kernel": ("fpc_ransac_voting_v3_bits" if use_bits else "fpc_ransac_voting_v3") +
                                       " launch sequence (k_vote_scan, k_vote_plan, k_vote_count, k_vote_final)",
            "mask_source": ("bit words written by the aggregation layer (the model's pipeline): the scan does not read the f32 mask "
                            "planes, so `traffic` is below the algorithmic bytes; `achieved` still prices SURVEY 8(d)'s 12 H W per "
                            "instance, the bytes of the reference's own interface.  The stand-alone entry on f32 masks: "
                            "roofline_hn128_f32_masks, profiles/r04_vote_traffic_*.json, r04_vote_*_kernel_stats.csv") if use_bits else
                           "f32 mask planes through the reference's own interface (ransac_voting_layer_v3(mask, vertex, ...)): every "
                           "algorithmic byte is read",
            "workload": label, "algorithmic_bytes_per_launch": alg, "launch_ms": round(t * 1e3, 4),
            "timing": f"HIP events around {calls} calls on distinct cold inputs / {calls}, median of {reps}: {mode}"}


def post_network_rates(model_gpu, cat1, n1, cat32, n32, reps=15, calls=6):
    """Connected components + aggregation (the post-network stages in front of the vote) at B = 1 and B = 32, as a deferred
    enqueue (instance count kept on the device): HIP events around `calls` back-to-back enqueues, median of `reps`.
    Algorithmic bytes: CC reads the i64 mask (8 H W) and writes i32 labels (4 H W) per frame; aggregation reads labels + 9
    categorical planes (40 H W per frame) and writes instance masks + xy fields (12 H W per instance)."""
    import torch
    layer = model_gpu.aggregation_layer
    out = {}
    for tag, cat, n_inst in (("b1", cat1, n1), ("b32", cat32, n32)):
        B = cat["mask"].shape[0]
        import aggregation_layer as al
        cm = al.attach_fg_bits(cat["mask"].to(torch.int64).contiguous())      # as the class compression hands the mask over
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_cc, t_agg = [], []
        # device time: the enqueues of one stage are captured into a HIP graph and replayed (on one frame a stage is 15-30 us of
        # kernels behind ~30-60 us of Python per call; eager timing would report the host).  Eager fallback if capture fails.
        graphs = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.no_grad(), torch.cuda.stream(side):
                labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)      # warm the allocator pools
                layer._aggregate(cat, cm, labels, n_inst, n_dev)
                side.synchronize()
                g_cc, g_agg = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_cc, stream=side):
                    for _ in range(calls):
                        labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)
                with torch.cuda.graph(g_agg, stream=side):
                    for _ in range(calls):
                        keep = layer._aggregate(cat, cm, labels, n_inst, n_dev)
            torch.cuda.current_stream().wait_stream(side)
            graphs = (g_cc, g_agg)
        except Exception as e:      # noqa: BLE001 - any capture problem: measure eagerly
            print(f"post_network_rates: graph capture unavailable ({type(e).__name__}: {e}); eager timing", file=sys.stderr)
        with torch.no_grad():
            for _ in range(reps):
                ev[0].record()
                if graphs:
                    graphs[0].replay()
                else:
                    for _ in range(calls):
                        labels, n_dev = layer.batchwise_break_segmentation_mask(cm, return_device_count=True)
                ev[1].record()
                if graphs:
                    graphs[1].replay()
                else:
                    for _ in range(calls):
                        layer._aggregate(cat, cm, labels, n_inst, n_dev)
                ev[2].record()
                ev[2].synchronize()
                t_cc.append(ev[0].elapsed_time(ev[1]) / calls)
                t_agg.append(ev[1].elapsed_time(ev[2]) / calls)
        timing_mode = "HIP-graph replay of the captured enqueues (device time)" if graphs else "eager enqueues"
        cc_s, agg_s = median(t_cc) * 1e-3, median(t_agg) * 1e-3
        cc_b, agg_b = B * 12 * H * W, B * 40 * H * W + n_inst * 12 * H * W
        out[tag] = {"frames": B, "instances": n_inst,
                    "cc": {"us": round(cc_s * 1e6, 2), "bytes": cc_b, "GBps": round(cc_b / cc_s / 1e9, 1), "frac": round(cc_b / cc_s / 1e9 / HBM_PEAK_GBPS, 4)},
                    "aggregate": {"us": round(agg_s * 1e6, 2), "bytes": agg_b, "GBps": round(agg_b / agg_s / 1e9, 1),
                                  "frac": round(agg_b / agg_s / 1e9 / HBM_PEAK_GBPS, 4)},
                    "us": round((cc_s + agg_s) * 1e6, 2)}
    out["timing"] = f"HIP events around {calls} back-to-back calls / {calls}, median of {reps}, {timing_mode}; fractions of the 8 TB/s HBM peak"
    out["cc_mask_source"] = ("foreground bit words written by the class compression beside the i64 mask (the model's pipeline): "
                             "fpc_cc_label_bits reads 1/64 of the mask's bytes; `bytes` still prices the i64 mask of the reference's "
                             "interface (8 H W read + 4 H W written per frame).  The i64 entry fpc_cc_label: tools_dev/cc_time.py")
    return out


def encode_png_rgb_paeth(img):
    """8-bit RGB PNG with every row Paeth-filtered (type 4: the decoder's most expensive un-filter), written with zlib and
    numpy only — fixture set-up for `img_per_s_from_png_files`, outside every timed region."""
    import struct
    import zlib
    import numpy as np
    h, w, _ = img.shape
    x = img.astype(np.int16)
    a = np.zeros_like(x); a[:, 1:] = x[:, :-1]                 # left
    b = np.zeros_like(x); b[1:] = x[:-1]                       # up
    c = np.zeros_like(x); c[1:, 1:] = x[:-1, :-1]              # up-left
    pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    rows = ((x - pred) & 0xFF).astype(np.uint8).reshape(h, w * 3)
    raw = np.concatenate([np.full((h, 1), 4, np.uint8), rows], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def measure_copy_ceiling(dev):
    import torch
    try:
        src = torch.empty(256 << 20, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record(); c1.synchronize()
        return round(10 * 2 * src.numel() / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
    except Exception:
        return None


def check_gather(model_gpu, gatherer, cat, Bq, world, rank, cap, dev):
    """--check-gather: one more frame per rank through the post-network path and the gatherer; every rank then rebuilds
    EVERY rank's records on its own (the fixtures are functions of the frame index) and compares them with what the
    collective delivered: same records, rank order, sample ids offset by the shard's first image."""
    import torch
    import torch.distributed as dist
    from fastposecnn_amd import synth, parallel
    with torch.no_grad():
        agg = model_gpu.post_network_finish(model_gpu.post_network_enqueue(cat, seed=777))
    gatherer.add(agg, rank * Bq)
    if gatherer.pending:
        gatherer.flush()
    got = gatherer.latest().clone()
    slot = (gatherer.every if not gatherer.last_frames else gatherer.last_frames) - 1
    ok = True
    for r in range(world):
        cat_r_cpu, _ = synth.make_vote_batch(range(r * Bq, r * Bq + Bq))
        cat_r = {k: v.to(dev) for k, v in cat_r_cpu.items()}
        with torch.no_grad():
            agg_r = model_gpu.post_network_finish(model_gpu.post_network_enqueue(cat_r, seed=777))
        want = parallel.pack_pose_records(agg_r, r * Bq, cap)
        n = int(want[0, 0].view(torch.int32))
        ok = ok and n > 0 and torch.equal(got[r, slot, :n + 1], want[:n + 1])
        ids = got[r, slot, 1:n + 1, 0].contiguous().view(torch.int32)
        ok = ok and bool(((ids >= r * Bq) & (ids < (r + 1) * Bq)).all())
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def run_inference(args, encoder, Bq, hn, steps, warmup, world, rank, dev, want_backbone=True, split_precision=None, coalesce=1, tune_trials=1,
                  split_f16=None, split_f16_3p=None):
    """The timed hot path for one (encoder, batch) configuration.  Returns a dict of measurements and the objects
    later sections reuse."""
    import torch
    import torch.distributed as dist
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import config, synth, parallel
    from fastposecnn_amd.streaming import FrameStreamer

    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = hn
    hp.ENCODER = encoder
    hp.ENGINE_TUNE_MODE = args.tune_mode
    hp.ENGINE_SPLIT_PRECISION = (bool(int(os.environ.get('FPC_SPLIT_PRECISION', '1')))      # 0: plain f32 MFMA products only
                                 if split_precision is None else bool(split_precision))
    hp.ENGINE_SPLIT_F16 = (bool(int(os.environ.get('FPC_SPLIT_F16', '1')))      # 0: split-precision sites use the bf16 x 3 forms only
                           if split_f16 is None else bool(split_f16))
    hp.ENGINE_SPLIT_F16_3P = (bool(int(os.environ.get('FPC_SPLIT_F16_3P', '1')))      # 0: the fp16 forms keep all four piece products
                              if split_f16_3p is None else bool(split_f16_3p))
    hp.ENGINE_GRAPH = bool(int(os.environ.get('FPC_ENGINE_GRAPH', '1')))      # HIP graph replay of the frame-invariant launches
    torch.manual_seed(0)
    model = L.pose_regressor.MODELS[hp.MODEL].load_from_ckpt(None, hp).eval()
    model_gpu = model.to(dev)
    image = torch.stack([synth.make_image(rank * Bq + i) for i in range(Bq)])      # per-rank frames (weak scaling)
    cat_cpu, _ = synth.make_vote_batch(range(rank * Bq, rank * Bq + Bq))
    x = image.to(dev)
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    # the fixture stands in for the class compression's output: it carries the foreground bit words that stage writes beside
    # the i64 mask (engine.NetEngine.forward / gtf.class_compression_fused), which the connected-component labelling reads
    import aggregation_layer as al
    cat["mask"] = al.attach_fg_bits(cat["mask"].to(torch.int64).contiguous())
    n_inst = 6 * Bq                                           # vote-bench fixture: 6 instances per frame
    cap = 64 * Bq

    # the runtime's plans are autotuned for several frames in flight (latency x sqrt(share of the chip a launch occupies));
    # the model's own plan — what `backbone` times, one network alone — for latency
    stream_tune = None if (args.no_pipeline or args.stream_tune_mode < 0) else args.stream_tune_mode
    streamer = FrameStreamer(model_gpu, net_streams=1 if args.no_pipeline else args.net_streams,
                             post_inline=not args.post_stream, coalesce=coalesce, tune_mode=stream_tune)
    s_net = streamer.net_streams[0]
    depth = 0 if args.no_pipeline else (args.frames_in_flight - 1 if args.frames_in_flight > 0 else len(streamer.models))
    if coalesce > 1:                                          # a group per stream in flight, and one being filled
        depth = coalesce * (len(streamer.models) + 1) - 1
    pending = []
    # pose records of `gather_every` frames per RCCL all-gather, issued on a side stream: no frame waits for it
    gatherer = parallel.PoseGatherer(cap, every=args.gather_every or max(1, depth + 1), device=dev) if world > 1 else None

    def finish(ticket):
        out = {"aggregated": model_gpu.post_network_finish(ticket)} if args.vote_only else streamer.collect(ticket)
        if gatherer is not None:
            gatherer.add(out["aggregated"], rank * Bq)       # pack = one native launch; the collective follows on the side stream
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
        if coalesce > 1:
            streamer.flush()
        while pending:
            finish(pending.pop(0))
        if gatherer is not None and gatherer.pending:
            gatherer.flush()         # every rank runs the same number of steps here: the same number of collectives

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: every stream's plan is built and autotuned here (seconds), whatever --warmup says
    if not args.vote_only:
        # (tune_trials > 1: the set of plans is built that many times and the set that streams fastest is kept — untimed set-up)
        streamer.prepare(x, categorical_override=cat, tune_trials=1 if args.no_pipeline else tune_trials)
    for _ in range(2 * (depth + 1)):       # ... and every stream's allocator pool has seen a full pipeline of frames
        step()
    drain()
    for _ in range(warmup):
        step()
    drain()
    # EXACTLY `steps` steps between barrier + synchronize on both sides, max over ranks; the region is repeated until
    # `--min-seconds` are covered (20 steps are 16 ms, a quarter of it pipeline fill and drain) and the MEDIAN repeat is reported
    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        torch.cuda.synchronize()
        local.append(time.perf_counter() - t0)               # this rank's own time, before it waits for the others
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    local = []
    dts = [timed_region()]
    repeats = 1
    if args.min_seconds > 0:
        want = int(min(200, max(1, -(-args.min_seconds // max(dts[0], 1e-6)))))
        if world > 1:                                         # every rank must run the same number of regions
            t = torch.tensor([want], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            want = int(t.item())
        while repeats < want:
            dts.append(timed_region())
            repeats += 1
    dt = median(dts)

    # per-frame latency with ONE frame in flight (not the headline number)
    nlat = max(5, min(steps, 20))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(nlat):
        step(pipelined=False)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / nlat * 1e3

    pose_gather = None
    if gatherer is not None:
        # one collective, issue -> complete, serial (nothing else in flight): the cost the side stream hides per round
        ncol = gatherer.collectives
        barrier()
        t_c = time.perf_counter()
        for _ in range(20):
            gatherer.flush()
            gatherer.latest()
        us_col = (time.perf_counter() - t_c) / 20 * 1e6
        pose_gather = {"frames_per_collective": gatherer.every, "collectives": ncol, "stream": "side", "record_bytes": 160,
                       "capacity_per_frame": cap, "bytes_per_rank_per_collective": gatherer.every * (cap + 1) * 160,
                       "us_per_collective": round(us_col, 1),
                       "us_per_collective_note": "20 serial empty rounds after the timed region, issue -> host-visible completion; "
                                                 "warmed by the warm-up steps' collectives; inside the timed region it runs on a side stream"}
        if args.check_gather:
            pose_gather["verified"] = check_gather(model_gpu, gatherer, cat, Bq, world, rank, cap, dev)
    res = {"value": round(world * Bq * steps / dt, 3), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
           "repeats": repeats, "ms_per_step_min_max": [round(min(dts) / steps * 1e3, 4), round(max(dts) / steps * 1e3, 4)],
           "local_s": median(local), "pose_gather": pose_gather,
           "workload": f"{encoder}-FPN + all heads, batch={Bq} 640x480 per GPU per step, {1 + depth} frames in flight on "
                       f"{len(streamer.models)} streams, hn={hn}, {n_inst} instances per step (vote-bench fixture), random-init weights",
           "global_batch": world * Bq, "frames_in_flight": 1 + depth, "net_streams": len(streamer.models), "stream_tune_mode": stream_tune,
           "tune_trials": tune_trials, "trial_rates_img_per_s": getattr(streamer, "trial_rates", []),
           "ms_per_frame_one_in_flight": round(latency_ms, 4)}

    # the same pipeline fed from HOST memory: decoded u8 frames -> pinned staging -> H2D -> preprocessing kernels -> network
    # (F/tools/dataset.py:249-262 on the device).  PCIe-inclusive, so never `value`.
    if want_backbone and not args.vote_only and world == 1:
        import numpy as np
        from fastposecnn_amd.tools.dataset import FrameUploader
        AHEAD = 2       # uploads run this many frames ahead of the submissions: a frame's H2D copy + preprocessing (~60 us) are
        #                 finished when its stream gets to it (issued just before the submit, every frame started with its stream
        #                 waiting for its own upload: 1330-1358 img/s against 1440-1476 from resident tensors)
        up = FrameUploader(Bq, 480, 640, device=dev, slots=depth + 2 + AHEAD)
        frames = np.random.default_rng(0).integers(0, 256, (Bq, 480, 640, 3), dtype=np.uint8)
        uploaded = []

        def step_host(last=False):
            if not last:
                uploaded.append(up.upload(frames))
            if len(uploaded) > AHEAD or (last and uploaded):
                t, ready = uploaded.pop(0)
                pending.append(streamer.submit(t, categorical_override=cat, ready=ready))
                if len(pending) > depth:
                    finish(pending.pop(0))

        gaps = []

        def run_host(n):
            tp = time.perf_counter()
            for _ in range(n):
                step_host()
                tn = time.perf_counter()
                gaps.append(tn - tp)          # what one upload + submit (+ collect) cost the submitting thread
                tp = tn
            while uploaded:
                step_host(last=True)
            drain()

        # round 6: >= 600 frames per section, three sections, the MEDIAN reported with the submitting thread's longest stall —
        # round 5's section was 75 frames (50 ms) on a process that did not confine its thread pools to the job's CPU share
        nh = int(os.environ.get("FPC_BENCH_HOST_FRAMES", "0")) or max(-(-600 // Bq), steps // 4)
        run_host(depth + 3)
        sections = []
        for _ in range(3):
            del gaps[:]
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            run_host(nh)
            torch.cuda.synchronize()
            dt_h = time.perf_counter() - t2
            g = sorted(gaps)
            sections.append({"img_per_s": round(Bq * nh / dt_h, 2), "host_thread_max_gap_ms": round(g[-1] * 1e3, 3),
                             "host_thread_p99_gap_ms": round(g[int(0.99 * (len(g) - 1))] * 1e3, 3),
                             "host_thread_median_gap_ms": round(g[len(g) // 2] * 1e3, 3)})
        mid = sorted(sections, key=lambda d: d["img_per_s"])[1]
        res["host_frames_img_per_s"] = mid["img_per_s"]
        res["host_frames"] = {"frames_per_section": Bq * nh, "sections": sections, "host_thread_max_gap_ms": mid["host_thread_max_gap_ms"],
                              "note": "median of three sections; gap = time the submitting thread spent in one upload + submit (+ collect) step"}

        # ... and from ENCODED frames: `*_color.png` files (in memory, as a loader's read-ahead would hold them) -> native PNG
        # decode on a few host threads straight into the pinned staging slot -> the same path (F/tools/dataset.py:158 on
        # our side of the boundary).  Synthetic 640x480 RGB frames with smooth + noisy content (~600 KB each as PNG).
        yy, xx = np.mgrid[0:480, 0:640]
        pngs = []
        for i in range(4):
            img = np.stack([(128 + 100 * np.sin(xx / (23.0 + i)) * np.cos(yy / 31.0)), (xx * 255 / 639 + 20 * i) % 256, (yy * 255 / 479)], -1)
            img = (img + np.random.default_rng(i).integers(0, 24, (480, 640, 3))).clip(0, 255).astype(np.uint8)
            pngs.append(encode_png_rgb_paeth(img))
        from fastposecnn_amd.tools.dataset import PngFramePrefetcher
        workers = max(1, min(14, cpu_share() - 2))        # the job's CPU share (16 on a GPU box) minus the submitting thread and HIP's own
        npng = max(6, nh // 2) if Bq > 1 else max(40, nh)
        pre = PngFramePrefetcher(lambda k: [pngs[(k + j) % len(pngs)] for j in range(Bq)], npng + depth + 2, Bq, 480, 640, workers=workers)

        def step_png(frames):
            uploaded.append(up.upload(frames))
            if len(uploaded) > AHEAD:
                t, ready = uploaded.pop(0)
                pending.append(streamer.submit(t, categorical_override=cat, ready=ready))
                if len(pending) > depth:
                    finish(pending.pop(0))

        def flush_png():
            while uploaded:
                t, ready = uploaded.pop(0)
                pending.append(streamer.submit(t, categorical_override=cat, ready=ready))
                if len(pending) > depth:
                    finish(pending.pop(0))
            drain()

        it = iter(pre)
        for _ in range(depth + 2):
            step_png(next(it))
        flush_png()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for frames in it:
            step_png(frames)
        flush_png()
        torch.cuda.synchronize()
        res["png_files_img_per_s"] = round(Bq * npng / (time.perf_counter() - t3), 2)
        res["png_decode_workers"] = workers
        res["png_bytes_per_frame"] = int(sum(len(p) for p in pngs) / len(pngs))

    # backbone alone: HIP events on its stream around 8 back-to-back forwards (the host's enqueue time then hides behind the
    # previous forward's kernels: this is device time per network, as for the vote's roofline), median of 9 groups
    if want_backbone and not args.vote_only:
        net_ms = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(9):
            with torch.no_grad(), torch.cuda.stream(s_net):
                e0.record()
                for _ in range(8):
                    logits = model_gpu.pure_model_forward(x)
                    model_gpu.class_compression(logits)
                e1.record()
            e1.synchronize()
            net_ms.append(e0.elapsed_time(e1) / 8)
        t_net = median(net_ms) * 1e-3
        eng = next(iter(model_gpu._engines.values()), None)
        direct, executed, wino_share = eng.flops() if eng is not None else (0.0, 0.0, 0.0)
        tf_exec, tf_direct = executed / t_net / 1e12, direct / t_net / 1e12
        res["backbone"] = {"bound": "mfma", "achieved": round(tf_exec, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(tf_exec / MFMA_F32_PEAK_TFLOPS, 4), "ms": round(t_net * 1e3, 4),
                           "achieved_executed": round(tf_exec, 2), "achieved_direct_equiv": round(tf_direct, 2),
                           "executed_gflop_per_step": round(executed / 1e9, 2), "direct_gflop_per_step": round(direct / 1e9, 2),
                           "winograd_share_of_direct_flop": round(wino_share, 3),
                           "peak_note": "157.3 TFLOP/s = the f32 matrix-core peak; split-precision sites (config.matrix_products) issue six "
                                        "bf16 products per multiply-add: 2500 / 6 = 416.7 TFLOP/s f32-equivalent is their ceiling",
                           "frac_of_bf16x3_equiv_peak": round(tf_exec / (2500.0 / 6.0), 4),
                           "timing": "HIP events on the network's stream around 8 back-to-back forwards (+ class compression) / 8, median of 9",
                           "note": "`achieved` / `frac` price the multiply-adds the engine's current plans execute (a Winograd "
                                   "F(2x2,3x3) site does 1/2.25 of the direct convolution's); achieved_direct_equiv divides "
                                   "the direct-convolution FLOP by the same time"}
    return res, dict(model=model, model_gpu=model_gpu, image=image, cat_cpu=cat_cpu, cat=cat, n_inst=n_inst, hp=hp)


def attach_profiled_counters(roof, name):
    """`traffic` (HBM bytes per launch, FETCH_SIZE doubled per the guide's gfx950 note + WRITE_SIZE) and the VALU instruction
    rate come from PMC passes that rocprofv3 runs in its own processes (tools_dev/vote_traffic.py -> profiles/<name>): they
    are attached LABELLED, never as if measured by this run."""
    prof = load_profile_json(name) if name else None
    if not prof:
        return
    roof["traffic"] = prof.get("traffic_bytes_per_launch")
    roof["from_profile"] = {"file": "profiles/" + name, "commit": prof.get("commit"), "fields": ["traffic", "valu.wave_instructions_per_launch"]}
    vinst = prof.get("valu_wave_instructions_per_launch")
    if vinst:
        rate = vinst / (roof["launch_ms"] * 1e-3)
        roof["valu"] = {"bound": "valu", "achieved": round(rate / 1e9, 2), "peak": round(VALU_PEAK_WAVE_INSTR_PER_S / 1e9, 1),
                        "unit": "G wave-instr/s", "frac": round(rate / VALU_PEAK_WAVE_INSTR_PER_S, 4), "wave_instructions_per_launch": vinst,
                        "note": "SQ_INSTS_VALU of the four kernels (the profile) / the live launch time; peak = 1024 SIMD-32 x 2.4 GHz / 2 "
                                "cycles per wave64 instruction (tools_dev/mfma_vote_probe.hip measured 2.6 for v_sub + v_alignbit at >= 2 waves per SIMD)"}


def attach_mfma_busy(backbone, name):
    """`backbone.mfma_busy`: matrix-pipe busy share from SQ counters (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024
    SIMDs)) of a separately profiled forward of the same configuration (tools_dev/r5_conv_pmc.sh -> profiles/<name>): the
    whole forward and its dominant convolution kernels.  Attached LABELLED (`from_profile`), not measured by this run."""
    prof = load_profile_json(name) if name else None
    if not prof or backbone is None:
        return
    fam = prof.get("families", {})
    keep = {k: {"mfma_busy": v.get("mfma_busy"), "us_under_pmc": v.get("us_under_pmc"), "clock_GHz_under_pmc": v.get("clock_GHz_under_pmc")}
            for k, v in fam.items() if v.get("mfma_busy", 0) and ("k_conv" in k or "k_lateral" in k)}
    backbone["mfma_busy"] = {"forward": prof.get("forward", {}).get("mfma_busy"), "kernels": keep,
                             "unit": "share of cycles x SIMDs with the matrix pipe busy (counter-based; all kernels of the forward in the denominator)",
                             "from_profile": {"file": "profiles/" + name, "commit": prof.get("commit")}}


def load_profile_json(name):
    """A counter file under profiles/ that names the commit it was measured at; one without a commit is REFUSED (round 5 attached
    three files with `"commit": null`: nothing to check the kernels against)."""
    path = os.path.join(REPO, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            prof = json.load(f)
        if prof.get("commit"):
            return prof
        print(f"bench.py: profiles/{name} carries no commit: not attached", file=sys.stderr)
    return None


def promote_config3(line, c3, r3, args):
    """The default invocation's record: BASELINE.json configs[2] (ResNet34, batch 32 — the largest single-GPU configuration,
    and one rank's share of configs[3]) at the TOP level — value, ms_per_step, config, roofline, backbone, cpu_baseline —
    and configs[1] (ResNet18, one frame per step: pipelined `value` and `ms_per_frame_one_in_flight`) with its side
    measurements under `configs.config2`.  The 32-frame vote rooflines (`roofline_hn128*`), `post_network` and `train` are
    not tied to either backbone and stay where they were."""
    moved = ("value", "ms_per_step", "repeats", "ms_per_step_min_max", "config", "roofline", "backbone", "cpu_baseline",
             "plain_f32_products", "frames_per_launch", "steps", "warmup")      # (config 3's own `bf16x3_products` goes to the top level)
    c2 = {"metric": "img/s end-to-end 640x480 inference", "unit": "img/s", "dtype": "f32"}
    for k in moved:
        if k in line:
            c2[k] = line.pop(k)
    c2["value_note"] = ("`value` = frames per second with `config.frames_in_flight` frames in flight on `config.net_streams` streams; the "
                        "figure comparable to the reference's report_runtime fps (one frame, full sync: F/tools/timer.py:8-63) is "
                        "1000 / config.ms_per_frame_one_in_flight")
    cfg2 = c2.get("config", {})
    shared = {k: cfg2[k] for k in ("matrix_products", "post_network_input", "img_per_s_from_host_u8_frames_note") if k in cfg2}
    top = {
        "metric": line.pop("metric"), "value": r3["value"], "unit": "img/s", "n_gpus": line.pop("n_gpus"), "steps": r3["steps"],
        "warmup": r3["warmup"], "ms_per_step": r3["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "scaling_measured": line.pop("scaling_measured", False), "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": r3["repeats"], "ms_per_step_min_max": r3["ms_per_step_min_max"],
        "config": {"workload": r3["workload"], "baseline_config": "BASELINE.json configs[2] (per GPU also configs[3]'s share)",
                   "global_batch": r3["global_batch"], "parallelism": "single GPU", "vote_only": False,
                   "frames_in_flight": r3["frames_in_flight"], "net_streams": r3["net_streams"],
                   "ms_per_step_one_in_flight": r3["ms_per_frame_one_in_flight"],
                   "stream_tune_mode": r3.get("stream_tune_mode"), "tune_trials": r3.get("tune_trials"),
                   "trial_rates_img_per_s": r3.get("trial_rates_img_per_s"), "pose_gather": r3.get("pose_gather"),
                   "img_per_s_from_host_u8_frames": r3.get("host_frames_img_per_s"), "host_frames": r3.get("host_frames"),
                   "img_per_s_from_png_files": r3.get("png_files_img_per_s"), **shared},
        "roofline": c3["roofline"],
    }
    for k in ("backbone", "cpu_baseline", "bf16x3_products", "fp16_four_products"):
        if k in c3:
            top[k] = c3[k]
    for k in ("higher_is_better", "scaling", "vs_baseline", "dtype", "data", "unit"):
        line.pop(k, None)
    line.pop("configs", None)
    top.update(line)                                   # roofline_hn128*, post_network, train, ...
    top["configs"] = {"config2": c2}
    return top


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.train:
        from fastposecnn_amd import train_bench
        return train_bench.main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # this rank's share of the host cores, before torch starts its thread pools and before anything touches the GPU
    cores, host_threads = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    import torch
    if host_threads:
        torch.set_num_threads(host_threads)
    if os.environ.get("FPC_BENCH_DRYRUN"):
        # launch-path check without a GPU (tests/test_host_logic.py): rendezvous over gloo, one collective, rank 0 reports
        import torch.distributed as dist
        extra = {"scaling_measured": False}
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank + 1)])
            dist.all_reduce(t)
            dist.barrier()
            total = float(t.item())
            # the same self-description the real N > 1 line carries (here over gloo, with made-up per-rank times)
            extra = multi_rank_fields(dist, torch, world, rank, torch.device("cpu"), 0.5 * (rank + 1), 100.0, "gloo", cores)
            dist.destroy_process_group()
        else:
            total = 1.0
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "rank_sum": total, "local_rank": local_rank,
                              "master": os.environ.get("MASTER_ADDR"), **extra}), flush=True)
        return
    if args.promote and world > 1:
        args.encoder, args.batch = "resnet34", 32     # BASELINE.json configs[3]: 32 frames per GPU per step, the N = 1 top level's workload
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))    # (several ranks on one GPU only in tests)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FPC_BENCH_BACKEND", "nccl")        # "gloo": lets two ranks share one GPU in a smoke test
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from fastposecnn_amd import synth, _native
    _native.lib()      # fail loudly if the HIP library is missing

    res, ctx = run_inference(args, args.encoder, args.batch, args.hn, args.steps, args.warmup, world, rank, dev, tune_trials=args.tune_trials)
    multi = (multi_rank_fields(dist, torch, world, rank, dev, res["local_s"], args.batch * args.steps, backend, cores)
             if world > 1 else {"scaling_measured": False})

    line = None
    if rank == 0:
        roof = vote_roofline(ctx["model_gpu"], ctx["cat"], ctx["n_inst"], max(5, min(args.steps, 20)),
                             f"batch {args.batch}, hn {args.hn}, {ctx['n_inst']} instances")
        roof["measured_copy_GBps"] = measure_copy_ceiling(dev)
        attach_profiled_counters(roof, "r06_vote_bits_traffic_b1_hn1000.json" if (args.hn == 1000 and args.batch == 1) else None)
        roof["note"] = ("HIP events on the launch stream around the whole call, live in this run; `traffic` and `valu` are PMC "
                        "counters of a separate profiled run of the same call (`from_profile` names the file and the commit it was "
                        "taken at): rocprofv3 cannot collect them inside this process")
        if args.batch == 1:
            roof["bound_note"] = ("launch latency: four dependent launches (scan, plan, count, final: 6.5 + 10.3 + 14.6 + 10.2 us on one frame's "
                                  "six instances, profiles/r06_vote_bits_b1_hn1000_kernel_stats.csv) for ~3 us of HBM time; fewer launches were "
                                  "built in round 4 and were slower (tools_dev/r4_vote_fused/README.md); since round 6 the RT assembly rides on "
                                  "the last of the four (fpc_ransac_voting_v3_pose)")
        line = {
            "metric": "img/s end-to-end 640x480 inference; hough-vote kernel HBM GB/s vs roofline",
            "value": res["value"], "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", **multi, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": res["repeats"], "ms_per_step_min_max": res["ms_per_step_min_max"],
            "config": {"workload": res["workload"], "global_batch": res["global_batch"],
                       "parallelism": f"image-sharded dp{world}" if world > 1 else "single GPU",
                       "vote_only": bool(args.vote_only), "frames_in_flight": res["frames_in_flight"],
                       "net_streams": res["net_streams"], "ms_per_frame_one_in_flight": res["ms_per_frame_one_in_flight"],
                       "stream_tune_mode": res.get("stream_tune_mode"), "tune_trials": res.get("tune_trials"),
                       "trial_rates_img_per_s": res.get("trial_rates_img_per_s"),
                       "pose_gather": res["pose_gather"],
                       "matrix_products": ("f32 operands, f32 accumulation, f32 results (dtype f32).  Where the autotuner finds it "
                                           "faster a convolution's products run in SPLIT PRECISION on the 16-bit matrix instructions: (a) the "
                                           "exact three-way bf16 split of both operands, six partial products on v_mfma_f32_32x32x16_bf16, "
                                           "dropped terms < 2^-23 — any operand range; (b) round 6, 3x3 / stride-1 sites only "
                                           "(csrc/wino_h2.hip, wino_h3.hip): two fp16 pieces per operand (22 significant bits) on "
                                           "v_mfma_f32_32x32x16_f16 — all four piece products in two instructions per 8 channels, or (the "
                                           "default where Cin is a multiple of 16) three of them in three instructions per 16 channels: the "
                                           "dropped h2 g2 is <= 2^-22 of the term, the size of the two terms any two-piece form drops; "
                                           "weights scaled by a power of two on the device, activations as they "
                                           "come: 2^-22 relative for |v| >= 2^-3, 2^-25 absolute below, saturation beyond 1.3e5 — f32-level "
                                           "for activations of ordinary scale, held to the same bars (2e-5 per convolution, 1e-4 of the "
                                           "logits against float64 with EVERY eligible site forced onto it: tests/test_gpu_net.py).  "
                                           "FPC_SPLIT_F16_3P=0 (HPARAM.ENGINE_SPLIT_F16_3P = False) keeps all four products — "
                                           "`fp16_four_products`; FPC_SPLIT_F16=0 (HPARAM.ENGINE_SPLIT_F16 = False) keeps (a) only — `bf16x3_products`; "
                                           "FPC_SPLIT_PRECISION=0 keeps every product on v_mfma_f32_32x32x2_f32")
                                          if os.environ.get("FPC_SPLIT_PRECISION", "1") != "0" else
                                          "plain f32 matrix products (v_mfma_f32_32x32x2_f32) everywhere: FPC_SPLIT_PRECISION=0",
                       "post_network_input": "synthetic vote-bench fixture (SURVEY.md 8d), not the random-weight network's output",
                       "img_per_s_from_host_u8_frames": res.get("host_frames_img_per_s"), "host_frames": res.get("host_frames"),
                       "img_per_s_from_png_files": res.get("png_files_img_per_s"),
                       "img_per_s_from_png_files_note": "encoded *_color.png frames held in host memory -> native zlib-based decode on "
                                                        f"{res.get('png_decode_workers')} host threads running ahead (PngFramePrefetcher; ~10 ms of inflate + "
                                                        "un-filtering per frame and core) -> pinned staging -> H2D -> the same pipeline; "
                                                        f"{res.get('png_bytes_per_frame')} bytes per frame",
                       "img_per_s_from_host_u8_frames_note": "PCIe-inclusive: pinned u8 frames -> H2D -> preprocessing kernels "
                                                             "(tools/dataset.py:249-262 on the device) -> the same pipeline; "
                                                             "`value` starts from tensors resident in HBM"},
            "roofline": roof,
        }
        if "backbone" in res:
            line["backbone"] = res["backbone"]
            attach_mfma_busy(line["backbone"], "r06_conv_pmc_b1.json" if (args.encoder == "resnet18" and args.batch == 1) else
                             ("r06_conv_pmc_c3.json" if (args.encoder == "resnet34" and args.batch == 32) else None))

    if world == 1:
        # the training value of hn on a 32-frame batch (F/config.py:93): where the sequence is closest to its HBM bound
        if not args.no_hn128:
            hp128 = ctx["hp"]
            hp128.HV_NUM_OF_HYPOTHESES = 128
            cat32_cpu, _ = synth.make_vote_batch(range(32))
            cat32 = {k: v.to(dev) for k, v in cat32_cpu.items()}
            import aggregation_layer as al
            cat32["mask"] = al.attach_fg_bits(cat32["mask"].to(torch.int64).contiguous())
            line["roofline_hn128"] = vote_roofline(ctx["model_gpu"], cat32, 6 * 32, 9, "batch 32, hn 128, 192 instances")
            attach_profiled_counters(line["roofline_hn128"], "r06_vote_bits_traffic_b32_hn128.json")
            line["roofline_hn128_f32_masks"] = vote_roofline(ctx["model_gpu"], cat32, 6 * 32, 9, "batch 32, hn 128, 192 instances, f32 masks",
                                                             use_bits=False)
            attach_profiled_counters(line["roofline_hn128_f32_masks"], "r06_vote_traffic_b32_hn128.json")
            line["post_network"] = post_network_rates(ctx["model_gpu"], ctx["cat"], ctx["n_inst"], cat32, 6 * 32)
            hp128.HV_NUM_OF_HYPOTHESES = args.hn
            del cat32
        if not args.no_cpu_baseline:
            one = {k: v[:1] for k, v in ctx["cat_cpu"].items()}
            line["cpu_baseline"] = cpu_baseline(ctx["model"].to("cpu"), ctx["image"][:1], one, args.hn,
                                                torch.inverse(torch.from_numpy(ctx["hp"].NUMPY_INTRINSICS).float()).numpy(),
                                                args.encoder)
        if not args.no_plain_f32 and os.environ.get("FPC_SPLIT_PRECISION", "1") != "0" and not args.vote_only:
            # the same streamed measurement with every product on the f32 matrix instruction, so that one line holds both
            torch.cuda.empty_cache()
            rp, ctxp = run_inference(args, args.encoder, args.batch, args.hn, max(50, args.steps // 2), max(5, args.warmup // 2), 1, 0,
                                     dev, split_precision=False)
            line["plain_f32_products"] = {"value": rp["value"], "unit": "img/s", "ms_per_step": rp["ms_per_step"], "steps": rp["steps"],
                                          "backbone": {k: rp["backbone"][k] for k in ("ms", "achieved", "frac")} if "backbone" in rp else None,
                                          "note": "HPARAM.ENGINE_SPLIT_PRECISION = False: the engine's plans may only use "
                                                  "v_mfma_f32_32x32x2_f32 (a shorter timed region than `value`'s)"}
            del rp, ctxp
        if not args.no_batch_scan and args.batch == 1 and not args.vote_only:
            # the same one-frame-per-step stream with the runtime coalescing 2 / 4 consecutive frames per engine launch
            # (FrameStreamer(coalesce=k)): how far the batch-1 rate is from the kernels' own throughput.  NOT `value`.
            scan = {}
            for k in (2, 4):
                torch.cuda.empty_cache()
                rb, ctxb = run_inference(args, args.encoder, 1, args.hn, 4 * max(15, args.steps // 8), 8, 1, 0, dev,
                                         want_backbone=False, coalesce=k)           # steps and warm-up: whole groups
                scan[f"coalesce_{k}"] = {"value": rb["value"], "unit": "img/s", "ms_per_step": rb["ms_per_step"], "steps": rb["steps"],
                                         "frames_in_flight": rb["frames_in_flight"]}
                del rb, ctxb
            scan["note"] = ("the headline pipeline, ONE frame per step, with FrameStreamer(coalesce=k): the runtime groups k consecutive "
                            "frames into one engine launch of batch k and one batched post-network enqueue (dynamic batching; every "
                            "frame keeps its own ticket and result, its latency grows by the wait for its partners).  Informational, "
                            "never `value`: configs[1] is one frame per launch.  It shows that the batch-1 headline is bound by kernel "
                            "size (parallelism-bound encoder layers, ~60 launches per frame on four hardware queues), not by the "
                            "kernels' arithmetic")
            line["frames_per_launch"] = scan
        if not args.no_config3 and not (args.encoder == "resnet34" and args.batch == 32):
            del ctx
            torch.cuda.empty_cache()
            # promoted to the top level (the default invocation): EXACTLY the K timed steps and W warm-up steps asked for
            st, wu3 = (args.steps, args.warmup) if args.promote else (max(10, args.steps // 10), max(2, args.warmup // 5))
            r3, ctx3 = run_inference(args, "resnet34", 32, args.hn, st, wu3, 1, 0, dev)
            c3 = {"metric": "img/s end-to-end 640x480 inference", "value": r3["value"], "unit": "img/s",
                  "ms_per_step": r3["ms_per_step"], "steps": r3["steps"], "warmup": r3["warmup"], "dtype": "f32",
                  "config": {"workload": r3["workload"], "global_batch": 32, "frames_in_flight": r3["frames_in_flight"],
                             "img_per_s_from_host_u8_frames": r3.get("host_frames_img_per_s"),
                             "ms_per_step_one_in_flight": r3["ms_per_frame_one_in_flight"]}}
            if "backbone" in r3:
                c3["backbone"] = r3["backbone"]
            c3["roofline"] = vote_roofline(ctx3["model_gpu"], ctx3["cat"], ctx3["n_inst"], 5, f"batch 32, hn {args.hn}, 192 instances")
            attach_profiled_counters(c3["roofline"], "r06_vote_bits_traffic_b32_hn1000.json" if args.hn == 1000 else None)
            attach_mfma_busy(c3.get("backbone"), "r06_conv_pmc_c3.json")
            c3["roofline"]["bound_note"] = ("this configuration's count kernel is bound by vector-ALU issue, not HBM: 2 instructions per "
                                            "(entry, hypothesis) register pair behind 1/512 MFMA, ~3.0e9 pairs per call (`valu`); "
                                            "tools_dev/r4_vote_fused/README.md")
            if args.promote and not args.no_plain_f32 and os.environ.get("FPC_SPLIT_F16", "1") != "0" and os.environ.get("FPC_SPLIT_PRECISION", "1") != "0":
                # the same configuration with the fp16 x 2 Winograd form switched off: every split-precision site on bf16 x 3 pieces
                torch.cuda.empty_cache()
                rb3, ctxb3 = run_inference(args, "resnet34", 32, args.hn, max(6, st // 3), 2, 1, 0, dev, split_f16=False)
                c3["bf16x3_products"] = {"value": rb3["value"], "unit": "img/s", "ms_per_step": rb3["ms_per_step"], "steps": rb3["steps"],
                                         "backbone": {k: rb3["backbone"][k] for k in ("ms", "achieved", "frac")} if "backbone" in rb3 else None,
                                         "note": "HPARAM.ENGINE_SPLIT_F16 = False: split-precision sites may only use the three-way bf16 split "
                                                 "(no operand-range limit); a shorter timed region than `value`'s"}
                del rb3, ctxb3
                torch.cuda.empty_cache()
                if os.environ.get("FPC_SPLIT_F16_3P", "1") != "0":
                    rb4, ctxb4 = run_inference(args, "resnet34", 32, args.hn, max(6, st // 3), 2, 1, 0, dev, split_f16_3p=False)
                    c3["fp16_four_products"] = {"value": rb4["value"], "unit": "img/s", "ms_per_step": rb4["ms_per_step"], "steps": rb4["steps"],
                                                "backbone": {k: rb4["backbone"][k] for k in ("ms", "achieved", "frac")} if "backbone" in rb4 else None,
                                                "note": "HPARAM.ENGINE_SPLIT_F16_3P = False: the fp16-pieces sites keep all four piece products "
                                                        "(csrc/wino_h2.hip); a shorter timed region than `value`'s"}
                    del rb4, ctxb4
                    torch.cuda.empty_cache()
            if args.promote and not args.no_cpu_baseline:
                one3 = {k: v[:1] for k, v in ctx3["cat_cpu"].items()}
                c3["cpu_baseline"] = cpu_baseline(ctx3["model"].to("cpu"), ctx3["image"][:1], one3, args.hn,
                                                  torch.inverse(torch.from_numpy(ctx3["hp"].NUMPY_INTRINSICS).float()).numpy(), "resnet34")
            c3["_res"] = r3
            line["configs"] = {"config3": c3}
    if world == 1 and not args.no_train_line and not args.vote_only:
        # BASELINE.json configs[4] at its per-GPU share (B = 8) on this GPU, so that the driver's run times it as well
        # — in a CHILD process (round 6): inside this one, behind the inference sections' plans, streams and graphs, the same step took
        # 39 ms where `python bench.py --train` takes 33-34 (the earlier sections' allocator and runtime state, not the kernels: with
        # only config 2 in front of it 33.2).  The parent releases its cached device memory and idles meanwhile; nothing is exec'ed.
        try:
            ctx = ctx3 = None                                  # plans, streamers and workspaces of the inference sections
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--train", "--steps", "12", "--warmup", "3",
                                  "--train-batch", str(args.train_batch), "--bucket-mb", str(args.bucket_mb)],
                                 capture_output=True, text=True, timeout=600)
            rows = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if out.returncode != 0 or not rows:
                raise RuntimeError(f"child exited with {out.returncode}: {out.stderr[-400:]}")
            line["train"] = json.loads(rows[-1])
            line["train"]["measured_in"] = "a child process of this run (python bench.py --train --steps 12 --warmup 3), the parent idle"
        except Exception as e:                                 # the inference line must not be lost to the extra section
            line["train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        c3 = (line.get("configs") or {}).get("config3")
        if c3 is not None:
            r3 = c3.pop("_res")
            if args.promote:
                line = promote_config3(line, c3, r3, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
