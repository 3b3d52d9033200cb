"""Frame-streaming runtime: keeps several frames of `PoseRegressor.forward` in flight on separate HIP
streams so that the GPU always has wide AND latency-bound kernels to run side by side.

At batch 1 a frame is a chain of ~80 dependent launches whose deep-encoder / post-network kernels
occupy a fraction of the 256 CUs.  `FrameStreamer` deals consecutive frames round-robin to `net_streams`
native plans (each with its own workspace; the parameters are shared), each on its own HIP stream:

    frame i   : network -> aggregation / voting / RT   on stream S[i % k]
    frame i+1 : the same                               on stream S[(i+1) % k]   (overlaps frame i)

k = 4 is the measured optimum on MI355X (one stream per hardware compute pipe: 1210 img/s against 1040 with
two network streams + a shared post-network stream, 1020 with five).  The HIP runtime multiplexes streams onto
GPU_MAX_HW_QUEUES hardware queues (default 4, and the null stream takes part): with fewer queues than streams
two frames share a queue and serialise (measured: 990 img/s).  `fastposecnn_amd/__init__.py` therefore sets
GPU_MAX_HW_QUEUES=8 unless the variable is already set — import the package before the first HIP call.
`post_inline=False` runs the post-network stages of all frames on one extra stream instead.

`coalesce=k` (round 4, off by default): consecutive single frames are grouped k at a time into ONE engine launch of batch k
(and one batched post-network enqueue) — dynamic batching, as a serving runtime would do it: at batch 1 the encoder's deep
layers and the ~60 launches per frame leave the chip under-filled even with four streams (measured: 1465 img/s at one frame
per launch, 1663 at two, 1753 at four, same streams).  Every frame still gets its own ticket and its own forward() dict
(instances of the group are split by sample id; `sample_ids` are frame-local again); a frame completes when its group does,
so per-frame latency grows by the wait for its partners.  `flush()` enqueues an incomplete group (end of a stream).

`submit(x)` enqueues one frame and returns a ticket without synchronising; `collect(ticket)` waits on
that frame's own event (the only host wait of the frame), trims the per-instance tensors to the
instance count found and returns the reference's forward() dict
{'logits', 'categorical', 'aggregated'}.  Every frame does all of its work; only its completion is
deferred.  Nothing like this exists in the reference (single stream, >= 3 host syncs per instance).
"""
import copy

import torch


class FrameStreamer:

    def __init__(self, model, net_streams=4, device=None, post_inline=True, coalesce=1, tune_mode=None):
        p = next(model.parameters())
        self.device = device if device is not None else p.device
        if self.device.type != "cuda":
            raise RuntimeError("FrameStreamer needs the model on a GPU")
        if model.training:
            raise RuntimeError("FrameStreamer is an inference runtime: call model.eval() first")
        # tune_mode (fpc_net_autotune_next: 1 = latency x sqrt(share of the chip) — the objective for several frames in flight):
        # every plan of the runtime, the first included, is then a copy with its own HPARAM.ENGINE_TUNE_MODE, and the caller's
        # model keeps whatever plan it tunes for itself (the tilings differ, so results equal the caller's own forward only to
        # f32 rounding, like any two tuned plans).  None: the first plan is the caller's model with its own setting.
        self.models = [model] if tune_mode is None else []
        while len(self.models) < max(1, int(net_streams)):
            m = copy.copy(model)            # shares parameters / sub-modules, owns its native plans
            m._engines = {}
            m._fused = None
            m._pinned_counts, m._pinned_next = [], -1     # its own read-back slots (post_network_enqueue)
            if tune_mode is not None:
                m.HPARAM = copy.copy(model.HPARAM)
                m.HPARAM.ENGINE_TUNE_MODE = int(tune_mode)
            self.models.append(m)
        self.net_streams = [torch.cuda.Stream(device=self.device) for _ in self.models]
        self.post_stream = None if post_inline else torch.cuda.Stream(device=self.device)
        self._n = 0
        self._warm = set()          # (plan, input shape) pairs whose native plan exists
        self.coalesce = max(1, int(coalesce))
        self._group = None          # frames waiting for their partners: {"frames": [...], "tickets": [...]}
        self._cat_cache = {}
        self._sid_pool, self._sid_next = [], -1      # pinned read-back slots of the groups' sample ids
        self._zeros = torch.zeros(1024, dtype=torch.int64, device=self.device)

    def _enqueue(self, k, x, x_ready, categorical_override, seed):
        model, stream = self.models[k], self.net_streams[k]
        stream.wait_event(x_ready)                            # x was produced on the caller's stream
        # x is read on `stream` after the caller may have dropped it: tell the caching allocator, or the block can be
        # handed out again (and overwritten) on the caller's stream while this frame's first kernel still reads it
        for t in (x if isinstance(x, (list, tuple)) else (x,)):
            t.record_stream(stream)
        grouped = isinstance(x, (list, tuple))
        with torch.no_grad():
            with torch.cuda.stream(stream):
                if isinstance(x, (list, tuple)):              # a coalesced group: one batch, built on the frame's own stream
                    x = torch.cat(list(x), dim=0)             # (on the caller's null stream the copy would cost the host what
                    #                                            every null-stream operation beside busy streams costs)
                model._inv_k(x.device)
                logits = model.pure_model_forward(x)
                cat = model.class_compression(logits)
                ev = torch.cuda.Event()
                ev.record()
            ticket = {"logits": logits, "categorical": cat, "post": None, "model": model, "net_event": ev}
            if model.HPARAM.PERFORM_AGGREGATION:
                src = categorical_override if categorical_override is not None else cat
                ps = self.post_stream if self.post_stream is not None else stream
                ticket["post_stream"] = ps
                with torch.cuda.stream(ps):
                    if ps is not stream:
                        ps.wait_event(ev)
                    for t in src.values():
                        t.record_stream(ps)
                    if model.HPARAM.PERFORM_HOUGH_VOTING:
                        ticket["post"] = model.post_network_enqueue(src, seed=seed)
                        if grouped:
                            # which frame of the group an instance belongs to: read back with the frame's own work (pinned,
                            # asynchronous, its own event), so that splitting the group's instances later needs no device
                            # operation behind whatever has been queued on this stream since
                            sid = ticket["post"]["agg"]["sample_ids"]
                            if len(self._sid_pool) < 32:
                                self._sid_pool.append(torch.empty(max(64, sid.numel()), dtype=torch.int64).pin_memory())
                            self._sid_next = (self._sid_next + 1) % len(self._sid_pool)
                            host = self._sid_pool[self._sid_next]
                            if host.numel() < sid.numel():
                                host = self._sid_pool[self._sid_next] = torch.empty(sid.numel(), dtype=torch.int64).pin_memory()
                            host[:sid.numel()].copy_(sid, non_blocking=True)
                            ev2 = torch.cuda.Event()
                            ev2.record()
                            ticket["sid_host"] = (host, ev2)
                    else:
                        ticket["agg_only"] = model.aggregate(src)
        return ticket

    def prepare(self, x, categorical_override=None, tune_trials=1, trial_frames=64):
        """Build (and autotune) every stream's native plan for this input shape now, one at a time on an otherwise
        idle GPU, by running one frame through each.  Optional: `submit` does the same lazily on a plan's first frame.

        tune_trials = N > 1 (one frame per launch only): the per-convolution autotuning times candidates alone on the chip,
        and the tilings it picks vary from run to run by 2-3 % of the STREAMED rate; so the whole set of plans is built N
        times, each set is run as it will be used — `trial_frames` frames through all streams with the usual number in flight —
        and the set that streams fastest is kept (`self.trial_rates`: frames per second of every trial)."""
        import time
        trials = max(1, int(tune_trials)) if self.coalesce == 1 else 1
        best, self.trial_rates = None, []
        for trial in range(trials):
            if trial > 0:                                      # new plans: drop the ones just measured (the best set stays referenced)
                for m in self.models:
                    m._engines, m._fused = {}, None
                self._warm.clear()
            for _ in range(len(self.models)):
                ts = [self.submit(x, categorical_override=categorical_override) for _ in range(self.coalesce if x.shape[0] == 1 else 1)]
                for t in ts:
                    self.collect(t)
            if trials == 1:
                return
            rates = []
            for _ in range(3):
                pend = []
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(int(trial_frames)):
                    pend.append(self.submit(x, categorical_override=categorical_override))
                    if len(pend) > len(self.models):
                        self.collect(pend.pop(0))
                while pend:
                    self.collect(pend.pop(0))
                torch.cuda.synchronize(self.device)
                rates.append(int(trial_frames) * x.shape[0] / (time.perf_counter() - t0))
            rate = sorted(rates)[1]
            self.trial_rates.append(round(rate, 1))
            if best is None or rate > best[0]:
                best = (rate, [(m._engines, m._fused) for m in self.models])
        for m, (eng, fused) in zip(self.models, best[1]):
            m._engines, m._fused = eng, fused

    def submit(self, x, categorical_override=None, ready=None):
        """Enqueue one frame (see `_submit_now`).  With coalesce = k > 1 and a single-frame x: the frame joins the current group;
        the k-th one launches the group as one batch.  Returns this frame's ticket either way."""
        if self.coalesce == 1 or x.shape[0] != 1:
            return self._submit_now(x, categorical_override, ready)
        if self._group is None:
            self._group = {"frames": [], "tickets": [], "launched": None, "result": None}
        g = self._group
        t = {"group": g, "index": len(g["frames"])}
        g["frames"].append((x, categorical_override, ready))
        g["tickets"].append(t)
        if len(g["frames"]) == self.coalesce:
            self._launch_group()
        return t

    def _launch_group(self):
        g, self._group = self._group, None
        readies = [f[2] for f in g["frames"] if f[2] is not None]
        xs = [f[0] for f in g["frames"]]
        x = xs[0] if len(xs) == 1 else xs
        cats = [f[1] for f in g["frames"]]
        cat = None
        if all(c is not None for c in cats):
            # (benchmark facility) the fixtures of a group as one batch; the same fixture objects again: the same batch again
            key = tuple(id(c) for c in cats)
            hit = self._cat_cache.get(key)
            if hit is not None:
                cat = hit[1]
            elif len(cats) == 1:
                cat = cats[0]
            else:
                import aggregation_layer as al
                cat = {k: torch.cat([c[k] for c in cats], dim=0) for k in cats[0]}
                bits = [al.fg_bits_of(c["mask"]) for c in cats]
                if all(b is not None for b in bits):     # the foreground bit words travel with the concatenated mask
                    cat["mask"]._fpc_fg_bits = (torch.cat(bits, dim=0), cat["mask"]._version)
                if len(self._cat_cache) < 8:
                    self._cat_cache[key] = (cats, cat)   # (holds the fixtures: their ids stay theirs)
        g["launched"] = self._submit_now(x, cat, None, extra_ready=readies)
        g["frames"] = None

    def flush(self):
        """Launch the frames that are still waiting for partners (the end of a stream): a smaller batch, whose plan is built
        on first use."""
        if self._group is not None and self._group["frames"]:
            self._launch_group()

    def _collect_group(self, g):
        if g["result"] is None:
            if g["launched"] is None:                    # collected before its group filled up
                assert self._group is g
                self._launch_group()
            out = self._collect_now(g["launched"])
            n = len(g["tickets"])
            agg = out["aggregated"]
            bounds = None
            if agg is not None:
                # instances per frame of the group, from the sample ids read back with the group's own work
                cnt = int(agg["class_ids"].shape[0])
                tag = g["launched"].get("sid_host")
                if tag is not None and cnt <= g["launched"]["post"]["cap"]:      # (a capacity overflow re-ran the stages: ids from the device)
                    tag[1].synchronize()
                    ids = tag[0][:cnt].tolist()
                else:
                    ids = agg["sample_ids"].tolist()
                counts = [0] * n
                for v in ids:
                    counts[int(v)] += 1
                bounds = [0]
                for c in counts:
                    bounds.append(bounds[-1] + c)
            g["result"] = (out, bounds)
            g["launched"] = None
        return g["result"]

    def _submit_now(self, x, categorical_override=None, ready=None, extra_ready=()):
        """Enqueue one frame (x f32 [B,3,H,W] on the device).  `categorical_override` replaces the
        network's categorical output as the input of the post-network stages (benchmark fixture).
        `ready`: the event after which x is valid (e.g. FrameUploader.upload's); default: everything enqueued so far on
        the caller's current stream."""
        k = self._n % len(self.models)
        self._n += 1
        # the frame stream always waits for everything queued so far on the caller's stream: that is what collect()'s
        # ownership rule rests on (readers of earlier outputs finish before a later frame can reuse their blocks)
        caller_done = torch.cuda.Event()
        caller_done.record(torch.cuda.current_stream(self.device))
        x_ready = ready
        if x_ready is None:
            x_ready = caller_done
        else:
            self.net_streams[k].wait_event(caller_done)
        for ev in extra_ready:                                # the upload events of a coalesced group's frames
            self.net_streams[k].wait_event(ev)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())    # the vote's sampler seed: drawn here, in submission order
        shape = (len(x),) + tuple(x[0].shape[1:]) if isinstance(x, (list, tuple)) else tuple(x.shape)
        key = (k, shape)
        if key not in self._warm:
            # first frame of this plan / shape: the plan is built and autotuned (on-device timing of every
            # candidate tiling) — alone on the GPU, not under the other streams' frames
            torch.cuda.synchronize(self.device)
            t = self._enqueue(k, x, x_ready, categorical_override, seed)
            torch.cuda.synchronize(self.device)
            self._warm.add(key)
            return t
        return self._enqueue(k, x, x_ready, categorical_override, seed)

    def collect(self, ticket):
        """Wait for the ticket's frame (only; with coalescing: its group) and return forward()'s dict for THAT frame."""
        if "group" not in ticket:
            return self._collect_now(ticket)
        out, bounds = self._collect_group(ticket["group"])
        j = ticket["index"]
        res = {"logits": {k: v[j:j + 1] for k, v in out["logits"].items()} if out["logits"] is not None else None,
               "categorical": {k: v[j:j + 1] for k, v in out["categorical"].items()}, "aggregated": None}
        if out["aggregated"] is not None:
            lo, hi = bounds[j], bounds[j + 1]
            agg = {k: v[lo:hi] for k, v in out["aggregated"].items()}
            # frame-local again: every instance of this slice belongs to sample 0 of its own frame (a view of a standing zeros
            # tensor: a subtraction here would be a kernel on the caller's stream per collected frame)
            if hi - lo > self._zeros.numel():
                self._zeros = torch.zeros(2 * (hi - lo), dtype=torch.int64, device=self.device)
            agg["sample_ids"] = self._zeros[:hi - lo]
            res["aggregated"] = agg
        return res

    def _collect_now(self, ticket):
        """Wait for the ticket's frame (only) and return forward()'s dict.

        Ownership rule: the returned tensors live in the frame stream's allocator pool and the frame is complete when
        this returns.  Consume them — synchronously or by kernels enqueued asynchronously — on the stream you call
        `submit` from: every submit makes its frame stream wait for all work enqueued on that caller stream so far, so
        a block you have dropped is never handed to a later frame before your queued readers finished.  (Registering
        every output with `record_stream` instead costs 15 % of the streamed rate: the allocator then defers each
        block's reuse behind an event query.)  A consumer on any OTHER stream must call `t.record_stream(that_stream)`
        itself before dropping the tensor."""
        model = ticket["model"]
        agg = None
        if ticket["post"] is not None:
            with torch.cuda.stream(ticket["post_stream"]):
                agg = model.post_network_finish(ticket["post"])
        elif "agg_only" in ticket:
            ticket["post_stream"].synchronize()
            agg = ticket["agg_only"]
        else:
            ticket["net_event"].synchronize()
        return {"logits": ticket["logits"], "categorical": ticket["categorical"], "aggregated": agg}
