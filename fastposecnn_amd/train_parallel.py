"""Data-parallel training step across the GPUs of one node (BASELINE config 5; SURVEY.md 8e "Training").

The reference delegates this to Lightning's DDP (F/config.py:60, F/train.py:316-327): NCCL all-reduce of every gradient
bucket, then every rank runs the whole Lookahead(RAdam) step on all 18.7 M parameters.  Here the same mathematics is laid
out for xGMI, which is point-to-point and per-link bound:

    backward  ->  per bucket, as soon as its last gradient has been accumulated:  reduce-scatter (RCCL, side stream)
    step      ->  every rank owns 1/world of each bucket: gradient norm of its shards (one scalar all-reduce), then ONE
                  native Lookahead(RAdam) launch per shard with clip coefficient, 1/world and the inf/NaN guard folded in
                  as device scalars (csrc/train.hip) — optimiser state and work are 1/world per rank
              ->  all-gather of the updated parameters, in place in the flat parameter buffer

reduce-scatter + all-gather move the same bytes as an all-reduce; the optimiser runs once per element instead of `world`
times, and its state (m, v, slow weights: 3 x 74.7 MB for ResNet18-FPN) is sharded.  Parameters and gradients live in two
flat f32 buffers (bucket-contiguous, each bucket padded to world x 4 elements): `p.data` / `p.grad` are views, so the
collectives and the optimiser see contiguous memory and no per-step flatten / unflatten copy exists.
No host synchronisation inside a step: the clip coefficient and the skip flag stay on the device.
"""
import math

import torch
import torch.distributed as dist


def _slot(numel):
    """Elements a parameter occupies in the flat buffers: every parameter starts on a 16-byte boundary (the network
    plan's packing kernels and fpc_net_load_params read parameters with 16-byte loads and refuse unaligned pointers);
    the padding stays zero in the parameters, gradients and moments, so the step leaves it zero."""
    return (numel + 3) // 4 * 4


class ShardedLookaheadRAdam:

    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=3e-4, la_k=5, la_alpha=0.5,
                 clip_norm=0.15, bucket_mb=16.0, group=None, step_fn=None, always_collective=False):
        """lr / weight_decay: F/config.py:56-57; clip_norm: F/train.py `gradient_clip_val`; la_k / la_alpha / betas / eps:
        catalyst's defaults (F/lib/pose_regressor.py:420-423).  `step_fn(p, g, m, v, slow, step, ctl)`: replaces the native
        kernel (CPU tests only; without it CPU parameters are refused — there is no CPU fallback).  `always_collective`:
        run the reduce-scatter / all-reduce / all-gather also in a process group of ONE rank (the single-GPU test of the
        RCCL device path, tests/test_gpu_rccl_one_rank.py); a lone rank normally skips them."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self._collective = self.world > 1 or (bool(always_collective) and dist.is_initialized())
        self.hp = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay, la_k=la_k, la_alpha=la_alpha)
        self.clip_norm = clip_norm
        self.step_count = 0
        self.step_fn = step_fn
        self._model = model
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        self.device = params[0].device
        if self.device.type != "cuda" and step_fn is None:
            raise RuntimeError("fastposecnn_amd: the optimiser step is a HIP kernel; parameters must be on a GPU")
        # buckets in reverse registration order (~ the order backward produces gradients)
        quantum = 4 * self.world
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []                  # dicts: params, offset (in the flat buffers), numel (padded)
        cur, cur_n = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_n += _slot(p.numel())
            if cur_n >= limit:
                self.buckets.append({"params": cur, "raw": cur_n})
                cur, cur_n = [], 0
        if cur:
            self.buckets.append({"params": cur, "raw": cur_n})
        off = 0
        for b in self.buckets:
            b["offset"] = off
            b["numel"] = (b["raw"] + quantum - 1) // quantum * quantum
            off += b["numel"]
        self.total = off
        f32 = dict(dtype=torch.float32, device=self.device)
        self.flat_p = torch.zeros(self.total, **f32)
        self.flat_g = torch.zeros(self.total, **f32)
        shard_total = self.total // self.world
        self.m = torch.zeros(shard_total, **f32)
        self.v = torch.zeros(shard_total, **f32)
        self.slow = torch.zeros(shard_total, **f32)
        self.g_shard = torch.zeros(shard_total, **f32) if self._collective else None
        soff = 0
        self._hooks = []
        self._sinks = []                   # (weight data pointer, GradSink) of the convolution weights: written by the weight-gradient kernel
        self._counted = set()              # parameters whose gradient has arrived this step (hook or sink, whichever came first)
        try:
            from fastposecnn_amd.lib import train_conv as _tc      # the module object the model's convolutions use (lib/backbone.py)
        except ImportError:
            _tc = None
        for bi, b in enumerate(self.buckets):
            o = b["offset"]
            for p in b["params"]:
                n = p.numel()
                if p.dtype != torch.float32:
                    raise TypeError("f32 parameters only")
                view = self.flat_p[o:o + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + n].view_as(p)
                hook = self._make_hook(bi, id(p))
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
                if _tc is not None and p.dim() == 4 and self.device.type == "cuda":
                    sink = _tc.GradSink(p.grad, (lambda h=hook, q=p: h(q)), self)
                    _tc.grad_sinks[p.data_ptr()] = sink
                    self._sinks.append((p.data_ptr(), sink))
                o += _slot(n)
            b["shard"] = b["numel"] // self.world
            b["shard_offset"] = soff
            soff += b["shard"]
            b["pending"] = len(b["params"])
            b["launched"] = False
        self._next = 0                     # first bucket whose reduction has not been launched this step
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.stat = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.ctl = torch.zeros(2, **f32)
        self.skipped = torch.zeros(1, dtype=torch.int64, device=self.device)     # steps whose gradients the inf / NaN guard zeroed

    # ---- backward-time reduction -------------------------------------------------------------------------------
    def _make_hook(self, bi, pid):
        def hook(_p):
            if pid in self._counted:       # (a weight whose first gradient went through its sink and a second one through autograd)
                return
            self._counted.add(pid)
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        # Collectives pair up across ranks by CALL ORDER, so every rank must issue the buckets in the same order whatever
        # order its backward completes them in (a rank with no matched instance produces no gradient for the rotation /
        # translation / scales branches at all: those buckets complete only in step()).  Bucket i is launched once buckets
        # 0..i-1 have been launched, as DDP does; step() flushes the remainder in index order.
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._reduce_bucket(self.buckets[self._next])
            self._next += 1

    def _bucket_grad(self, b):
        return self.flat_g[b["offset"]:b["offset"] + b["numel"]]

    def _bucket_param(self, b):
        return self.flat_p[b["offset"]:b["offset"] + b["numel"]]

    def _reduce_bucket(self, b):
        b["launched"] = True
        if not self._collective:
            return
        g = self._bucket_grad(b)
        out = self.g_shard[b["shard_offset"]:b["shard_offset"] + b["shard"]]
        if self.backend == "gloo":          # gloo has no reduce-scatter: all-reduce and keep the own shard (tests only)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(g[self.rank * b["shard"]:(self.rank + 1) * b["shard"]])
        elif self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group)

    def zero_grad(self):
        self.flat_g.zero_()
        self._next = 0
        self._counted.clear()
        for _, sink in self._sinks:
            sink.written = False
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["launched"] = False
            for p in b["params"]:           # a caller's zero_grad(set_to_none=True) must not detach the views
                if p.grad is None or p.grad.data_ptr() < self.flat_g.data_ptr() or \
                        p.grad.data_ptr() >= self.flat_g.data_ptr() + 4 * self.total:
                    raise RuntimeError("a parameter's .grad no longer aliases the flat gradient buffer: use "
                                       "ShardedLookaheadRAdam.zero_grad(), not zero_grad(set_to_none=True)")

    # ---- the step -----------------------------------------------------------------------------------------------
    def _shard_views(self, b):
        so, sh = b["shard_offset"], b["shard"]
        p = self._bucket_param(b)[self.rank * sh:(self.rank + 1) * sh]
        g = self.g_shard[so:so + sh] if self._collective else self._bucket_grad(b)
        return p, g, self.m[so:so + sh], self.v[so:so + sh], self.slow[so:so + sh]

    def step(self):
        while self._next < len(self.buckets):   # buckets whose gradients never all arrived (no matched instance): zeros
            self._reduce_bucket(self.buckets[self._next])
            self._next += 1
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self.step_count += 1
        # gradient norm of the SUMMED gradients -> clip coefficient of the MEAN gradients, all on the device
        self.stat.zero_()
        native = self.device.type == "cuda" and self.step_fn is None
        if native:
            from fastposecnn_amd import _native as nat
            L = nat.lib()
        for b in self.buckets:
            _, g, _, _, _ = self._shard_views(b)
            if native:
                with torch.cuda.device(self.device):
                    nat.check(L.fpc_grad_sumsq(nat.ptr(g), g.numel(), nat.ptr(self.stat), nat.stream()), "fpc_grad_sumsq")
            else:
                self.stat[0] += (g.double() ** 2).sum()
                self.stat[1] += (~torch.isfinite(g)).sum()
        if self._collective:
            dist.all_reduce(self.stat, op=dist.ReduceOp.SUM, group=self.group)
        norm = torch.sqrt(self.stat[0]) / self.world
        bad = (self.stat[1] != 0) | ~torch.isfinite(norm)
        coef = torch.clamp(self.clip_norm / (norm + 1e-6), max=1.0) if self.clip_norm else torch.ones_like(norm)
        self.ctl[0] = (coef / self.world).float()
        self.ctl[1] = bad.float()
        self.skipped += bad.long()
        hp = self.hp
        for b in self.buckets:
            p, g, m, v, slow = self._shard_views(b)
            if native:
                with torch.cuda.device(self.device):
                    nat.check(L.fpc_lookahead_radam_step(nat.ptr(p), nat.ptr(g), nat.ptr(m), nat.ptr(v), nat.ptr(slow), p.numel(),
                                                         hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"],
                                                         self.step_count, hp["la_k"], hp["la_alpha"], nat.ptr(self.ctl),
                                                         nat.stream()), "fpc_lookahead_radam_step")
            else:
                self.step_fn(p, g, m, v, slow, self.step_count, self.ctl, hp)
        if self._collective:
            for b in self.buckets:
                p = self._shard_views(b)[0]
                dist.all_gather_into_tensor(self._bucket_param(b), p, group=self.group)
        # the kernel and the all-gather wrote the parameters through raw pointers / the flat buffer: tell autograd's
        # version counters, which is what NetEngine.stale() (and torch's own saved-tensor checks) look at
        torch.autograd.graph.increment_version([p for b in self.buckets for p in b["params"]])
        note = getattr(self._model, "note_unversioned_write", None)
        if note is not None:
            note()
        return norm

    def grad_norm_and_flag(self):
        """(norm of the mean gradient, non-finite flag) of the last step — device tensors, no synchronisation."""
        return torch.sqrt(self.stat[0]) / self.world, self.stat[1] != 0

    def state_bytes(self):
        return 4 * (self.m.numel() + self.v.numel() + self.slow.numel())
