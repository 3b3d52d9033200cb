"""Image-sharded inference across the GPUs of one node.

The path shards by image: no stage looks across images (connected components are per image,
RANSAC and RT per instance), so each rank runs the whole path on its own images and the only
exchange is ONE fixed-capacity all-gather of per-instance pose records (RCCL over xGMI with
backend "nccl"; "gloo" in the CPU tests).  A record is 40 x 4 bytes:
    sample_id(i32) class_id(i32) quaternion[4] scales[3] xy[2] z[1] R[9] T[3] RT[16]
Row 0 of every rank's buffer carries its instance count, so the result needs no second
collective and no variable-length gatherv.  Nothing like this exists in the reference (its
evaluate/inference scripts are single-GPU, F/evaluate.py:90,127).
"""
import torch
import torch.distributed as dist

RECORD_WIDTH = 40
_FIELDS = (("quaternion", 4), ("scales", 3), ("xy", 2), ("z", 1), ("R", 9), ("T", 3), ("RT", 16))


def shard_indices(num_images, rank, world_size):
    """Contiguous block of image indices owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(num_images, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def pack_pose_records(agg, sample_offset, capacity, out=None):
    """AggData (after RT calculation) -> f32 [capacity + 1, 40]; rank-local sample ids are shifted
    by `sample_offset` (the index of the shard's first image) so ids are global.  `out`: a contiguous
    [capacity + 1, 40] f32 tensor to fill instead of a new one (PoseGatherer's staging slots)."""
    n = int(agg["class_ids"].shape[0])
    if n > capacity:
        raise RuntimeError(f"pose record capacity {capacity} exceeded by {n} instances")
    dev = agg["quaternion"].device
    if dev.type == "cuda":            # one native launch (fpc_pack_pose_records) instead of ~15 small torch ops per frame
        from fastposecnn_amd import _native as nat
        buf = out if out is not None else torch.empty((capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=dev)
        f = lambda k: agg[k].contiguous().float()
        t = [agg["sample_ids"].contiguous().long(), agg["class_ids"].contiguous().long()] + [f(k) for k, _ in _FIELDS]
        with torch.cuda.device(dev):
            nat.check(nat.lib().fpc_pack_pose_records(*[nat.ptr(x) for x in t], n, int(sample_offset), int(capacity), nat.ptr(buf),
                                                      nat.stream()), "fpc_pack_pose_records")
        return buf
    buf = torch.zeros((capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=dev)
    buf[0, 0] = torch.tensor(n, dtype=torch.int32).view(torch.float32)
    if n:
        ids = torch.stack([agg["sample_ids"].to(torch.int32) + int(sample_offset),
                           agg["class_ids"].to(torch.int32)], dim=1)
        cols = [ids.view(torch.float32)] + [agg[k].reshape(n, w).float() for k, w in _FIELDS]
        buf[1:n + 1] = torch.cat(cols, dim=1)
    if out is not None:
        out.copy_(buf)
        return out
    return buf


def unpack_pose_records(gathered):
    """f32 [world, capacity + 1, 40] -> dict of concatenated per-instance tensors (rank order)."""
    parts = []
    for r in range(gathered.shape[0]):
        n = int(gathered[r, 0, 0].view(torch.int32))
        parts.append(gathered[r, 1:n + 1])
    rec = torch.cat(parts, dim=0) if parts else gathered.new_zeros((0, RECORD_WIDTH))
    n = rec.shape[0]
    ids = rec[:, :2].contiguous().view(torch.int32)
    out = {"sample_ids": ids[:, 0].to(torch.int64), "class_ids": ids[:, 1].to(torch.int64)}
    c = 2
    for k, w in _FIELDS:
        out[k] = rec[:, c:c + w]
        c += w
    out["R"] = out["R"].reshape(n, 3, 3); out["RT"] = out["RT"].reshape(n, 4, 4)
    return out


def _all_gather_flat(dst, src, group=None):
    """all_gather_into_tensor; gloo (tests: several ranks sharing one GPU) has no device path for it, so device tensors take
    a host round trip there.  RCCL ("nccl") gathers in place on the device."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(dst.numel(), dtype=dst.dtype)
        dist.all_gather_into_tensor(h, src.reshape(-1).cpu(), group=group)
        dst.view(-1).copy_(h)
        return
    dist.all_gather_into_tensor(dst.view(-1), src.view(-1), group=group)


def all_gather_pose_records(agg, sample_offset, capacity, group=None, out=None):
    """One collective: every rank receives every rank's records. Returns [world, capacity+1, 40]."""
    buf = pack_pose_records(agg, sample_offset, capacity)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=buf.device)
    _all_gather_flat(out, buf, group)
    return out


class PoseGatherer:
    """The pose gather off the frames' critical path ("issue once per batch on a side stream").

    `add(agg, sample_offset)` packs one frame's records into the next slot of a staging buffer (one native launch on the
    caller's current stream, no wait); every `every` frames ONE all-gather of the whole staging buffer runs on a side
    stream behind the packs' events, while the next frames pack into the other staging buffer.  No frame waits for RCCL; a
    consumer calls `latest()` (waits for the last issued collective only).

    Lockstep: a collective is a rendezvous, so every rank must issue the SAME number of them whatever its own frame count
    (`shard_indices` hands out shards that differ by one image: 17 images on 4 ranks with every = 4 is 5 / 4 / 4 / 4 frames).
    `flush()` therefore ALWAYS issues one collective — with every slot marked empty when this rank has nothing pending — and
    `finish(rounds_for(num_images, world, every))` tops a rank up to the job's round count with empty rounds.  Call `flush` /
    `finish` on every rank alike.

    Streams: frames may be packed from different streams (FrameStreamer's network streams).  Every pack records an event
    on its own stream and the side stream waits for all of them; every `add` orders its pack behind the collective that last
    read the staging buffer.  `latest()` returns a view of a result buffer that the collective two rounds later
    overwrites: clone it to keep it.  With one rank, or on CPU tensors (gloo tests), the same calls run inline."""

    def __init__(self, capacity, every=4, group=None, device=None, always_collective=False):
        """always_collective: issue the all-gather also in a process group of ONE rank (RCCL copies; the single-GPU test of
        the device path, tests/test_gpu_rccl_one_rank.py) instead of the plain copy a lone rank normally takes."""
        self.capacity, self.every, self.group = int(capacity), max(1, int(every)), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._collective = self.world > 1 or (bool(always_collective) and dist.is_initialized())
        self.device = device
        self._stage = None            # two staging buffers [every, capacity + 1, 40]
        self._out = None              # two results [world, every, capacity + 1, 40]
        self._cur, self._fill = 0, 0
        self._stream = None
        self._done = [None, None]     # events: the collective that reads staging buffer i has finished
        self._packed = []             # events: the packs of the current round, each on the stream it ran on
        self._last = None             # index of the most recent collective's buffers
        self.last_frames = 0
        self.collectives = 0

    @property
    def pending(self):
        """Frames packed since the last collective."""
        return self._fill

    @staticmethod
    def rounds_for(num_images, world_size, every):
        """Collectives a job of `num_images` images needs on EVERY rank: the largest shard's frames in rounds of `every`."""
        largest = -(-int(num_images) // max(1, int(world_size)))
        return -(-largest // max(1, int(every)))

    def _ensure(self, dev):
        if self._stage is None:
            self.device = dev = torch.device(dev)
            shape = (self.every, self.capacity + 1, RECORD_WIDTH)
            self._stage = [torch.zeros(shape, dtype=torch.float32, device=dev) for _ in range(2)]
            self._out = [torch.zeros((self.world,) + shape, dtype=torch.float32, device=dev) for _ in range(2)]
            if dev.type == "cuda":
                self._stream = torch.cuda.Stream(device=dev)

    def add(self, agg, sample_offset):
        dev = agg["quaternion"].device
        self._ensure(dev)
        if self._stream is not None:
            cur = torch.cuda.current_stream(dev)
            if self._done[self._cur] is not None:
                # this staging buffer was read by the collective issued two rounds ago: order this stream's pack behind it
                # (on the stream, no host wait; every add, because each may come from a different stream)
                cur.wait_event(self._done[self._cur])
        pack_pose_records(agg, sample_offset, self.capacity, out=self._stage[self._cur][self._fill])
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._packed.append(ev)
        self._fill += 1
        if self._fill == self.every:
            self._issue()

    def _issue(self):
        i, frames = self._cur, self._fill
        src, dst = self._stage[i], self._out[i]
        if self._stream is not None:
            cur = torch.cuda.current_stream(self.device)
            if frames < self.every:                               # a partial or empty round: mark the unused slots empty
                if self._done[i] is not None:
                    cur.wait_event(self._done[i])
                src[frames:, 0, 0] = 0
            ready = torch.cuda.Event()
            ready.record(cur)
            with torch.cuda.stream(self._stream):
                for ev in self._packed:
                    self._stream.wait_event(ev)
                self._stream.wait_event(ready)
                if self._collective:
                    _all_gather_flat(dst, src, self.group)
                else:
                    dst[0].copy_(src)
                done = torch.cuda.Event()
                done.record(self._stream)
            self._done[i] = done
            self._packed = []
        else:
            if frames < self.every:
                src[frames:, 0, 0] = 0
            if self._collective:
                _all_gather_flat(dst, src, self.group)
            else:
                dst[0].copy_(src)
        self.collectives += 1
        self._last = i
        self.last_frames = frames         # slots of the most recent collective THIS rank filled
        self._cur, self._fill = 1 - i, 0

    def flush(self, device=None):
        """ONE collective for the frames added since the last one — also when there are none (every slot empty): the
        number of collectives must not depend on this rank's frame count.  Call on every rank the same number of times.
        `device`: needed only when this rank never added a frame."""
        if self._stage is None:
            if device is None and self.device is None:
                raise RuntimeError("PoseGatherer.flush before any add: pass the device")
            self._ensure(device if device is not None else self.device)
        self._issue()

    def finish(self, total_rounds, device=None):
        """Flush the pending frames and top up with empty rounds until this rank has issued `total_rounds` collectives
        (`rounds_for(num_images, world, every)`): uneven shards stay in lockstep."""
        if self._fill:
            self.flush(device)
        if self.collectives > total_rounds:
            raise RuntimeError(f"{self.collectives} collectives issued, the job has {total_rounds} rounds")
        while self.collectives < total_rounds:
            self.flush(device)

    def latest(self):
        """Records of the most recent collective: [world, every, capacity + 1, 40]; slots a rank did not fill carry count 0
        (waits for that collective only; a view the collective two rounds later overwrites)."""
        if self._last is None:
            return None
        i = self._last
        if self._done[i] is not None:
            self._done[i].synchronize()
        return self._out[i]
