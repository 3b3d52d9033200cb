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


def pack_pose_records(agg, sample_offset, capacity):
    """AggData (after RT calculation) -> f32 [capacity + 1, 40]; rank-local sample ids are shifted
    by `sample_offset` (the index of the shard's first image) so ids are global."""
    n = int(agg["class_ids"].shape[0])
    if n > capacity:
        raise RuntimeError(f"pose record capacity {capacity} exceeded by {n} instances")
    dev = agg["quaternion"].device
    if dev.type == "cuda":            # one native launch (fpc_pack_pose_records) instead of ~15 small torch ops per frame
        from fastposecnn_amd import _native as nat
        buf = torch.empty((capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=dev)
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


def all_gather_pose_records(agg, sample_offset, capacity, group=None, out=None):
    """One collective: every rank receives every rank's records. Returns [world, capacity+1, 40]."""
    buf = pack_pose_records(agg, sample_offset, capacity)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, capacity + 1, RECORD_WIDTH), dtype=torch.float32, device=buf.device)
    dist.all_gather_into_tensor(out.view(-1), buf.view(-1), group=group)
    return out
