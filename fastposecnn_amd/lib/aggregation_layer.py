"""Drop-in for F/lib/aggregation_layer.py (AggregationLayer, :33-183) on libfpc_hip.so.

forward(cat_data) returns the same AggData keys, dtypes and instance order as the reference
(instances numbered by connected-component label: image by image, raster order of each
component's first pixel).  One device->host read (the instance count N, needed to shape the
outputs) replaces the reference's per-sample torch.unique / boolean-index synchronisations and
the torch->cupy->torch hop.
"""
import torch
import torch.nn as nn

from fastposecnn_amd import _native as nat

import hough_voting as hv


def mask_bits_of(masks):
    """The bit-word form of `masks` written by the aggregation call that produced this very tensor, or None (any other
    tensor, a slice / copy of it, or the tensor after an in-place write)."""
    tag = getattr(masks, "_fpc_mask_bits", None)
    if tag is None or tag[1] != masks._version or tag[0].shape[0] != masks.shape[0]:
        return None
    return tag[0]


def fg_bits_of(class_mask):
    """The foreground bit words the class compression wrote beside this very mask tensor (engine.NetEngine.forward,
    gtf.class_compression_fused), or None: any other tensor, a copy / slice / cast of it, or the tensor after an in-place
    write.  `attach_fg_bits` builds them for a mask that came from somewhere else (one launch)."""
    tag = getattr(class_mask, "_fpc_fg_bits", None)
    if tag is None or tag[1] != class_mask._version or tag[0].shape[0] != class_mask.shape[0]:
        return None
    return tag[0]


def attach_fg_bits(class_mask):
    """i64 [B,H,W] class mask -> the same tensor with its foreground bit words attached (fpc_fg_bits), where the labelling
    supports them; a fixture's stand-in for what the class compression does on the way."""
    nat.require_gpu(class_mask, what="attach_fg_bits")
    B, H, W = class_mask.shape
    L = nat.lib()
    if class_mask.dtype != torch.int64 or not class_mask.is_contiguous() or not L.fpc_cc_bits_supported(B, H, W):
        return class_mask
    bits = torch.empty((B, L.fpc_mask_bits_words(H, W)), dtype=torch.int64, device=class_mask.device)
    with torch.cuda.device(class_mask.device):
        nat.check(L.fpc_fg_bits(nat.ptr(class_mask), B, H, W, nat.ptr(bits), nat.stream()), "fpc_fg_bits")
    class_mask._fpc_fg_bits = (bits, class_mask._version)
    return class_mask


class AggregationLayer(nn.Module):

    def __init__(self, HPARAM, classes):
        super().__init__()
        self.HPARAM = HPARAM
        self.classes = classes  # including background
        self.hough_voting_layer = hv.HoughVotingLayer(self.HPARAM)
        # 4-connectivity inside an image, nothing across the batch axis (reference :43-59)
        self.s = torch.zeros((3, 3, 3), dtype=torch.bool)
        self.s[1, 1, :] = True
        self.s[1, :, 1] = True

    def batchwise_break_segmentation_mask(self, class_mask, return_device_count=False):
        """class_mask: bool/int [B,H,W] (foreground where != 0) -> (labels i32 [B,H,W], N)."""
        nat.require_gpu(class_mask, what="AggregationLayer")
        B, H, W = class_mask.shape
        dev = class_mask.device
        fg_bits = fg_bits_of(class_mask)
        cm = class_mask if class_mask.dtype == torch.int64 else class_mask.to(torch.int64)
        cm = cm.contiguous()
        labels = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        n_dev = torch.empty(1, dtype=torch.int32, device=dev)      # always written by fpc_cc_label
        # the first pixel of every component (labels 1..cap): tells the aggregation which image an instance lives in without
        # waiting for its sums; rides on the labels tensor (see _aggregate)
        root_pix = torch.empty(max(256, 64 * B), dtype=torch.int32, device=dev)
        L = nat.lib()
        with torch.cuda.device(dev):
            ws = nat.workspace("cc", dev, L.fpc_cc_workspace_bytes(B, H, W))
            if fg_bits is not None and L.fpc_cc_bits_supported(B, H, W):
                # the class compression left the foreground as bit words: the labelling never reads the i64 mask
                nat.check(L.fpc_cc_label_bits(nat.ptr(fg_bits), B, H, W, nat.ptr(labels), nat.ptr(n_dev), nat.ptr(root_pix),
                                              root_pix.numel(), nat.ptr(ws), ws.numel(), nat.stream()), "fpc_cc_label_bits")
            else:
                nat.check(L.fpc_cc_label(nat.ptr(cm), B, H, W, nat.ptr(labels), nat.ptr(n_dev), nat.ptr(root_pix), root_pix.numel(),
                                         nat.ptr(ws), ws.numel(), nat.stream()), "fpc_cc_label")
        labels._fpc_root_pix = (root_pix, labels._version)
        if return_device_count:
            return labels, n_dev
        return labels, int(n_dev.item())

    def _aggregate(self, cat_data, cm, labels, N, n_dev, stats=None):
        """Shared body: N is the exact count (n_dev None) or a capacity gated on the device by n_dev.
        `stats` (f32 [N,2], optional) receives each instance's pixel count and mean-quaternion norm (training backward)."""
        dev = cm.device
        B, H, W = cm.shape
        f32 = dict(dtype=torch.float32, device=dev)
        out = {
            'class_ids': torch.empty((N,), dtype=torch.int64, device=dev),
            'instance_masks': torch.empty((N, H, W), **f32),
            'sample_ids': torch.empty((N,), dtype=torch.int64, device=dev),
            'quaternion': torch.empty((N, 4), **f32),
            'scales': torch.empty((N, 3), **f32),
            'xy': torch.empty((N, 2, H, W), **f32),
            'z': torch.empty((N, 1), **f32),
        }
        if N == 0:
            # the reference returns float class ids for an empty batch (aggregation_layer.py:116)
            out['class_ids'] = torch.empty((0,), **f32)
            return out
        q, s, xy, z = (t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
                       for t in (cat_data['quaternion'], cat_data['scales'], cat_data['xy'], cat_data['z']))
        L = nat.lib()
        with torch.cuda.device(dev):
            ws = nat.workspace("agg", dev, L.fpc_aggregate_workspace_bytes(N))
            # the masks also as bit words (1/32 of the bytes) for the vote's scan, which then skips the f32 planes this call
            # has just written; they ride on the masks tensor (see mask_bits_of), not in the dict the reference defines
            bits = torch.empty((N, L.fpc_mask_bits_words(H, W)), dtype=torch.int64, device=dev)
            tag = getattr(labels, "_fpc_root_pix", None)
            root_pix = tag[0] if (tag is not None and tag[1] == labels._version and tag[0].numel() >= N) else None
            nat.check(L.fpc_aggregate_bits(nat.ptr(labels), nat.ptr(cm), nat.ptr(q), nat.ptr(s), nat.ptr(xy), nat.ptr(z),
                                           B, H, W, N, nat.ptr(n_dev), nat.ptr(out['class_ids']), nat.ptr(out['sample_ids']),
                                           nat.ptr(out['instance_masks']), nat.ptr(out['quaternion']),
                                           nat.ptr(out['scales']), nat.ptr(out['z']), nat.ptr(out['xy']), nat.ptr(stats),
                                           nat.ptr(bits), nat.ptr(root_pix), nat.ptr(ws), ws.numel(), nat.stream()), "fpc_aggregate_bits")
            out['instance_masks']._fpc_mask_bits = (bits, out['instance_masks']._version)
        return out

    def forward(self, cat_data):
        cat_mask = cat_data['mask']
        nat.require_gpu(cat_mask, what="AggregationLayer")
        cm = cat_mask.to(torch.int64).contiguous()
        labels, N = self.batchwise_break_segmentation_mask(cm)      # one host read: N shapes the outputs
        return self._aggregate(cat_data, cm, labels, N, None)

    def forward_deferred(self, cat_data, capacity):
        """Same result in the first n rows of `capacity`-row tensors, WITHOUT reading n back: returns
        (agg_data, n_dev).  Used by the model's fused post-network path, which enqueues voting and RT
        behind it and synchronises once at the end (lib/pose_regressor.py)."""
        cat_mask = cat_data['mask']
        nat.require_gpu(cat_mask, what="AggregationLayer")
        cm = cat_mask.to(torch.int64).contiguous()
        labels, n_dev = self.batchwise_break_segmentation_mask(cm, return_device_count=True)
        return self._aggregate(cat_data, cm, labels, int(capacity), n_dev), n_dev
