"""The input side of the inference path on the device: colour frame -> network tensor.

Mirrors what F/tools/dataset.py does per sample on the host in numpy, for frames that are already decoded:

    NOCSDataset.__getitem__ (:249-262)   preprocessing_fn(image) -> transpose(2,0,1) -> / max|.| -> img_as_float32
    my_collate_fn (:453-529)             stack the per-sample arrays, concatenate agg_data, add 'sample_ids'

`preprocess_frames` runs the first chain on the GPU (fpc_preprocess_u8: two kernels, bit-identical to the numpy
chain), `FrameUploader` owns the pinned staging buffers and the host->device copy stream in front of it, and
`my_collate_fn` keeps the reference's collate semantics for callers that bring their own samples.

File decoding (:158-176: skimage.io.imread of `*_color.png` / `*_mask.png`, cv2.imread of `*_depth.png`) is native too
(csrc/png_decode.hip over zlib; libpng's headers are not in the image): `imread_png` returns what imread returns,
`read_frame_files` mirrors the first lines of `__getitem__` (image, mask with background 255 -> 0, standardised depth),
and `FrameUploader.upload_png` decodes a batch of colour files straight into its pinned staging slot on a few host
threads.

The ground-truth half of a dataset item (round 4, SURVEY.md 8f rank 3 remainder): `NOCSDataset` / `CAMERADataset` /
`REALDataset` mirror F/tools/dataset.py:99-434 — the directory walk for `*_color.png` frames with wanted instances
(:278-357), `__getitem__` (:138-272: file reads, `*_meta+.json`, distractor and class filtering of the instance mask, the
per-instance ground truth of `generate_agg_data` :373-434, z <= 0 rejection, class mask, preprocessing chain) — and return the
reference's sample dict key for key (pinned by tests/golden/nocs_sample.npz, which the reference's own class wrote:
oracle/gen_golden.py).  `evaluate.py:148` and the training step feed `batchwise_find_matches` from exactly this dict.
Augmentation is not mirrored (the reference has it commented out, :239-242).
"""
import ctypes
import os
import pathlib

import numpy as np
import torch

from fastposecnn_amd import _native as nat
from fastposecnn_amd.tools import data_manipulation as dm
from fastposecnn_amd.tools import json_tools as jt

# segmentation_models_pytorch's preprocessing parameters for the resnet encoders with "imagenet" weights
# (smp.encoders.get_preprocessing_params; upstream package, absent from the reference tree): RGB, range [0, 1]
IMAGENET_PARAMS = {"input_space": "RGB", "input_range": [0, 1], "mean": [0.485, 0.456, 0.406],
                   "std": [0.229, 0.224, 0.225]}


def get_preprocessing_params(encoder_name="resnet18", pretrained="imagenet"):
    """smp.encoders.get_preprocessing_params for the encoders this package builds (lib/backbone.py)."""
    if pretrained != "imagenet" or not str(encoder_name).startswith("resnet"):
        raise ValueError("no preprocessing parameters for encoder %r / weights %r" % (encoder_name, pretrained))
    return {k: (list(v) if isinstance(v, list) else v) for k, v in IMAGENET_PARAMS.items()}


def preprocess_frames(images_u8, params=None, out=None):
    """images_u8: uint8 CUDA tensor [B,H,W,3] (or [H,W,3]) as skimage.io.imread returns frames ->
    f32 [B,3,H,W] ([3,H,W]), bit-identical to F/tools/dataset.py:249-262 applied to each frame."""
    params = params or IMAGENET_PARAMS
    if params.get("input_space", "RGB") != "RGB":
        raise ValueError("only RGB input space (the resnet encoders)")
    nat.require_gpu(images_u8, what="preprocess_frames")
    if images_u8.dtype != torch.uint8:
        # dataset.py:256 skips the max-normalisation for uint8 only because a preprocessing_fn always ran before;
        # float input has no reference behaviour to mirror here
        raise TypeError("preprocess_frames takes the decoded uint8 frame")
    single = images_u8.dim() == 3
    x = images_u8.unsqueeze(0) if single else images_u8
    if x.dim() != 4 or x.shape[-1] != 3:
        raise ValueError("expected [B,H,W,3] uint8, got %s" % (tuple(images_u8.shape),))
    x = x.contiguous()
    B, H, W, _ = x.shape
    dev = x.device
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    elif out.shape != (B, 3, H, W) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a contiguous f32 [B,3,H,W] tensor on the frames' device")
    mean = (ctypes.c_double * 3)(*params["mean"])
    std = (ctypes.c_double * 3)(*params["std"])
    rng = params.get("input_range")
    range01 = 1 if (rng is not None and rng[1] == 1) else 0
    L = nat.lib()
    with torch.cuda.device(dev):
        ws = nat.workspace("pre", dev, L.fpc_preprocess_workspace_bytes(B))
        nat.check(L.fpc_preprocess_u8(nat.ptr(x), B, H, W, mean, std, range01, nat.ptr(out), nat.ptr(ws), ws.numel(),
                                      nat.stream()), "fpc_preprocess_u8")
    return out[0] if single else out


def imread_png(src, rgb8=False):
    """A PNG file (path) or its bytes -> numpy array as skimage.io.imread / cv2.imread(path, -1) return it: [H,W] or
    [H,W,C] uint8, uint16 for 16-bit files, palette expanded to RGB (channel order R,G,B,A: cv2 would give B,G,R).
    rgb8=True: always [H,W,3] uint8 (grey replicated, alpha dropped)."""
    data = src if isinstance(src, (bytes, bytearray, memoryview)) else open(os.fspath(src), "rb").read()
    buf = np.frombuffer(data, dtype=np.uint8)
    L = nat.lib()
    info = (ctypes.c_int32 * 5)()
    nat.check(L.fpc_png_info(buf.ctypes.data, buf.size, info), "fpc_png_info")
    W, H, depth, ctype, C = (int(v) for v in info)
    if rgb8:
        out = np.empty((H, W, 3), np.uint8)
        nat.check(L.fpc_png_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.nbytes, 3), "fpc_png_decode")
        return out
    out = np.empty((H, W, C), np.uint16 if depth == 16 else np.uint8)
    nat.check(L.fpc_png_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.nbytes, 0), "fpc_png_decode")
    return out[:, :, 0] if C == 1 else out


def standardize_depth(depth_rgb_or_u16):
    """F/tools/data_manipulation.py:153-163 for an array in R,G,B order (the reference sees cv2's B,G,R: its channels
    [1], [2] are G, R): a 3-channel file encodes depth as G * 256 + R; a 16-bit grey file is the depth."""
    d = depth_rgb_or_u16
    if d.ndim == 3:
        return (d[:, :, 1].astype(np.uint16) * np.uint16(256) + d[:, :, 0].astype(np.uint16)).astype(np.uint16)
    if d.ndim == 2 and d.dtype == np.uint16:
        return d
    raise ValueError("unsupported depth image")


def read_frame_files(color_path, camera=True):
    """The file reads of NOCSDataset.__getitem__ (F/tools/dataset.py:158-176): {'image': u8 [H,W,3|4] as stored,
    'mask': float64 [H,W] with the background 255 set to 0 (first channel of the CAMERA set's RGBA masks),
    'depth': uint16 [H,W]}.  Missing mask / depth files are left out."""
    color_path = os.fspath(color_path)
    out = {"image": imread_png(color_path)}
    mask_fp = color_path.replace("_color.png", "_mask.png")
    if os.path.exists(mask_fp):
        m = imread_png(mask_fp)
        m = (m[:, :, 0] if (camera and m.ndim == 3) else m).astype("float")
        m[m == 255] = 0
        out["mask"] = m
    depth_fp = color_path.replace("_color.png", "_depth.png")
    if os.path.exists(depth_fp):
        out["depth"] = standardize_depth(imread_png(depth_fp))
    return out


class FrameUploader:
    """Host frames -> network tensors: pinned staging + asynchronous H2D copy + preprocess_frames, buffered `slots` deep
    on its own HIP stream so the copy of batch i+1 overlaps the network of batch i.  `upload(frames)` returns
    (tensor, event); wait on the event (stream.wait_event) before the first consumer kernel.

    Slot ownership: the tensor returned by an upload is overwritten `slots` uploads later.  Pass `consumed` (an event
    recorded after its last reader) to have the uploader wait for it on the device; without it the caller guarantees
    that the readers are done by then — FrameStreamer does when slots >= frames in flight + 2, because a ticket is
    collected (host wait on the frame) before more than that many newer frames are submitted.  (No event is recorded
    on the caller's stream by default: on the legacy null stream — PyTorch's default stream — every record / wait
    drags all other streams in, measured 1.6 ms of host time per upload beside four busy frame streams.)
    The staging copy is a single-threaded numpy copy: torch's intra-op thread pool takes milliseconds to wake up for a
    0.9 MB copy once the calling thread has been busy elsewhere (measured 4.9 ms per frame against 0.2 ms)."""

    def __init__(self, batch, height, width, device="cuda:0", slots=2, params=None):
        self.device = torch.device(device)
        self.params = params or IMAGENET_PARAMS
        self.stream = torch.cuda.Stream(device=self.device)
        shape = (batch, height, width, 3)
        self._host = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self._host_np = [h.numpy() for h in self._host]
        self._dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self._out = [torch.empty((batch, 3, height, width), dtype=torch.float32, device=self.device) for _ in range(slots)]
        self._free = [None] * slots          # event: the slot's previous output has been consumed
        self._busy = [None] * slots          # event: the slot's H2D copy has left the pinned buffer
        self._i = 0

    def upload(self, frames, consumed=None):
        """frames: uint8 array / CPU tensor [B,H,W,3].  `consumed`: event after which the tensor returned by THIS call
        may be overwritten (waited for on the device when the slot comes round again); see the class docstring."""
        k = self._i % len(self._host)
        self._i += 1
        src = frames if isinstance(frames, np.ndarray) else frames.numpy()
        if self._busy[k] is not None:
            self._busy[k].synchronize()      # the pinned buffer is about to be rewritten by the CPU
        np.copyto(self._host_np[k], src)
        return self._enqueue(k, consumed)

    def upload_png(self, files, consumed=None, threads=4):
        """files: B PNG files as bytes (`*_color.png`, all of the uploader's H x W): decoded by `threads` host threads
        straight into the pinned staging slot (fpc_png_decode_batch: RGB8, alpha dropped), then the same copy and kernels."""
        k = self._i % len(self._host)
        self._i += 1
        B, H, W, _ = self._host_np[k].shape
        if len(files) != B:
            raise ValueError(f"expected {B} files, got {len(files)}")
        if self._busy[k] is not None:
            self._busy[k].synchronize()
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
        ptrs = (ctypes.c_void_p * B)(*[b.ctypes.data for b in bufs])
        sizes = (ctypes.c_size_t * B)(*[b.size for b in bufs])
        nat.check(nat.lib().fpc_png_decode_batch(ptrs, sizes, B, self._host_np[k].ctypes.data, H, W, int(threads)),
                  "fpc_png_decode_batch")
        return self._enqueue(k, consumed)

    def _enqueue(self, k, consumed):
        with torch.cuda.stream(self.stream):
            if self._free[k] is not None:
                self.stream.wait_event(self._free[k])
            self._dev[k].copy_(self._host[k], non_blocking=True)
            self._busy[k] = torch.cuda.Event()
            self._busy[k].record()
            preprocess_frames(self._dev[k], self.params, out=self._out[k])
            done = torch.cuda.Event()
            done.record()
        self._free[k] = consumed
        return self._out[k], done


class PngFramePrefetcher:
    """Decoded frames ahead of the consumer: `workers` host threads run the native PNG decoder (ctypes releases the GIL)
    `ahead` batches in front of `next()`.  A 640x480 colour PNG takes ~10 ms of zlib + un-filtering on one core, the
    streamed pipeline consumes a frame every 0.8 ms: the pool, not the GPU, sets the rate from encoded files.

        pre = PngFramePrefetcher(lambda i: [bytes of the B files of batch i], n_batches, B, H, W)
        for frames in pre:            # uint8 [B, H, W, 3]
            tensor, ready = uploader.upload(frames)
    """

    def __init__(self, read_batch, n_batches, batch, height, width, workers=12, ahead=None):
        from concurrent.futures import ThreadPoolExecutor
        self.read_batch, self.n, self.shape = read_batch, int(n_batches), (batch, height, width, 3)
        self.ahead = int(ahead) if ahead else max(2, -(-2 * workers // max(1, batch)))
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._pending, self._next = [], 0
        nat.lib()

    def _decode_one(self, buf, dst):
        nat.check(nat.lib().fpc_png_decode(buf.ctypes.data, buf.size, dst.ctypes.data, dst.nbytes, 3), "fpc_png_decode")

    def _read_and_decode(self, i, out):
        """Pool task of batch i: read its files (the I/O of a cold page cache or slow storage stays off the consumer's
        thread), then one decode task per FILE (a batch of 32 decoded by one worker kept 2 of 14 workers busy: 350 img/s at
        batch 32).  Returns the per-file futures; the buffers stay referenced by them."""
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in self.read_batch(i)]
        if len(bufs) != out.shape[0]:
            raise ValueError(f"read_batch({i}) returned {len(bufs)} files for a batch of {out.shape[0]}")
        return [self._pool.submit(self._decode_one, b, out[j]) for j, b in enumerate(bufs)]

    def _submit(self, i):
        out = np.empty(self.shape, np.uint8)
        return out, self._pool.submit(self._read_and_decode, i, out)

    def __iter__(self):
        return self

    def __next__(self):
        while self._next < self.n and len(self._pending) < self.ahead:
            self._pending.append(self._submit(self._next))
            self._next += 1
        if not self._pending:
            self._pool.shutdown(wait=False)
            raise StopIteration
        out, reader = self._pending.pop(0)
        for f in reader.result():      # (a reader never waits for its decode tasks inside the pool: no worker can starve them)
            f.result()
        return out


# ------------------------------------------------------------------------------------------------ ground-truth samples

# F/tools/project.py:78-126
CAMERA_CLASSES = ['bg', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']
CAMERA_SYMMETRIC_CLASSES = ['bowl', 'can', 'bottle']
REAL_SYMMETRIC_CLASSES = ['bowl', 'can', 'bottle']
INTRINSICS = {'CAMERA': np.array([[577.5, 0, 319.5], [0., 577.5, 239.5], [0., 0., 1.]]),
              'REAL': np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])}


def smp_preprocess_input(x, mean=None, std=None, input_space="RGB", input_range=None, **kwargs):
    """segmentation_models_pytorch's `preprocess_input` (encoders/_preprocessing.py; upstream package, absent from the
    reference tree): the host-side form of what fpc_preprocess_u8 does on the device, float64 like numpy's defaults."""
    if input_space == "BGR":
        x = x[..., ::-1].copy()
    if input_range is not None:
        if x.max() > 1 and input_range[1] == 1:
            x = x / 255.0
    if mean is not None:
        x = x - np.array(mean)
    if std is not None:
        x = x / np.array(std)
    return x


def get_preprocessing_fn(encoder_name="resnet18", pretrained="imagenet"):
    """smp.encoders.get_preprocessing_fn (F/tools/dataset.py:567)."""
    import functools
    return functools.partial(smp_preprocess_input, **get_preprocessing_params(encoder_name, pretrained))


def get_preprocessing(preprocessing_fn):
    """F/tools/transforms/pose_regression.py:22-28 (an albumentations Compose of one Lambda on the image): a callable
    `sample = f(**sample)` that applies preprocessing_fn to sample['image'] and passes every other key through."""
    def apply(**sample):
        sample = dict(sample)
        sample['image'] = preprocessing_fn(sample['image'])
        return sample
    return apply


def to_tensor(x, **kwargs):
    """F/tools/transforms/general.py:7-8"""
    return x.transpose(2, 0, 1) if len(x.shape) == 3 else x


class NOCSDataset(torch.utils.data.Dataset):
    """F/tools/dataset.py:99-434.  dataset_dir: a pathlib.Path (or str) searched recursively for `*_color.png` frames
    whose `*_meta+.json` holds at least one instance of the wanted `classes` (names from CLASSES; default: all)."""

    CLASSES = CAMERA_CLASSES
    SYMMETRIC_CLASSES = CAMERA_SYMMETRIC_CLASSES
    INTRINSICS = INTRINSICS['CAMERA']
    CAMERA_MASKS = True            # masks are RGBA files whose first channel holds the instance ids (REAL: single channel)

    def __init__(self, dataset_dir, max_size=None, classes=None, augmentation=None, preprocessing=None):
        if classes is None:
            classes = self.CLASSES
        self.classes = classes
        self.class_values_map = {self.CLASSES.index(cls.lower()): self.classes.index(cls) for cls in self.classes}
        self.symmetric_classes = [self.classes.index(cls.lower()) for cls in self.SYMMETRIC_CLASSES if cls in self.classes]
        self.images_fps = self.get_image_paths_in_dir(pathlib.Path(dataset_dir), max_size=max_size)
        self.augmentation = augmentation
        self.preprocessing = preprocessing

    def __len__(self):
        return len(self.images_fps)

    def get_image_paths_in_dir(self, dir_path, max_size=None):
        """The frames of the data set in the reference's order (F/tools/dataset.py:295-352): directories level by level
        (breadth-first, children in the file system's listing order), inside a directory the `*color*.png` frames in
        listing order, kept when their side file lists an instance of a wanted class; the walk stops after the first
        directory that brings the count to `max_size`."""
        from collections import deque
        frames, todo = [], deque([dir_path])
        while todo and (max_size is None or len(frames) < max_size):
            here = todo.popleft()
            entries = list(here.iterdir())
            frames.extend(self.remove_empty_samples([e for e in entries if e.is_file() and e.suffix == '.png' and 'color' in e.name]))
            todo.extend(e for e in entries if e.is_dir())
        return frames if max_size is None else frames[:max_size]

    def remove_empty_samples(self, file_paths):
        good = []
        for fp in file_paths:
            json_data = jt.load_from_json(str(fp).replace('_color.png', '_meta+.json'))
            if any(class_value in self.class_values_map for class_value in json_data['instance_dict'].values()):
                good.append(fp)
        return good

    def __getitem__(self, i):
        color_fp = str(self.images_fps[i])
        image = imread_png(color_fp)
        mask = imread_png(color_fp.replace('_color.png', '_mask.png'))
        mask = (mask[:, :, 0] if self.CAMERA_MASKS else mask).astype('float')
        mask[mask == 255] = 0                                      # background
        depth = standardize_depth(imread_png(color_fp.replace('_color.png', '_depth.png')))
        json_data = jt.load_from_json(color_fp.replace('_color.png', '_meta+.json'))

        # Which pixels survive: instances the side file lists (distractor objects are not listed) whose class is one of
        # the wanted ones (F/tools/dataset.py:183-228).  Two 256-entry tables indexed by the mask's instance id — id -> id
        # and id -> class index — replace the reference's per-instance sweeps over the image.
        listed = [(int(k), c) for k, c in json_data['instance_dict'].items()]
        rows = [r for r, (_, c) in enumerate(listed) if c in self.class_values_map]
        id_of = np.zeros(256, dtype=mask.dtype)
        class_of = np.zeros(256, dtype=mask.dtype)
        good_json_data = {'instance_dict': {}}
        for r in rows:
            inst_id, c = listed[r]
            id_of[inst_id] = inst_id
            class_of[inst_id] = self.class_values_map[c]
            good_json_data['instance_dict'][inst_id] = self.class_values_map[c]
        if rows:
            for key, per_instance in json_data.items():
                if key != 'instance_dict':
                    good_json_data[key] = np.stack([per_instance[r] for r in rows])
        pixel_ids = mask.astype(np.intp)
        good_instances_mask = id_of[pixel_ids]

        agg_data = self.generate_agg_data(good_instances_mask, good_json_data)
        if (agg_data['z'] <= 0).any():                             # invalid / corrupt sample
            return None
        class_mask = class_of[pixel_ids]

        sample = {'clean_image': image, 'image': image, 'mask': class_mask, 'depth': depth}
        if self.preprocessing:
            sample = self.preprocessing(**sample)
        sample['image'] = to_tensor(sample['image'])               # numpy_to_torch(): the image target only
        if sample['image'].dtype != np.uint8:
            sample['image'] /= np.max(np.abs(sample['image']))
        sample.update({
            'path': self.images_fps[i],
            'image': sample['image'].astype(np.float32),           # skimage.img_as_float32 of a float array
            'mask': sample['mask'].astype('long'),
            'depth': sample['depth'].astype('float32'),
            'agg_data': agg_data,
        })
        return sample

    def get_random_batched_sample(self, batch_size=1, device=None):
        ids = np.random.choice(np.arange(len(self)), size=batch_size, replace=False)
        return my_collate_fn([self[int(k)] for k in ids], device)

    def generate_agg_data(self, instances_mask, json_data):
        """Per-instance ground truth, one row per entry of the side file's instance_dict in file order
        (F/tools/dataset.py:373-434): n = the number of instance ids present in the mask; class / symmetric ids, the
        instance's binary mask, quaternion, scales / norm factor, and from the RT matrices the projected origin (flipped to
        the other coordinate style), z, T and R (tools/data_manipulation.py).  Filled one fancy-indexed block per key."""
        n = np.unique(instances_mask).size - 1
        listed = list(json_data['instance_dict'].items())             # (instance id, class id)
        k = len(listed)
        inst = np.array([i for i, _ in listed]).reshape(k)
        cls = np.array([c for _, c in listed], dtype=np.float64).reshape(k)
        per_instance = dict(dm.extract_xyz_R_T_from_RTs(json_data['RTs'], self.INTRINSICS),      # xy, z, T, R
                            quaternion=json_data['quaternions'], scales=json_data['scales'], RT=json_data['RTs'])

        def block(values, *shape):
            out = np.zeros((n,) + shape)
            out[:k] = np.asarray(values, dtype=np.float64).reshape((-1,) + shape)[:k]
            return out

        agg_data = {
            'class_ids': block(cls), 'symmetric_ids': block(np.isin(cls, self.symmetric_classes)),
            'instance_masks': block(instances_mask[None] == inst[:, None, None], *instances_mask.shape),
            'quaternion': block(per_instance['quaternion'], 4),
            'scales': block(per_instance['scales'], 3) / np.expand_dims(json_data['norm_factors'], axis=1),
            'xy': block(per_instance['xy'], 2)[:, ::-1],              # (row, col) -> the other style
            'z': block(per_instance['z'], 1), 'T': block(per_instance['T'], 3),
            'R': block(per_instance['R'], 3, 3), 'RT': block(per_instance['RT'], 4, 4),
        }
        return agg_data


class CAMERADataset(NOCSDataset):
    pass


class REALDataset(NOCSDataset):
    SYMMETRIC_CLASSES = REAL_SYMMETRIC_CLASSES
    INTRINSICS = INTRINSICS['REAL']
    CAMERA_MASKS = False


def my_collate_fn(batch, device=None):
    """F/tools/dataset.py:453-529: drop None samples; stack array-valued keys; concatenate every agg_data entry along
    axis 0 and add agg_data['sample_ids'] (the sample index repeated once per instance)."""
    batch = [s for s in batch if s is not None]
    if not batch:
        return None
    columns, agg = {}, {"sample_ids": []}
    for sample_id, sample in enumerate(batch):
        for key, value in sample.items():
            if key != "agg_data":
                columns.setdefault(key, []).append(value)
        if "agg_data" in sample:
            for subkey, value in sample["agg_data"].items():
                agg.setdefault(subkey, []).append(value)
            agg["sample_ids"].append(np.repeat(np.array(sample_id), sample["agg_data"]["class_ids"].shape[0]))

    def put(a):
        t = torch.from_numpy(a)
        return t.to(device) if device else t

    out = {key: (put(np.stack(vals)) if isinstance(vals[0], np.ndarray) else vals) for key, vals in columns.items()}
    out["agg_data"] = {subkey: put(np.concatenate(vals, axis=0)) for subkey, vals in agg.items()}
    return out
