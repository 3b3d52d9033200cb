"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) -- never the product.

numpy restatement of the reference's per-frame input chain, F/tools/dataset.py:249-262:
    sample = self.preprocessing(**sample)                  albu.Lambda(image=preprocessing_fn), transforms/pose_regression.py:22-28
    sample = numpy_to_torch()(**sample)                    transforms/general.py:7-8: x.transpose(2, 0, 1)
    sample['image'] /= np.max(np.abs(sample['image']))     (float64, in place)
    image = skimage.img_as_float32(sample['image'])        float -> float conversion: astype(float32)

preprocessing_fn is segmentation_models_pytorch's `preprocess_input` bound to the encoder's parameters
(smp.encoders.get_preprocessing_fn, F/tools/dataset.py:564-569; third-party, a git submodule the reference tree does not vendor and
the environment files do not pin; not importable here).  Its published algorithm is restated in `smp_preprocess_input`.
PARITY UNPINNED for this function: the reference holds no fixture or test for it and the upstream package cannot be run
here; the restatement follows the published source and numpy's IEEE double arithmetic.
"""
import numpy as np


def smp_preprocess_input(x, mean=None, std=None, input_space="RGB", input_range=None):
    """segmentation_models_pytorch/encoders/_preprocessing.py: BGR flip; x / 255.0 when the input range's upper bound is 1
    and x.max() > 1; subtract mean; divide by std (numpy broadcasting over the last axis, float64)."""
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


def preprocess_frame(image_u8, params):
    """image_u8 [H,W,3] uint8 -> float32 [3,H,W] exactly as NOCSDataset.__getitem__ leaves sample['image']."""
    x = smp_preprocess_input(image_u8, **params)
    x = x.transpose(2, 0, 1)
    if x.dtype != np.uint8:
        x = x / np.max(np.abs(x))
    return x.astype(np.float32)
