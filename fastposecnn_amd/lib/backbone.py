"""ResNet encoder + FPN decoder + segmentation head with segmentation_models_pytorch-compatible
module / parameter names.

The reference builds these from `segmentation_models_pytorch` (call sites
F/lib/pose_regressor.py:608-666), which is NOT vendored in /root/reference and is not installed
in this image, so the graph below is re-derived from upstream knowledge of smp at the commit the
reference cites (F/lib/pose_regressor.py:578-580) and could not be cross-checked here
("parity unpinned" for the conv stack — SURVEY.md section 7.2).  What IS checked: every op
against torch.nn.functional on the same tensors (tests/).

Structure (encoder_depth = 5, FPN, merge "add"):
  encoder(x) -> [x, relu(bn1(conv1 x)), layer1(maxpool .), layer2, layer3, layer4]
  FPNDecoder: p5 = 1x1(c5); p4 = up2_nearest(p5) + 1x1(c4); p3; p2;
              seg_blocks[i] = n_i x [conv3x3(no bias) -> GroupNorm(32) -> ReLU (-> up2 bilinear,
              align_corners=True)] with n = 3,2,1,0 upsamples (at least one conv);
              sum of the four 1/4-scale maps; Dropout2d(0.2)
  SegmentationHead: conv 1x1 -> UpsamplingBilinear2d(x4) (align_corners=True) -> Identity
state_dict names follow smp: encoder.conv1.weight, encoder.layer1.0.conv1.weight, ...,
<dec>.p5.weight, <dec>.p4.skip_conv.weight, <dec>.seg_blocks.0.block.0.block.0.weight (conv),
...block.1.weight (GroupNorm), <head>.0.weight.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose TRAINING-mode forward on a GPU runs on the native kernels, forward and backward
    (lib/train_conv.py); evaluation goes through the engine's plan (or torch, on CPU) as before.  Same parameters and
    state_dict names as nn.Conv2d."""

    def forward(self, x):
        if self.training and x.is_cuda and self.groups == 1 and self.dilation == (1, 1) and self.padding_mode == "zeros" \
                and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and not isinstance(self.padding, str):
            from fastposecnn_amd.lib import train_conv
            if train_conv.ENABLED:
                return train_conv.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0])
        return super().forward(x)


def _upsample_bilinear(x, scale, out_nchw=False):
    """Bilinear, align_corners=True.  Under autograd on a GPU (the training step) the native forward / backward kernels of
    lib/train_conv.py; plain torch otherwise (inference goes through the engine's plan and never gets here)."""
    if x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
        from fastposecnn_amd.lib import train_conv
        return train_conv.upsample_bilinear(x, scale, out_nchw)
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


class UpsamplingBilinear2d(nn.UpsamplingBilinear2d):
    """nn.UpsamplingBilinear2d of the heads; the training step's forward / backward run natively (NCHW result)."""

    def forward(self, x):
        if float(self.scale_factor) in (2.0, 4.0):
            return _upsample_bilinear(x, int(self.scale_factor), out_nchw=True)
        return super().forward(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


_RESNET_LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}


class ResNetEncoder(nn.Module):
    """torchvision-style ResNet (BasicBlock) without avgpool/fc, returning 6 feature levels."""

    def __init__(self, name="resnet18", in_channels=3, depth=5):
        super().__init__()
        if name not in _RESNET_LAYERS:
            raise KeyError(f"encoder {name!r} not available (have {sorted(_RESNET_LAYERS)})")
        layers = _RESNET_LAYERS[name]
        self.name = name
        self._depth = depth
        self.out_channels = (in_channels, 64, 64, 128, 256, 512)
        self.inplanes = 64
        self.conv1 = Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        feats = [x]
        x = self.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(self.maxpool(x))
        feats.append(x)
        x = self.layer2(x)
        feats.append(x)
        x = self.layer3(x)
        feats.append(x)
        x = self.layer4(x)
        feats.append(x)
        return feats


_WARNED_WEIGHTS = set()


def get_encoder(name, in_channels=3, depth=5, weights=None):
    """smp.encoders.get_encoder stand-in (call site F/lib/pose_regressor.py:608-613).

    smp downloads the torchvision ImageNet checkpoint for weights='imagenet' (every HPARAM preset asks for it,
    F/config.py).  There is no network here, so pretrained weights come from a local file:
    `FPC_ENCODER_WEIGHTS_DIR/<name>.pth` (or `FPC_ENCODER_WEIGHTS=<file>`) holding the torchvision ResNet state
    dict (keys conv1.weight, bn1.*, layer1.0.conv1.weight, ...; fc.* is dropped, as smp does).  When `weights` is
    requested and no file is configured the encoder stays randomly initialised and a warning says so once per
    encoder — a checkpoint loaded afterwards (load_from_ckpt) overwrites the encoder anyway."""
    import logging
    import os
    enc = ResNetEncoder(name, in_channels=in_channels, depth=depth)
    enc.requested_weights = weights
    enc.loaded_weights = None
    if weights is not None:
        path = os.environ.get("FPC_ENCODER_WEIGHTS")
        if not path and os.environ.get("FPC_ENCODER_WEIGHTS_DIR"):
            path = os.path.join(os.environ["FPC_ENCODER_WEIGHTS_DIR"], f"{name}.pth")
        if path:
            import torch
            sd = torch.load(path, map_location="cpu")
            sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
            enc.load_state_dict(sd)                  # strict: a wrong file fails loudly
            enc.loaded_weights = path
        elif (name, weights) not in _WARNED_WEIGHTS:
            _WARNED_WEIGHTS.add((name, weights))
            logging.getLogger('fastposecnn').warning(
                "encoder %s: ENCODER_WEIGHTS=%r requested but no local file is configured "
                "(FPC_ENCODER_WEIGHTS / FPC_ENCODER_WEIGHTS_DIR): the encoder is RANDOMLY initialised, unlike "
                "smp.get_encoder, until a checkpoint is loaded", name, weights)
    return enc


class Conv3x3GNReLU(nn.Module):
    def __init__(self, in_channels, out_channels, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.block = nn.Sequential(
            Conv2d(in_channels, out_channels, (3, 3), stride=1, padding=1, bias=False),
            nn.GroupNorm(32, out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        if self.training and x.is_cuda and torch.is_grad_enabled() and isinstance(self.block[2], nn.ReLU):
            # the training step: GroupNorm + ReLU as one native channel-last op behind the native convolution
            from fastposecnn_amd.lib import train_conv
            x = train_conv.groupnorm_relu(self.block[0](x), self.block[1])
        else:
            x = self.block(x)
        if self.upsample:
            x = _upsample_bilinear(x, 2)
        return x


class FPNBlock(nn.Module):
    def __init__(self, pyramid_channels, skip_channels):
        super().__init__()
        self.skip_conv = Conv2d(skip_channels, pyramid_channels, kernel_size=1)

    def forward(self, x, skip=None):
        c = self.skip_conv
        if c.training and x.is_cuda and torch.is_grad_enabled():      # the training step: lateral convolution + merge in one launch
            from fastposecnn_amd.lib import train_conv
            y = train_conv.conv2d_up_add(skip, c.weight, c.bias, x, c.padding[0])
            if y is not None:
                return y
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        skip = self.skip_conv(skip)
        return x + skip


class SegmentationBlock(nn.Module):
    def __init__(self, in_channels, out_channels, n_upsamples=0):
        super().__init__()
        blocks = [Conv3x3GNReLU(in_channels, out_channels, upsample=bool(n_upsamples))]
        if n_upsamples > 1:
            for _ in range(1, n_upsamples):
                blocks.append(Conv3x3GNReLU(out_channels, out_channels, upsample=True))
        self.block = nn.Sequential(*blocks)

    def forward(self, x):
        return self.block(x)


class MergeBlock(nn.Module):
    def __init__(self, policy):
        super().__init__()
        if policy not in ("add", "cat"):
            raise ValueError(f"`merge_policy` must be one of: ['add', 'cat'], got {policy}")
        self.policy = policy

    def forward(self, x):
        if self.policy == "add":
            return sum(x)
        return torch.cat(x, dim=1)


class FPNDecoder(nn.Module):
    def __init__(self, encoder_channels, encoder_depth=5, pyramid_channels=256, segmentation_channels=128,
                 dropout=0.2, merge_policy="add"):
        super().__init__()
        self.out_channels = segmentation_channels if merge_policy == "add" else segmentation_channels * 4
        if encoder_depth < 3:
            raise ValueError(f"Encoder depth for FPN decoder cannot be less than 3, got {encoder_depth}.")
        encoder_channels = encoder_channels[::-1]
        encoder_channels = encoder_channels[:encoder_depth + 1]
        self.p5 = Conv2d(encoder_channels[0], pyramid_channels, kernel_size=1)
        self.p4 = FPNBlock(pyramid_channels, encoder_channels[1])
        self.p3 = FPNBlock(pyramid_channels, encoder_channels[2])
        self.p2 = FPNBlock(pyramid_channels, encoder_channels[3])
        self.seg_blocks = nn.ModuleList([
            SegmentationBlock(pyramid_channels, segmentation_channels, n_upsamples=n) for n in [3, 2, 1, 0]
        ])
        self.merge = MergeBlock(merge_policy)
        self.dropout = nn.Dropout2d(p=dropout, inplace=True)

    def forward(self, *features):
        c2, c3, c4, c5 = features[-4:]
        p5 = self.p5(c5)
        p4 = self.p4(p5, c4)
        p3 = self.p3(p4, c3)
        p2 = self.p2(p3, c2)
        feature_pyramid = [seg_block(p) for seg_block, p in zip(self.seg_blocks, [p5, p4, p3, p2])]
        x = self.merge(feature_pyramid)
        x = self.dropout(x)
        return x


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, activation=None, upsampling=1):
        conv2d = Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2)
        up = UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity()
        if activation is not None:
            raise ValueError("only activation=None is used by FastPoseCNN (pose_regressor.py:596)")
        super().__init__(conv2d, up, nn.Identity())
