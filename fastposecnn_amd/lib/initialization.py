"""Parameter initialisation of the FPN decoders and the 1x1 heads.

Same scheme as the reference (F/lib/initialization.py:6-27, itself taken from segmentation_models_pytorch):
decoder convolutions He-uniform (fan-in, ReLU gain), normalisation layers to the identity, linear layers and
every head layer Glorot-uniform, all biases zero.  Table-driven: one rule per layer type.
"""
import torch.nn as nn
from torch.nn import init


def _he_uniform(weight):
    init.kaiming_uniform_(weight, mode="fan_in", nonlinearity="relu")


def _apply(module, weight_rules, norm_to_identity):
    """weight_rules: ((layer types, weight initialiser), ...) — first match wins; biases of matched layers -> 0."""
    for layer in module.modules():
        if norm_to_identity and isinstance(layer, nn.BatchNorm2d):
            init.ones_(layer.weight)
            init.zeros_(layer.bias)
            continue
        for kinds, fill in weight_rules:
            if isinstance(layer, kinds):
                fill(layer.weight)
                if getattr(layer, "bias", None) is not None:
                    init.zeros_(layer.bias)
                break


def initialize_decoder(module):
    _apply(module, ((nn.Conv2d, _he_uniform), (nn.Linear, init.xavier_uniform_)), norm_to_identity=True)


def initialize_head(module):
    _apply(module, (((nn.Linear, nn.Conv2d), init.xavier_uniform_),), norm_to_identity=False)
