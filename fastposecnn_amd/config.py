"""HPARAM presets restricted to the fields the hot path reads ("Config / flags").
Mirrors the attribute names and values of F/config.py:11-160 (DEFAULT_POSE_HPARAM, MASK_TRAINING,
HEAD_TRAINING, EVALUATING, INFERENCE) and the CAMERA constants of F/tools/project.py:78-88 so the
benchmarks and tests can build a model without the reference's `tools` package."""
import argparse

import numpy as np

CAMERA_INTRINSICS = np.array([[577.5, 0, 319.5], [0., 577.5, 239.5], [0., 0., 1.]])
CAMERA_CLASSES = ['bg', 'bottle', 'bowl', 'camera', 'can', 'laptop', 'mug']


class DEFAULT_POSE_HPARAM(argparse.Namespace):
    RUNTIME_TIMING = False
    MODEL = 'PoseRegressor'
    DATASET_NAME = 'CAMERA'
    SELECTED_CLASSES = CAMERA_CLASSES
    NUMPY_INTRINSICS = CAMERA_INTRINSICS
    BATCH_SIZE = 3

    FREEZE_ENCODER = False
    FREEZE_MASK_TRAINING = False
    FREEZE_ROTATION_TRAINING = False
    FREEZE_TRANSLATION_TRAINING = False
    FREEZE_SCALES_TRAINING = False

    PERFORM_AGGREGATION = True
    PERFORM_HOUGH_VOTING = True
    PERFORM_RT_CALCULATION = True
    PERFORM_MATCHING = True

    BACKBONE_ARCH = 'FPN'
    ENCODER = 'resnet18'
    ENCODER_WEIGHTS = 'imagenet'

    HV_NUM_OF_HYPOTHESES = 128


class MASK_TRAINING(DEFAULT_POSE_HPARAM):
    FREEZE_ROTATION_TRAINING = True
    FREEZE_TRANSLATION_TRAINING = True
    FREEZE_SCALES_TRAINING = True
    PERFORM_AGGREGATION = False
    PERFORM_HOUGH_VOTING = False
    PERFORM_RT_CALCULATION = False
    PERFORM_MATCHING = False


class HEAD_TRAINING(DEFAULT_POSE_HPARAM):
    pass


class EVALUATING(DEFAULT_POSE_HPARAM):
    HV_NUM_OF_HYPOTHESES = 1000


class INFERENCE(DEFAULT_POSE_HPARAM):
    HV_NUM_OF_HYPOTHESES = 1000
    BATCH_SIZE = 1
    RUNTIME_TIMING = True
