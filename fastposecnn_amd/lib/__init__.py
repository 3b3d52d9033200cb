"""Drop-in for the reference's `lib` package (F/lib/__init__.py:1-10).

Like the reference, the directory itself is put on sys.path and the modules are imported
under their bare names (`gpu_tensor_funcs`, `aggregation_layer`, `hough_voting`,
`pose_regressor`, `ransac_voting_gpu_layer.*`), so `train.py` / `evaluate.py` /
`inference.py` keep working with `import lib` pointed at this directory.
`matching.batchwise_find_matches`, `loss` and `metrics` (ranks 2 and 4: the classes
train.py selects) are shipped; the rest of matching.py is outside the hot path and is not.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)          # makes `fastposecnn_amd` importable when only lib/ was on the path
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import gpu_tensor_funcs as gtf  # noqa: E402
import matching as mg  # noqa: E402,F401   (the reference's alias, F/lib/__init__.py:8)
import aggregation_layer  # noqa: E402
import hough_voting  # noqa: E402
import pose_regressor  # noqa: E402
import matching  # noqa: E402
import loss  # noqa: E402
import metrics  # noqa: E402
