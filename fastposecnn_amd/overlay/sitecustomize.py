"""Zero-edit overlay for the reference's scripts.

    PYTHONPATH=/path/to/this/repo/fastposecnn_amd/overlay  python inference.py ...      # or evaluate.py / train.py, unchanged

The reference's scripts say `import lib` (F/inference.py:19, F/evaluate.py, F/train.py) and find the package next to the script,
because Python puts the script's directory in front of PYTHONPATH.  A `sitecustomize` module, however, is imported by the
interpreter itself before the script runs: this one installs a meta-path finder that answers the top-level name `lib` with
`fastposecnn_amd/lib/` (same module names and call signatures: lib.gtf, lib.mg, lib.pose_regressor, ...), so the scripts need
no edit.  Nothing is imported until the script's own `import lib` (no torch start-up cost for unrelated Python processes);
FPC_OVERLAY=0 switches the overlay off without touching PYTHONPATH.
"""
import importlib.abc
import importlib.util
import os
import sys

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")


class _LibFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "lib" or os.environ.get("FPC_OVERLAY", "1") == "0":
            return None
        return importlib.util.spec_from_file_location("lib", os.path.join(_LIB_DIR, "__init__.py"),
                                                      submodule_search_locations=[_LIB_DIR])


if not any(isinstance(f, _LibFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _LibFinder())
