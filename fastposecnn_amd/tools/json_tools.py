"""F/tools/json_tools.py:36-77: the `*_meta+.json` side files of a NOCS frame (instance ids -> class ids, quaternions, RTs,
scales, norm factors; written by the reference's create_meta+.py:602-700)."""
import json

import numpy as np


class NumpyEncoder(json.JSONEncoder):
    """numpy arrays / scalars as lists / numbers."""

    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, np.generic):
            return obj.item()
        return json.JSONEncoder.default(self, obj)


def _path(file_path):
    file_path = file_path if isinstance(file_path, str) else str(file_path)
    assert file_path.endswith('.json'), 'Given file_path is invalid for a json file'
    return file_path


def save_to_json(file_path, data):
    with open(_path(file_path), 'w') as outfile:
        json.dump(data, outfile, cls=NumpyEncoder)


def load_from_json(file_path):
    with open(_path(file_path), 'r') as infile:
        return json.load(infile)
