"""
utils.py - the two file / JSON helpers user scripts import from qoc.standard
(qoc/standard/utils/fileutil.py:7-38, jsonutil.py:10-24): same names, arguments and results.
`ans_jacobian` has no counterpart: this package has no AD engine (see qoc_amd.models.Cost).
"""

import json
import os

import numpy as np


def generate_save_file_path(save_file_name, save_path):
    """
    Full path "<save_path>/<NNNNN>_<save_file_name>.h5" with NNNNN one more than the largest
    numeric prefix already used for that name in `save_path` (created if missing).
    """
    os.makedirs(save_path, exist_ok=True)
    suffix = "_{}.h5".format(save_file_name)
    taken = [int(name.split("_")[0]) for name in os.listdir(save_path) if suffix in name]
    return os.path.join(save_path, "{:05d}{}".format(max(taken, default=-1) + 1, suffix))


class CustomJSONEncoder(json.JSONEncoder):
    """json.dumps(..., cls=CustomJSONEncoder): NumPy integers, floats and arrays as JSON."""

    def default(self, obj):
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return json.JSONEncoder.default(self, obj)
