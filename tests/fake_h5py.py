"""
fake_h5py.py - TEST INFRASTRUCTURE: a dictionary-backed stand-in for the few h5py calls the
save-file code makes (h5py is not installed in the build image). `File(path, mode)` is a context
manager; `f[name] = value` creates a dataset (a NumPy array), `f[name][index] = value` writes
into it. Data persists per path in STORE for the lifetime of the process.
"""

import numpy as np

STORE = {}


class _Dataset(object):
    def __init__(self, array):
        self.array = array

    def __setitem__(self, index, value):
        self.array[index] = value

    def __getitem__(self, index):
        return self.array[index]

    @property
    def shape(self):
        return self.array.shape


class File(object):
    def __init__(self, path, mode="r"):
        self.path = path
        if mode == "w":
            STORE[path] = {}
        elif path not in STORE:
            raise OSError("no such file: {}".format(path))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __setitem__(self, name, value):
        if name in STORE[self.path]:
            raise ValueError("dataset {} exists".format(name))
        if isinstance(value, str):
            value = np.array(value)
        STORE[self.path][name] = _Dataset(np.array(value))

    def __getitem__(self, name):
        return STORE[self.path][name]

    def keys(self):
        return STORE[self.path].keys()
