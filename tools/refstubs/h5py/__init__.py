"""h5py stand-in: the golden generator never saves."""


class File(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("h5py stand-in: saving is not available")
