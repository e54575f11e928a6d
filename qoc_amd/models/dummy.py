"""dummy.py - an attribute bag (the GRAPE `reporter`), as qoc/models/dummy.py:5-14."""


class Dummy(object):
    def __init__(self):
        super().__init__()
