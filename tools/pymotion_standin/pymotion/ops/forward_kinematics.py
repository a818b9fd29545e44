"""placeholder (off the hot path)"""


def fk(*a, **k):
    raise NotImplementedError("stand-in: off the hot path")
