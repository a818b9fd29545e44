"""placeholder (off the hot path)"""


def to_root_dual_quat(*a, **k):
    raise NotImplementedError("stand-in: off the hot path")


def from_root_dual_quat(*a, **k):
    raise NotImplementedError("stand-in: off the hot path")
