"""placeholder (off the hot path)"""


class BVH:
    def __init__(self, *a, **k):
        raise NotImplementedError("stand-in: off the hot path")
