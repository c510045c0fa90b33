"""Import-only stand-in (the dataset's `get` path never builds a Quaternion)."""


class Quaternion:
    def __init__(self, *a, **k):
        raise NotImplementedError("pyquaternion stand-in: preprocessing is out of scope for the oracle")
