"""Stand-in Brownian motion: hands out injected increments (see oracle/noise_source.py)."""
import torch

from noise_source import SOURCE
from ..settings import LEVY_AREA_APPROXIMATIONS


class BaseBrownian:
    pass


class BrownianInterval(BaseBrownian):
    """bm(t0, t1) -> W(t1)-W(t0) of shape `size`, iid N(0, t1-t0) per element."""

    def __init__(self, t0=0.0, t1=1.0, size=None, dtype=torch.float32, device="cpu", entropy=None,
                 levy_area_approximation=LEVY_AREA_APPROXIMATIONS.none, **unused):
        self.shape = tuple(size)
        self.dtype = dtype
        self.device = device
        self.levy_area_approximation = levy_area_approximation

    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        return SOURCE.brownian_increment(self.shape, ta, tb, dtype=self.dtype, device=self.device, tag="bm")
