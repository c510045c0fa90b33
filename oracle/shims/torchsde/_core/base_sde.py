"""Stand-in for torchsde._core.base_sde: BaseSDE and the stock ForwardSDE wrapper (f/g, diagonal, Ito)."""
from torch import nn

from ..settings import NOISE_TYPES, SDE_TYPES


class BaseSDE(nn.Module):
    def __init__(self, noise_type, sde_type):
        super().__init__()
        if noise_type not in NOISE_TYPES:
            raise ValueError(f"bad noise type {noise_type}")
        if sde_type not in SDE_TYPES:
            raise ValueError(f"bad sde type {sde_type}")
        self.noise_type = noise_type
        self.sde_type = sde_type


class ForwardSDE(BaseSDE):
    """Stock wrapper as used by torchsde.sdeint for an SDE exposing f and g (diagonal noise).

    Semantics restated in SURVEY.md App. A; the reference's own patched copy of this class
    (models/utils/sdeint.py:488-566) is the in-tree evidence for prod_diagonal / f_and_g_prod_default2.
    """

    def __init__(self, sde):
        super().__init__(sde_type=sde.sde_type, noise_type=sde.noise_type)
        self._base_sde = sde
        self.f = sde.f
        self.g = sde.g

    def f_and_g(self, t, y):
        return self.f(t, y), self.g(t, y)

    def prod(self, g, v):
        if self.noise_type != NOISE_TYPES.diagonal:
            raise NotImplementedError("stand-in covers diagonal noise only")
        return g * v

    def f_and_g_prod(self, t, y, v):
        f, g = self.f_and_g(t, y)
        return f, self.prod(g, v)


class RenameMethodsSDE(BaseSDE):
    def __init__(self, sde, **names):
        raise NotImplementedError("names= is not used on the hot path")


class SDELogqp(BaseSDE):
    def __init__(self, sde):
        raise NotImplementedError("logqp is False on the hot path (SURVEY a7)")
