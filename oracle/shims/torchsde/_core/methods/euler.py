"""Stand-in for torchsde._core.methods.euler.Euler (stock step; cf. the commented copy at sdeint.py:447-465)."""
from .. import base_solver
from ...settings import LEVY_AREA_APPROXIMATIONS, NOISE_TYPES, SDE_TYPES


class Euler(base_solver.BaseSDESolver):
    weak_order = 1.0
    sde_type = SDE_TYPES.ito
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()

    def __init__(self, sde, **kwargs):
        self.strong_order = 1.0 if sde.noise_type == NOISE_TYPES.additive else 0.5
        super().__init__(sde=sde, **kwargs)

    def step(self, t0, t1, y0, extra0):
        del extra0
        dt = t1 - t0
        I_k = self.bm(t0, t1)
        f, g_prod = self.sde.f_and_g_prod(t0, y0, I_k)
        y1 = y0 + f * dt + g_prod
        return y1, ()
