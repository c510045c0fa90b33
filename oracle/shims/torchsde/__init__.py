"""Stand-in for torchsde==0.2.5 (env.yml:293) -- TEST INFRASTRUCTURE, build-authored.

Only what the TrajSDE hot path touches: fixed-step Euler-Maruyama, diagonal Ito noise, f/g SDEs.
Semantics restated from SURVEY.md App. A; Brownian increments come from oracle/noise_source.py so that
the reference can be run here with injected noise.
"""
import torch
from torch import nn

from . import _brownian, _core, settings, types  # noqa: F401
from ._brownian import BaseBrownian, BrownianInterval  # noqa: F401
from ._core import misc
from ._core.base_sde import BaseSDE, ForwardSDE
from ._core.methods import select
from .settings import LEVY_AREA_APPROXIMATIONS, METHODS, NOISE_TYPES, SDE_TYPES


class SDEIto(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type=SDE_TYPES.ito)


class SDEStratonovich(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type=SDE_TYPES.stratonovich)


def sdeint(sde, y0, ts, bm=None, method=None, dt=1e-3, adaptive=False, rtol=1e-5, atol=1e-4, dt_min=1e-5,
           options=None, names=None, logqp=False, extra=False, extra_solver_state=None, **unused_kwargs):
    """Stock driver: check_contract -> ForwardSDE -> BrownianInterval -> Euler.integrate.

    Mirrors the order of operations the reference's patched copy keeps (models/utils/sdeint.py:174-197,
    913-921, 973-984): one shape probe each of f and g at ts[0] (no randomness consumed), then the solver.
    """
    misc.handle_unused_kwargs(unused_kwargs, msg="`sdeint`")
    if names or logqp or extra or adaptive:
        raise NotImplementedError("stand-in covers the plain fixed-step call of dec_hivt_nusargo_sde.py:88")
    if not torch.is_tensor(ts):
        ts = torch.tensor(ts, dtype=y0.dtype, device=y0.device)
    if not misc.is_strictly_increasing(ts):
        raise ValueError("Evaluation times `ts` must be strictly increasing.")
    f_shape = tuple(sde.f(ts[0], y0).size())
    g_shape = tuple(sde.g(ts[0], y0).size())
    if f_shape != tuple(y0.shape) or g_shape != tuple(y0.shape):
        raise ValueError("drift/diffusion shapes must equal the state shape for diagonal noise")
    fsde = ForwardSDE(sde)
    if bm is None:
        bm = BrownianInterval(t0=ts[0], t1=ts[-1], size=(y0.size(0), g_shape[1]), dtype=y0.dtype, device=y0.device,
                              levy_area_approximation=LEVY_AREA_APPROXIMATIONS.none)
    solver = select(method or METHODS.euler, fsde.sde_type)(
        sde=fsde, bm=bm, dt=dt, adaptive=adaptive, rtol=rtol, atol=atol, dt_min=dt_min, options=options or {})
    ys, _ = solver.integrate(y0, ts, solver.init_extra_solver_state(ts[0], y0))
    return ys


def sdeint_adjoint(*a, **k):
    raise NotImplementedError("adjoint: false on the hot path (CFG:41)")
