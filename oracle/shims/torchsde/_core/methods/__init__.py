from . import euler  # noqa: F401
from .euler import Euler


def select(method, sde_type):
    if method != "euler":
        raise NotImplementedError("stand-in implements the Euler-Maruyama method only (CFG:44,76 method: euler)")
    return Euler
