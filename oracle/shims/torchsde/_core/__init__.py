from . import base_sde, base_solver, methods, misc  # noqa: F401
