"""Stand-in for torchdiffeq==0.2.3: imported by models/utils/ode_utils.py:7 for a class the SDE path never uses."""


def odeint(*a, **k):
    raise NotImplementedError("torchdiffeq.odeint is not on the TrajSDE SDE hot path")


odeint_adjoint = odeint
