"""Stand-in for torchsde._core.misc: the helpers sdeint.py touches on the Euler path."""
import warnings

import torch


def handle_unused_kwargs(unused_kwargs, msg=None):
    if len(unused_kwargs) > 0:
        warnings.warn(f"{msg}: Unexpected arguments {unused_kwargs}" if msg else f"Unexpected arguments {unused_kwargs}")


def assert_no_grad(names, maybe_tensors):
    for name, t in zip(names, maybe_tensors):
        if torch.is_tensor(t) and t.requires_grad:
            raise ValueError(f"Argument {name} must not require gradient.")


def is_strictly_increasing(ts):
    return all(x < y for x, y in zip(ts[:-1], ts[1:]))


def batch_mvp(m, v):
    return torch.bmm(m, v.unsqueeze(-1)).squeeze(dim=-1)


def vjp(*a, **k):
    raise NotImplementedError("not on the Euler/diagonal path")


jvp = vjp
