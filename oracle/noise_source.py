"""Noise source for the oracle (TEST INFRASTRUCTURE ONLY).

The reference draws randomness in three places:
  1. torch.randn_like at enc_hivt_nusargo_sde_sep2.py:95       -> [A,21,2]
  2. BrownianInterval(t0,t1) inside sdeint_dual (sdeint.py:480) -> 21 x [Nt,64]
  3. BrownianInterval inside stock torchsde.sdeint (dec_hivt_nusargo_sde.py:88) -> T_euler x [K*N,64]
torchsde's real BrownianInterval stream cannot be reproduced without the
library, so parity is defined on *injected* standard normals z: every draw goes
through this module, which either replays a queue of tensors or draws from a
seeded generator, and records what it handed out.
"""
import math
import torch


class NoiseSource:
    def __init__(self):
        self.reset()

    def reset(self, seed=None, replay=None):
        self.gen = torch.Generator().manual_seed(0 if seed is None else int(seed))
        self.replay = list(replay) if replay is not None else None
        self.record = []

    def standard_normal(self, shape, dtype=torch.float32, device="cpu", tag=""):
        shape = tuple(int(s) for s in shape)
        if self.replay is not None:
            if not self.replay:
                raise RuntimeError(f"noise replay queue exhausted at draw '{tag}' {shape}")
            z = self.replay.pop(0)
            z = torch.as_tensor(z, dtype=dtype)
            if tuple(z.shape) != shape:
                raise RuntimeError(f"replayed noise has shape {tuple(z.shape)}, wanted {shape} ({tag})")
        else:
            z = torch.randn(shape, generator=self.gen, dtype=dtype)
        self.record.append((tag, z.clone()))
        return z.to(device)

    def brownian_increment(self, shape, t0, t1, dtype=torch.float32, device="cpu", tag="bm"):
        """W(t1)-W(t0) ~ N(0, t1-t0) as z * sqrt(h).

        h = float(t1) - float(t0) in double (torchsde's BrownianInterval converts its
        query times with float()), the scale is rounded to float32 when it meets z.
        The same number is what trajsde_amd.schedule stores as `sqrt_h`.
        """
        z = self.standard_normal(shape, dtype=dtype, device=device, tag=tag)
        return z * math.sqrt(float(t1) - float(t0))


SOURCE = NoiseSource()
