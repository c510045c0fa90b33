"""Philox4x32-10 counter-based normals: the bit-exact host twin of the in-kernel generator
(trajsde_amd/csrc/philox.hpp).  curand-free; the counter is keyed by *global* ids so that the stream
does not change when scenes are re-sharded across GPUs.

element (stream, step, row, col):  key = (seed_lo, seed_hi)
                                   ctr = (row, step, stream, col // 4)
the 4 output words give the normals of columns 4*(col//4) .. +3 through two Box-Muller pairs:
    u = ((x >> 8) + 0.5) * 2^-24 ;  r = sqrt(-2 ln u0) ; n0 = r cos(2 pi u1) ; n1 = r sin(2 pi u1)
The uint32 words are bit-exact between host and device; the float transform agrees to float32
rounding of log/sqrt/sin/cos (about 1e-6 relative), far inside the 1e-4 parity budget.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_FAKE_AGENT = 1   # 2*randn perturbation of the target agents (enc_hivt_nusargo_sde_sep2.py:95)
STREAM_ENCODER = 2      # 21 Brownian increments [Nt,64] (sdeint.py:480)
STREAM_DECODER = 3      # T_euler Brownian increments [K*N,64] (dec_hivt_nusargo_sde.py:88)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr [...,4] uint32, key [2] uint32 -> [...,4] uint32."""
    c0, c1, c2, c3 = (ctr[..., i].astype(np.uint32) for i in range(4))
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * PHILOX_M0
            p1 = c2.astype(np.uint64) * PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + PHILOX_W0)
            k1 = np.uint32(k1 + PHILOX_W1)
    return np.stack([c0, c1, c2, c3], axis=-1)


def _uniform(x: np.ndarray) -> np.ndarray:
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def normals(seed: int, stream: int, step: int, rows: np.ndarray, ncols: int) -> np.ndarray:
    """Standard normals [len(rows), ncols] (ncols % 4 == 0) for the given global row ids."""
    assert ncols % 4 == 0
    rows = np.asarray(rows, dtype=np.uint32)
    nq = ncols // 4
    ctr = np.zeros((rows.shape[0], nq, 4), dtype=np.uint32)
    ctr[..., 0] = rows[:, None]
    ctr[..., 1] = np.uint32(step)
    ctr[..., 2] = np.uint32(stream)
    ctr[..., 3] = np.arange(nq, dtype=np.uint32)[None, :]
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    w = philox4x32_10(ctr, key)
    u = _uniform(w)
    two_pi = np.float32(6.283185307179586)
    out = np.empty((rows.shape[0], nq, 4), dtype=np.float32)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[..., a])).astype(np.float32)
        th = (two_pi * u[..., a + 1]).astype(np.float32)
        out[..., a] = r * np.cos(th).astype(np.float32)
        out[..., a + 1] = r * np.sin(th).astype(np.float32)
    return out.reshape(rows.shape[0], ncols)
