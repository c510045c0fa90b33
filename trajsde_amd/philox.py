"""Philox4x32-7 counter-based normals: the bit-exact host twin of the in-kernel generator
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


PHILOX_ROUNDS = 7      # csrc/philox.hpp: Random123's Philox4x32-7 (the Crush-resistant minimum; 10 is its default with a margin)


def philox4x32(ctr: np.ndarray, key: np.ndarray, rounds: int = PHILOX_ROUNDS) -> np.ndarray:
    """ctr [...,4] uint32, key [2] uint32 -> [...,4] uint32."""
    c0, c1, c2, c3 = (ctr[..., i].astype(np.uint32) for i in range(4))
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
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
    w = philox4x32(ctr, key)
    u = _uniform(w)
    two_pi = np.float32(6.283185307179586)
    out = np.empty((rows.shape[0], nq, 4), dtype=np.float32)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[..., a])).astype(np.float32)
        th = (two_pi * u[..., a + 1]).astype(np.float32)
        out[..., a] = r * np.cos(th).astype(np.float32)
        out[..., a + 1] = r * np.sin(th).astype(np.float32)
    return out.reshape(rows.shape[0], ncols)


# ---------------------------------------------------------------------------------------------- dropout masks
# Host twin of trajsde_amd/csrc/dropout.hpp (bit-exact: integer comparisons on the Philox words).  A mask element is a 16-bit
# field of a Philox block, kept when field >= round(p * 65536), then scaled by 1 / (1 - p).
STREAM_DROPOUT = 16
DK_ATTN, DK_PROJ, DK_HIDDEN, DK_OUT = 0, 1, 2, 3
BLOCK_AA, BLOCK_AL, BLOCK_GLOBAL0 = 0, 1, 2            # attention block ids: AAEncoder, ALEncoder, global layer i -> 2 + i


def words(seed: int, stream: int, step, rows, quad: int = 0) -> np.ndarray:
    """raw Philox words [n, 4] of the counters (row, step, stream, quad); `step` and `rows` broadcast"""
    rows = np.asarray(rows, dtype=np.uint32)
    step = np.broadcast_to(np.asarray(step, dtype=np.uint32), rows.shape)
    ctr = np.zeros(rows.shape + (4,), dtype=np.uint32)
    ctr[..., 0], ctr[..., 1], ctr[..., 2], ctr[..., 3] = rows, step, np.uint32(stream), np.uint32(quad)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32(ctr, key)


def dropout_threshold(p: float) -> int:
    return int(float(p) * 65536.0 + 0.5)


def dropout_scale(p: float) -> np.float32:
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))


def _fields(w: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """16-bit field `idx` (0..7) of each 4-word block w[..., 4]"""
    word = np.take_along_axis(w, (idx >> 1)[..., None].astype(np.int64), axis=-1)[..., 0]
    return (word >> (np.uint32(16) * (idx & 1).astype(np.uint32))) & np.uint32(0xFFFF)


def dropout_feature_mask(seed: int, block: int, kind: int, n_rows: int, n_feat: int, p: float) -> np.ndarray:
    """[n_rows, n_feat] float32 of {0, 1/(1-p)}: feature f = 64 blk + 16 jt + 4 g + c of row r comes from the block of counter
    (r, 8 blk + 2 g + (jt >> 1), stream, 0), field 4 (jt & 1) + c"""
    f = np.arange(n_feat)
    blk, within = f >> 6, f & 63
    jt, g, c = within >> 4, (within >> 2) & 3, within & 3
    call, idx = blk * 8 + 2 * g + (jt >> 1), 4 * (jt & 1) + c
    rows = np.repeat(np.arange(n_rows, dtype=np.uint32)[:, None], n_feat, axis=1)
    w = words(seed, STREAM_DROPOUT + 4 * block + kind, np.broadcast_to(call[None, :], rows.shape), rows)
    keep = _fields(w, np.broadcast_to(idx[None, :], rows.shape)) >= np.uint32(dropout_threshold(p))
    return np.where(keep, dropout_scale(p), np.float32(0.0)).astype(np.float32)


def dropout_attn_mask(seed: int, block: int, dst, rank, heads: int, p: float) -> np.ndarray:
    """[E, heads] float32 of {0, 1/(1-p)} for edges given by their target node and their rank inside the target's segment of the
    canonical (ascending sender) order: counter (dst, rank, stream, 0), field = head"""
    dst = np.asarray(dst, dtype=np.uint32)
    w = words(seed, STREAM_DROPOUT + 4 * block + DK_ATTN, np.asarray(rank, dtype=np.uint32), dst)
    h = np.arange(heads)
    wrep = np.repeat(w[:, None, :], heads, axis=1)
    keep = _fields(wrep, np.broadcast_to(h[None, :], (dst.shape[0], heads))) >= np.uint32(dropout_threshold(p))
    return np.where(keep, dropout_scale(p), np.float32(0.0)).astype(np.float32)


TEMPORAL_BLOCK0 = 16         # dropout block ids of the vanilla variant's TemporalEncoder layers: 16 + layer (csrc/dropout.hpp)


def dropout_temporal_attn_mask(seed: int, block: int, n_actors: int, heads: int, p: float, tokens: int = 22) -> np.ndarray:
    """[n_actors, heads, tokens, tokens] float32 of {0, 1/(1-p)}: the factor of attention weight (query i, key j) of head h of actor n in
    a TemporalEncoder layer (nn.MultiheadAttention's dropout on the softmax output, GENC:262) comes from the block of counter
    (n, (i * heads + h) * 3 + (j >> 3), stream, 0), field j & 7 -- csrc/dropout.hpp drop_tr_attn"""
    n, h, i, j = np.meshgrid(np.arange(n_actors, dtype=np.uint32), np.arange(heads, dtype=np.uint32), np.arange(tokens, dtype=np.uint32),
                             np.arange(tokens, dtype=np.uint32), indexing="ij")
    call = (i * np.uint32(heads) + h) * np.uint32(3) + (j >> np.uint32(3))
    w = words(seed, STREAM_DROPOUT + 4 * block + DK_ATTN, call.reshape(-1), n.reshape(-1))
    keep = _fields(w, (j & np.uint32(7)).reshape(-1)) >= np.uint32(dropout_threshold(p))
    return np.where(keep, dropout_scale(p), np.float32(0.0)).astype(np.float32).reshape(n_actors, heads, tokens, tokens)


def segment_ranks(src, dst) -> np.ndarray:
    """rank of every edge inside its target's segment when segments are ordered by ascending sender (ties: input order) --
    the order of the compacted lists of csrc/prep.hip"""
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    n = src.shape[0]
    order = np.lexsort((np.arange(n), src, dst))
    ds = dst[order]
    first = np.r_[0, np.flatnonzero(ds[1:] != ds[:-1]) + 1] if n else np.zeros(0, dtype=np.int64)
    start = np.repeat(first, np.diff(np.r_[first, n])) if n else first
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n) - start
    return rank
