"""Deterministic synthetic cost matrices for the k-best assignment path.

These are the inputs SURVEY.md 8(d) defines for BASELINE.json's configs
(C1..C5).  The reference ships no data (its cost matrices come out of a KITTI
run, system.cpp:271), so every benchmark / parity input is regenerated from a
seed.  Generator: splitmix64, ``u01 = (z >> 11) * 2**-53``, matrices filled in
memory order (column-major ``C[row + col*numRow]`` as the reference expects,
shortestPathCPP.hpp:185-190), problems consecutive in one stream.
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

INF = float("inf")

# name -> (B, numRow, numCol, k, seed) for the dense u01 configs of SURVEY 8(d)
DENSE_CONFIGS = {
    "c1": (1, 8, 8, 10, 12345),
    "c2": (1024, 16, 16, 50, 0x5EED0000 + 16050),
    "c3": (4096, 32, 32, 200, 0x5EED0000 + 32200),
    "c4": (1024, 64, 64, 200, 0x5EED0000 + 64200),
    # not a BASELINE config: the general-size kernel's profile case, same seed rule
    "w128": (512, 128, 128, 200, 0x5EED0000 + 128200),
}


def splitmix64_u01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n consecutive u01 draws of the splitmix64 stream `seed`, skipping `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def dense_batch(B: int, n_row: int, n_col: int, seed: int, first: int = 0) -> np.ndarray:
    """(B, n_row*n_col) float64; problem b is draws [(first+b)*n_row*n_col, ...)."""
    per = n_row * n_col
    return splitmix64_u01(seed, B * per, offset=first * per).reshape(B, per)


def dense_config(name: str, B: int | None = None, first: int = 0):
    """Returns (costs (B, N*M), N, M, k) for one of c1..c4."""
    Bc, N, M, k, seed = DENSE_CONFIGS[name]
    B = Bc if B is None else B
    return dense_batch(B, N, M, seed, first), N, M, k


class _Stream:
    """Sequential view of a splitmix64 stream (for data-dependent draw counts)."""

    def __init__(self, seed: int):
        self.seed = seed
        self.pos = 0
        self._buf = np.empty(0)
        self._base = 0

    def next(self) -> float:
        i = self.pos - self._base
        if i >= self._buf.size:
            self._base = self.pos
            self._buf = splitmix64_u01(self.seed, 1 << 16, offset=self.pos)
            i = 0
        self.pos += 1
        return float(self._buf[i])


def kitti_like_frames(F: int, nL: int = 20, nM: int = 10, seed: int = 0xC0FFEE, gate: float = 10.0):
    """C5: F frames of (nL+nM) x nM column-major cost blocks.

    For c in 0..nM-1, r in 0..nL-1 (one shared stream across frames): draw t;
    if t < 3/nL or r == c draw a, b and set 12*a*b ("plausible"), else draw a
    and set 60 + 400*a; dummy block +inf except C[nL+c, c] = gate
    (gate value: runOpts/calibSample.txt:7; layout: assignment.cpp:705-722).
    Returns a list of 1-D float64 arrays of length (nL+nM)*nM.
    """
    st = _Stream(seed)
    nR = nL + nM
    frames = []
    for _ in range(F):
        C = np.full(nR * nM, INF)
        for c in range(nM):
            for r in range(nL):
                t = st.next()
                if t < 3.0 / nL or r == c:
                    a = st.next()
                    b = st.next()
                    C[c * nR + r] = 12.0 * a * b
                else:
                    a = st.next()
                    C[c * nR + r] = 60.0 + 400.0 * a
            C[c * nR + nL + c] = gate
        frames.append(C)
    return frames
