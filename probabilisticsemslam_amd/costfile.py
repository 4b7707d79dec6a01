"""The reference's on-disk cost-matrix format (`generatedData/<seq>/costMatrices/<ID>_frame<N>.dat`).

Written by saveAssignmentProb (assignment.cpp:799-833): one text line per ROW of the (nL+nM) x nM matrix,
comma-separated `std::to_string(double)` values (fixed notation, 6 decimals, "inf" for +infinity).
Read by getCosts (comparison.cpp:32-57): split on ',', a token starting with 'i' is +infinity, anything else
`std::stod`; the harness then unrolls the rows into the column-major vector the solver takes
(comparison.cpp:151-156).
"""
from __future__ import annotations

import numpy as np


def write_cost_matrix(path: str, cost_colmajor, n_rows: int, n_cols: int) -> None:
    """saveAssignmentProb's writer (assignment.cpp:821-831)."""
    c = np.asarray(cost_colmajor, dtype=np.float64).reshape(n_cols, n_rows)
    with open(path, "w") as fh:
        for r in range(n_rows):
            fh.write(",".join("inf" if np.isposinf(x) else ("-inf" if np.isneginf(x) else "%f" % x) for x in c[:, r]))
            fh.write("\n")


def read_cost_matrix(path: str):
    """getCosts + the unroll of comparison.cpp:151-156.  Returns (cost column-major 1-D, nL, nM)."""
    rows = []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            rows.append([np.inf if tok[:1] == "i" else float(tok) for tok in line.split(",")])
    a = np.asarray(rows, dtype=np.float64)        # [nRows][nCols]
    n_rows, n_cols = a.shape
    return np.ascontiguousarray(a.T).reshape(-1), n_rows - n_cols, n_cols
