"""Generate tests/golden/costfile_golden.npz from the reference's OWN cost-file writer and reader.

Run in the build container only (needs /root/reference):

    make -C oracle && python tests/golden/gen_costfile_golden.py

It loads oracle/_ref/libref_costfile.so -- verbatim line ranges of /root/reference (constsUtils.h:10,13-16,
comparison.cpp:32-57 getCosts, assignment.cpp:821-831 the writing loop of saveAssignmentProb; see
oracle/ref_costfile_shim.cpp and oracle/Makefile) -- and records, for seeded cost blocks, the bytes the reference writes
and the values its reader returns for them.  The fixture holds data only (inputs, file bytes, parsed values)."""
from __future__ import annotations

import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from probabilisticsemslam_amd import workloads as wl  # noqa: E402


def ref_lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_costfile.so"))
    lib.ref_get_costs.restype = C.c_int64
    lib.ref_get_costs.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.ref_write_costs.restype = None
    lib.ref_write_costs.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_char_p]
    return lib


def ref_write(lib, cost_colmajor, n_rows, n_cols, path):
    c = np.ascontiguousarray(cost_colmajor, dtype=np.float64)
    lib.ref_write_costs(c.ctypes.data, n_rows, n_cols, path.encode())


def ref_read(lib, root, ident, frame, cap=1 << 20):
    out = np.empty(cap, np.float64)
    nr, nc = C.c_int64(0), C.c_int64(0)
    n = lib.ref_get_costs(root.encode(), ident.encode(), frame, out.ctypes.data, cap, C.byref(nr), C.byref(nc))
    if n < 0:
        return None
    return out[:n].reshape(nr.value, nc.value).copy()


def cases():
    """(name, cost (nRows x nCols) column-major, nRows, nCols)"""
    for i, f in enumerate(wl.kitti_like_frames(3)):
        yield f"c5_f{i}", f, 30, 10
    for i, f in enumerate(wl.kitti_like_frames(2, nL=6, nM=3)):
        yield f"small_f{i}", f, 9, 3
    rng = np.random.default_rng(31)
    x = rng.random(7 * 4) * 1e-3            # tiny values: 6 decimals lose most digits
    yield "tiny", x, 7, 4
    x = rng.random(5 * 5) * 1e7             # large values
    x[3] = np.inf
    yield "large", x, 5, 5
    x = rng.random(4 * 2) - 0.5             # negative values
    yield "negative", x, 4, 2
    yield "one", np.array([0.1234565]), 1, 1   # a tie of the 6-decimal rounding


def main():
    lib = ref_lib()
    out = {"names": []}
    with tempfile.TemporaryDirectory() as root:
        d = os.path.join(root, "generatedData", "00", "costMatrices")
        os.makedirs(d)
        for i, (name, cost, nr, nc) in enumerate(cases()):
            path = os.path.join(d, f"gold_frame{i}.dat")
            ref_write(lib, cost, nr, nc, path)
            text = open(path, "rb").read()
            rows = ref_read(lib, root, "gold", i)
            assert rows is not None and rows.shape == (nr, nc), name
            out["names"].append(name)
            out[name + "/cost"] = np.asarray(cost, np.float64)
            out[name + "/shape"] = np.array([nr, nc], np.int64)
            out[name + "/text"] = np.frombuffer(text, np.uint8)
            out[name + "/rows"] = rows          # what getCosts returns: [row][col]
        assert ref_read(lib, root, "gold", 999) is None  # a missing file: getCosts returns false
    out["names"] = np.array(out["names"])
    np.savez_compressed(os.path.join(HERE, "costfile_golden.npz"), **out)
    print("wrote costfile_golden.npz:", len(out["names"]), "cases")


if __name__ == "__main__":
    main()
