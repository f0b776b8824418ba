"""Shared helpers for the parity tests (host-side only; no product numerics here)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "cases.json")))
BY_NAME = {c["name"]: c for c in CASES}


def triplets(c):
    t = np.array(c["triplets"])
    return t[:, 0].astype(np.int32), t[:, 1].astype(np.int32), t[:, 2].astype(np.float64)


def rows_of(n, rp):
    return np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))


def read_mtx(path):
    """Minimal coordinate reader used by the tests (the product's reader is russell_amd.sparse.read_matrix_market)."""
    rows, cols, vals, dims, sym = [], [], [], None, False
    with open(path) as fh:
        header = fh.readline().split()
        sym = header[4].lower() == "symmetric"
        for line in fh:
            s = line.strip()
            if not s or s.startswith("%"):
                continue
            a = s.split()
            if dims is None:
                dims = [int(v) for v in a]
                continue
            rows.append(int(a[0]) - 1), cols.append(int(a[1]) - 1), vals.append(float(a[2]))
    return dims, np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals), sym


def relative_error_metric(n, rp, ci, v, x, b):
    """verify_lin_sys.rs:60-96 on a full-storage CSR."""
    r = np.zeros(n)
    np.add.at(r, rows_of(n, rp), v * x[ci])
    return float(np.max(np.abs(r - b)) / (np.max(np.abs(v)) + 1.0))
