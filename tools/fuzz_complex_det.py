#!/usr/bin/env python3
"""Differential fuzzing of the complex twin's determinant and solutions (round 4: paired pivot searches, interface_complex_hipmf.cpp)
against numpy's dense complex LU: random complex matrices of several kinds (dominant diagonal of any phase, weak diagonal with a hidden
permutation -> matching on the moduli, complex symmetric in lower storage, band, disconnected blocks, natural order) at sizes up to
~500 through the C-ABI; stops at the first mismatch and prints the seed.  usage: fuzz_complex_det.py [CASES [SEED0 [LIB]]]"""
import ctypes as C, os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd._capi import load

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
lib = load(sys.argv[3] if len(sys.argv) > 3 else None)
sizes = [int(s) for s in os.environ.get("FUZZ_SIZES", "1,2,3,5,16,17,32,33,64,65,100,129,200,333,500").split(",")]


def make(rng):
    n = int(rng.choice(sizes))
    kind = str(rng.choice(["dominant", "weak", "symlower", "band", "blocks", "natural"]))
    k = int(rng.uniform(1.0, 5.0) * n) + 1
    cz = lambda m: rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)
    A = sp.coo_matrix((cz(k), (rng.integers(0, n, k), rng.integers(0, n, k))), shape=(n, n)).tolil()
    if kind == "band" and n > 1:
        offs = [o for o in (1, 2, 7) if o < n]
        A = (sp.diags([cz(n - o) for o in offs], offs, shape=(n, n)) + sp.diags([cz(n - o) for o in offs], [-o for o in offs], shape=(n, n))).tolil()
    if kind == "blocks" and n > 3:
        h = n // 3
        A[:h, h:] = 0.0
        A[h:, :h] = 0.0
    if kind == "symlower":
        A = (A + A.T).tolil()
    Ac = A.tocsr()
    rowsum = np.asarray(abs(Ac).sum(axis=1)).ravel() + np.asarray(abs(Ac).sum(axis=0)).ravel()
    phase = np.exp(2j * np.pi * rng.random(n))
    phase[rng.random(n) < 0.2] = 1j  # purely imaginary diagonal entries: the real diagonal of the real-equivalent form is zero there
    if kind == "weak":
        A.setdiag(0.0)
        perm = rng.permutation(n)
        for i in range(n):
            A[i, perm[i]] = (rowsum[i] + 1.0 + rng.random()) * phase[i]
    else:
        A.setdiag((rowsum * rng.uniform(0.3, 1.0, n) + 0.5) * phase)
        if kind == "symlower":
            A = sp.tril(A).tolil()
    A = sp.csr_matrix(A)
    A.sort_indices()
    return n, kind, A


worst = 0.0
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    n, kind, S = make(rng)
    full = (S + sp.tril(S, -1).T).tocsr() if kind == "symlower" else S
    dense = full.toarray()
    rp, ci = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    zv = np.ascontiguousarray(np.stack([S.data.real, S.data.imag], axis=1).ravel())
    h = lib.complex_solver_hipmf_new()
    ordering = 2 if kind == "natural" else 0
    code = lib.complex_solver_hipmf_initialize(h, ordering, 1, -1.0, -1, 0, int(kind == "symlower"), n, rp, ci, zv.ctypes.data if rng.random() < 0.7 else None)
    assert code == 0, (seed0 + c, "initialize", code)
    npert, dre, dim, dex = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
    code = lib.complex_solver_hipmf_factorize(h, None, None, C.byref(npert), None, C.byref(dre), C.byref(dim), C.byref(dex), 1, 0, zv)
    assert code == 0, (seed0 + c, "factorize", code)
    xs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = full @ xs
    x = np.zeros(2 * n)
    assert lib.complex_solver_hipmf_solve(h, x, np.ascontiguousarray(np.stack([b.real, b.imag], axis=1).ravel()), 0) == 0
    lib.complex_solver_hipmf_drop(h)
    cond = np.linalg.cond(dense)
    sign, logabs = np.linalg.slogdet(dense)
    m, e = complex(dre.value, dim.value), dex.value
    tol = 1e-13 * max(cond, 10.0) * n
    err_x = np.max(np.abs(x[0::2] + 1j * x[1::2] - xs)) / np.max(np.abs(xs))
    err_mod = abs(np.log10(abs(m)) + e - logabs / np.log(10.0)) if m != 0 else np.inf
    err_ph = abs(m / abs(m) - sign) if m != 0 else np.inf
    bad = not (1.0 <= abs(m) < 10.0) or err_mod > tol or err_ph > 10.0 * tol or err_x > max(tol, 1e-10) or npert.value != 0
    worst = max(worst, err_mod / tol, err_ph / (10.0 * tol))
    if bad:
        print("MISMATCH seed %d n %d kind %s: cond %.2e perturbed %d | log10|det| error %.2e, phase error %.2e, solution error %.2e (tol %.1e)" %
              (seed0 + c, n, kind, cond, npert.value, err_mod, err_ph, err_x, tol))
        sys.exit(1)
print("%d complex cases ok (seeds %d..%d): determinants (modulus and phase) and solutions against numpy, worst determinant error / tolerance = %.2e" %
      (cases, seed0, seed0 + cases - 1, worst))
