"""Generates tests/golden/sumfact_primitives.npz from the REFERENCE header compiled into
oracle/_ref/libref_sumfact.so (build: `make -C oracle ref`, only where /root/reference exists).
Inputs are seeded; outputs are what the reference's contract/transpose templates
(cpp/fenicsx-sf/common/sum_factorisation.hpp:43-49,70-86) return for them."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402

ref = oracle.ref_lib()
assert ref is not None, "build oracle/_ref first"
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
out = {"Ns": np.arange(2, 9)}
for N in range(2, 9):
    rng = np.random.default_rng(1000 + N)
    A, B = rng.standard_normal(N * N), rng.standard_normal(N**3)
    out[f"A{N}"], out[f"B{N}"] = A, B
    for tr, key in ((1, "Ct"), (0, "Cf")):
        Cc = np.zeros(N**3)
        ref.ref_contract_f64(N, tr, p(A), p(B), p(Cc))
        out[f"{key}{N}"] = Cc
    for pat in (0, 1):
        T = np.zeros(N**3)
        ref.ref_transpose_f64(N, pat, p(B.copy()), p(T))
        out[f"T{pat}{N}"] = T
phi, x = np.arange(6.0), np.arange(8.0)
o, ot = np.zeros(12), np.zeros(12)
ref.ref_contract_mwe_f64(p(phi), p(x), p(o), p(ot))
out["mwe_out"], out["mwe_out_t"] = o, ot
np.savez(os.path.join(HERE, "sumfact_primitives.npz"), **out)
print("wrote sumfact_primitives.npz")
