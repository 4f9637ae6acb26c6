"""Pins the oracle's contraction/transposition primitives (CPU only).

(1) against the reference's own iota known-answer demo (cpp/mwe/sum_factorisation/main.cpp:42-55,
    main.py:9-12) and numpy.tensordot; (2) against the reference header compiled here into
    oracle/_ref (skipped where the reference tree was never available); (3) against the committed
    golden fixture generated from that compiled reference (tests/golden/make_primitives_golden.py).
"""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sumfact_primitives.npz")


def test_mwe_iota_kat(orc):
    # the reference demo: M=3, N=2, x=iota(8), dphi=iota(6): out = tensordot(phi, x, [1],[0])
    M, N = 3, 2
    x = np.arange(N**3, dtype=np.float64)
    phi = np.arange(M * N, dtype=np.float64)
    out = orc.contract(phi, x, (N, M, N, N), True)
    ref = np.tensordot(phi.reshape(M, N), x.reshape(N, N, N), axes=[1, 0])
    assert np.array_equal(out.reshape(M, N, N), ref)
    out_t = orc.transpose3(out, (M, N, N), (N, 1, M * N))
    # B[N*a + b + M*N*c] = A[a,b,c]  <=>  B viewed (c,a,b)
    assert np.array_equal(out_t.reshape(N, M, N), np.transpose(ref, (2, 0, 1)))


def test_survey_kat(orc):
    # contract<double,3,3,3,3,true>(iota(9)+1, iota(27)) -> out[0]=72, out[1]=78, out[26]=426
    out = orc.contract(np.arange(9.0) + 1, np.arange(27.0), (3, 3, 3, 3), True)
    assert (out[0], out[1], out[26]) == (72.0, 78.0, 426.0)


@pytest.mark.parametrize("N", range(2, 9))
@pytest.mark.parametrize("tr", [True, False])
def test_contract_vs_tensordot(orc, N, tr):
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N))
    B = rng.standard_normal((N, N, N))
    out = orc.contract(A, B, (N, N, N, N), tr).reshape(N, N, N)
    ref = np.tensordot(A if tr else A.T, B, axes=[1, 0])
    assert np.allclose(out, ref, rtol=0, atol=1e-13)


@pytest.mark.parametrize("N", range(2, 9))
def test_against_compiled_reference(orc, N):
    ref = orc.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(100 + N)
    A = rng.standard_normal(N * N)
    B = rng.standard_normal(N**3)
    for tr in (1, 0):
        Cr = np.zeros(N**3)
        assert ref.ref_contract_f64(N, tr, A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p),
                                    Cr.ctypes.data_as(C.c_void_p)) == 0
        Co = orc.contract(A, B, (N, N, N, N), bool(tr))
        assert np.array_equal(Co, Cr)          # same k-outer summation order: bit exact
    for pat, offs in ((0, (N, N * N, 1)), (1, (1, N, N * N))):
        Br = np.zeros(N**3)
        Bc = B.copy()
        ref.ref_transpose_f64(N, pat, Bc.ctypes.data_as(C.c_void_p), Br.ctypes.data_as(C.c_void_p))
        assert np.array_equal(orc.transpose3(B, (N, N, N), offs), Br)


def test_against_golden_fixture(orc):
    g = np.load(GOLD)
    for N in g["Ns"]:
        A, B = g[f"A{N}"], g[f"B{N}"]
        assert np.array_equal(orc.contract(A, B, (N, N, N, N), True), g[f"Ct{N}"])
        assert np.array_equal(orc.contract(A, B, (N, N, N, N), False), g[f"Cf{N}"])
        assert np.array_equal(orc.transpose3(B, (N, N, N), (N, N * N, 1)), g[f"T0{N}"])
        assert np.array_equal(orc.transpose3(B, (N, N, N), (1, N, N * N)), g[f"T1{N}"])
    assert np.array_equal(g["mwe_out"], np.tensordot(np.arange(6.0).reshape(3, 2),
                                                      np.arange(8.0).reshape(2, 2, 2), axes=[1, 0]).ravel())
