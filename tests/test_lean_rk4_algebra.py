"""The accumulator-free classical RK4 of the fused stage update (kernels.hpp, stage kinds 4-7; fusmi.hip stage_vel)
restated in numpy on a small random second-order system and compared with the reference's form of the same step
(Linear.hpp:273-295: accumulators u_, v_ updated at every stage).  Host-side check of the algebra and of the buffer
rotation the device code uses; the device path itself is compared with the oracle in tests/test_gpu_parity.py."""
import numpy as np

A_RK = [0.0, 0.5, 0.5, 1.0, 0.0]
B_RK = [1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0]
C_RK = [0.0, 0.5, 0.5, 1.0]


def _system(n, seed):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((n, n))
    K = Q @ Q.T / n + np.eye(n)                  # stiffness-like, symmetric positive definite
    minv = 1.0 / rng.uniform(0.5, 2.0, n)        # lumped mass
    absb = np.where(rng.random(n) < 0.3, rng.uniform(0.1, 1.0, n), 0.0)   # absorbing-boundary weights on some dofs
    src = np.where(rng.random(n) < 0.2, rng.uniform(0.1, 1.0, n), 0.0)    # source-boundary weights
    return K, minv, absb, src


def _g(t):
    return np.cos(7.0 * t)


def _kv(K, minv, absb, src, t, u, v):
    """f1 of Linear.hpp:196-221 with the operator as a dense matrix: M^-1 (g(t) src - abs v - K u)."""
    return minv * (_g(t) * src - absb * v - K @ u)


def reference_step(K, minv, absb, src, t, dt, u0, v0):
    """Linear.hpp:273-295: u_ / v_ accumulate b_i dt k_i at every stage."""
    u_, v_ = u0.copy(), v0.copy()
    un, vn = u0.copy(), v0.copy()
    for i in range(4):
        tn = t + C_RK[i] * dt
        ku = vn.copy()
        kv = _kv(K, minv, absb, src, tn, un, vn)
        u_ += B_RK[i] * dt * ku
        v_ += B_RK[i] * dt * kv
        un = u0 + A_RK[i + 1] * dt * ku
        vn = v0 + A_RK[i + 1] * dt * kv
    return u_, v_


def lean_step(K, minv, absb, src, t, dt, u0, v0):
    """Stage kinds 4, 5, 6, 7: no accumulators.  Buffers A, B, C hold the stage velocities V_1, V_2, V_3 (the model's
    vn, v_, u_ rotated per stage by stage_vel); `un` holds the operator input of the next stage; u0 is read at stage 0
    only and rebuilt afterwards from the stage input, u0 = U_i - a_i dt V_{i-1}."""
    n = len(u0)
    A, B, C, un = (np.full(n, np.nan) for _ in range(4))
    rot = [(None, A, None), (A, B, None), (B, C, A), (C, B, A)]     # (vn read, v_ written [stage 3: read], u_ read)
    u0, v0 = u0.copy(), v0.copy()
    for i, (vn, v_, uA) in enumerate(rot):
        tn = t + C_RK[i] * dt
        adt, bdt, pdt = A_RK[i + 1] * dt, B_RK[i] * dt, A_RK[i] * dt
        x = u0 if i == 0 else un                                    # the operator's input (in LDS on the device)
        vstage = v0 if i == 0 else vn
        kv = _kv(K, minv, absb, src, tn, x, vstage)
        if i == 0:
            un[:] = v0 * adt + x
            v_[:] = kv * adt + v0
        elif i == 1:
            un[:] = x + (vn * adt - v0 * pdt)
            v_[:] = kv * adt + v0
        elif i == 2:
            un[:] = x + (vn * adt - uA * pdt)
            v_[:] = kv * adt + v0
        else:
            b0dt = B_RK[0] * dt
            unew = (x - v_ * pdt) + ((uA + v_) * 2.0 + (v0 + vn)) * b0dt
            vnew = kv * bdt + ((v_ * 2.0 + (uA + vn)) - v0) / 3.0
            u0, v0 = unew, vnew
    return u0, v0


def test_accumulator_free_rk4_equals_the_reference_form():
    n, dt = 60, 0.01
    K, minv, absb, src = _system(n, 0)
    rng = np.random.default_rng(1)
    u_ref, v_ref = rng.standard_normal(n), rng.standard_normal(n)
    u_lean, v_lean = u_ref.copy(), v_ref.copy()
    t = 0.3
    for _ in range(50):
        u_ref, v_ref = reference_step(K, minv, absb, src, t, dt, u_ref, v_ref)
        u_lean, v_lean = lean_step(K, minv, absb, src, t, dt, u_lean, v_lean)
        t += dt
    scale = max(np.abs(u_ref).max(), np.abs(v_ref).max())
    assert np.isfinite(scale) and scale > 0.1
    assert np.abs(u_lean - u_ref).max() < 1e-13 * scale
    assert np.abs(v_lean - v_ref).max() < 1e-13 * scale


def test_stream_counts_of_the_two_forms():
    """Values per interior dof and step (bench.py compulsory_bytes): the reference form 7 + 10 + 10 + 6, stage kinds
    4-7 move 4 + 5 + 6 + 7."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    assert sum(bench.LEAN_INTERIOR) == 22 and sum(bench.FULL_INTERIOR) == 33 and sum(bench.LEAN_SHARED) == 26
