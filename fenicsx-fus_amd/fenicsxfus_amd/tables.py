"""1-D GLL tables for the host-side mesh helpers (numpy; the C++ library has its own copy
in csrc/tables.cpp for the device path).  Definitions: SURVEY A.2."""
from __future__ import annotations

import numpy as np


def _legendre(n, x):
    p0, p1 = np.ones_like(x), x.copy()
    if n == 0:
        return p0, np.zeros_like(x)
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1, p0


def gll(N: int):
    """N Gauss-Lobatto-Legendre points and weights on [0, 1], increasing; weights sum to 1."""
    n = N - 1
    x = -np.cos(np.pi * np.arange(N) / n)
    for _ in range(100):
        pn, pnm1 = _legendre(n, x)
        dx = (pnm1 - x * pn) / ((n + 1) * pn)
        dx[0] = dx[-1] = 0.0
        x = x + dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x = 0.5 * (x - x[::-1])          # symmetrise
    pn, _ = _legendre(n, x)
    w = 1.0 / (N * n * pn * pn)
    return 0.5 * (x + 1.0), w


def gll_weights_at(nodes):
    nodes = np.asarray(nodes, dtype=np.float64)
    N = len(nodes)
    pn, _ = _legendre(N - 1, 2.0 * nodes - 1.0)
    return 1.0 / (N * (N - 1) * pn * pn)


def dphi(nodes):
    """D[q, i] = phi_i'(nodes[q]) (barycentric form), any node order."""
    x = np.asarray(nodes, dtype=np.float64)
    N = len(x)
    diff = x[:, None] - x[None, :]
    np.fill_diagonal(diff, 1.0)
    lam = 1.0 / diff.prod(axis=1)
    D = (lam[None, :] / lam[:, None]) / diff
    np.fill_diagonal(D, 0.0)
    np.fill_diagonal(D, -D.sum(axis=1))
    return D
