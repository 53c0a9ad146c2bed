"""Synthetic data generators shaped like the workloads of the benchmark configs.

No dataset can be downloaded here, so every benchmark and test runs on
synthetic matrices of the named shape:

* ``make_higgs_like``   - 28 dense features (21 "low-level" kinematic-style columns incl.
  four 3-valued discrete tags, 7 "high-level" non-linear combinations), binary label
  with a non-linear decision surface (Higgs-shaped: 10.5 M x 28).
* ``make_diamonds_like`` - 6 features, 3 of them small ordinal integers, log-price style
  target (the diamonds regression of r/gridsearchCV.R:5-30: 53 940 rows).
* ``make_curve_1d``     - the 1-D toy curve of bagging_boosting.ipynb:67-78.
* ``make_regression``   - generic dense regression (1 M x 50 config, 10 k x 10 config).
* ``make_epsilon_like`` - 2000 dense, weakly informative features (Epsilon-shaped).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def make_higgs_like(n: int, seed: int = 0, device: str = "cpu", dtype: str = "float32"):
    """Returns (X [n, 28], y [n]) as torch tensors on ``device`` (or numpy arrays when device == "numpy")."""
    import torch
    dev = "cpu" if device == "numpy" else device
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    tdt = torch.float32 if dtype == "float32" else torch.float64

    def rn(*s):
        return torch.randn(*s, generator=g, device=dev, dtype=torch.float32)

    def ru(*s):
        return torch.rand(*s, generator=g, device=dev, dtype=torch.float32)

    cols = []
    lep_pt = torch.exp(0.5 * rn(n)) * 0.9
    lep_eta = rn(n) * 1.0
    lep_phi = (ru(n) * 2 - 1) * math.pi
    met = -torch.log(ru(n).clamp_min(1e-7)) * 0.8
    met_phi = (ru(n) * 2 - 1) * math.pi
    cols += [lep_pt, lep_eta, lep_phi, met, met_phi]
    jets = []
    for j in range(4):
        pt = torch.exp(0.45 * rn(n)) * (1.1 - 0.15 * j)
        eta = rn(n) * 1.1
        phi = (ru(n) * 2 - 1) * math.pi
        u = ru(n)
        btag = torch.where(u < 0.55, torch.zeros_like(u), torch.where(u < 0.82, torch.full_like(u, 1.0865),
                                                                      torch.full_like(u, 2.1731)))
        jets.append((pt, eta, phi, btag))
        cols += [pt, eta, phi, btag]
    # high-level style combinations
    j0, j1, j2, j3 = jets
    m_jj = torch.sqrt((2 * j0[0] * j1[0] * (torch.cosh(j0[1] - j1[1]) - torch.cos(j0[2] - j1[2]))).clamp_min(0))
    m_jjj = torch.sqrt((m_jj ** 2 + 2 * j2[0] * (j0[0] + j1[0]) * 0.7).clamp_min(0))
    m_lv = torch.sqrt((2 * lep_pt * met * (1 - torch.cos(lep_phi - met_phi))).clamp_min(0))
    m_jlv = m_lv + 0.6 * j0[0] + 0.1 * rn(n)
    m_bb = torch.sqrt((2 * j2[0] * j3[0] * (torch.cosh(j2[1] - j3[1]) - torch.cos(j2[2] - j3[2]))).clamp_min(0))
    m_wbb = m_bb + 0.5 * m_jj + 0.1 * rn(n)
    m_wwbb = m_wbb + 0.5 * m_lv + 0.1 * rn(n)
    cols += [m_jj, m_jjj, m_lv, m_jlv, m_bb, m_wbb, m_wwbb]
    X = torch.stack(cols, dim=1)
    score = (1.3 * torch.tanh(1.2 * (m_bb - 1.0)) - 0.9 * torch.tanh(m_wwbb - 2.2) + 0.6 * (j2[3] > 1).float()
             + 0.5 * (j3[3] > 1).float() - 0.35 * torch.abs(lep_eta) + 0.4 * torch.sin(lep_phi - met_phi)
             + 0.5 * torch.tanh(m_jj - 1.1) * torch.tanh(m_lv - 0.8) - 0.25 * (met - 0.8) + 0.15 * j0[0] * (j0[3] > 0).float())
    p = torch.sigmoid(1.6 * score - 0.1)
    y = (ru(n) < p).float()
    X = X.to(tdt)
    if device == "numpy":
        return X.numpy(), y.numpy()
    return X, y


def make_diamonds_like(n: int = 53940, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(X [n, 6], y): columns log_carat, cut (1-5), color (1-7), clarity (1-8), depth, table; y = log price."""
    rng = np.random.RandomState(seed)
    log_carat = np.clip(rng.normal(-0.39, 0.58, n), -1.7, 1.6)
    log_carat = np.round(np.exp(log_carat), 2)          # carat is recorded to 2 decimals
    log_carat = np.log(np.clip(log_carat, 0.2, 5.01))
    cut = rng.choice(np.arange(1, 6), n, p=[0.03, 0.09, 0.22, 0.26, 0.40]).astype(np.float64)
    color = rng.choice(np.arange(1, 8), n, p=[0.125, 0.18, 0.18, 0.21, 0.155, 0.10, 0.05]).astype(np.float64)
    clarity = rng.choice(np.arange(1, 9), n, p=[0.014, 0.17, 0.24, 0.23, 0.15, 0.094, 0.068, 0.034]).astype(np.float64)
    depth = np.round(rng.normal(61.75, 1.43, n), 1)
    table = np.round(rng.normal(57.46, 2.23, n), 0)
    color_eff = np.array([0.0, -0.05, -0.09, -0.16, -0.25, -0.37, -0.51])[color.astype(int) - 1]
    clar_eff = np.array([0.0, 0.42, 0.59, 0.74, 0.81, 0.94, 1.02, 1.11])[clarity.astype(int) - 1]
    cut_eff = np.array([0.0, 0.08, 0.12, 0.14, 0.16])[cut.astype(int) - 1]
    y = (8.45 + 1.88 * log_carat + color_eff + clar_eff + cut_eff - 0.16 * log_carat ** 2
         - 0.004 * (depth - 61.75) ** 2 - 0.10 * clar_eff * log_carat + 0.35 * color_eff * clar_eff
         + 0.08 * np.tanh(4.0 * (log_carat + 0.35)) + rng.normal(0, 0.092, n))
    X = np.column_stack([log_carat, cut, color, clarity, depth, table])
    return X, y


def make_curve_1d(n: int = 1000, seed: int = 8657, noise: float = 0.05) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Toy curve y = |x| + cos(x): returns (x_grid [1000,1], y_grid, x_train [n,1], y_train)."""
    x = np.linspace(-1, 3, 1000)
    y = np.abs(x) + np.cos(x)
    rng = np.random.RandomState(seed)
    xt = rng.uniform(-1, 3, n)
    yt = np.abs(xt) + np.cos(xt) + rng.uniform(-noise, noise, n)
    return x.reshape(-1, 1), y, xt.reshape(-1, 1), yt


def make_regression(n: int, f: int, seed: int = 0, informative: int | None = None, noise: float = 0.3,
                    dtype=np.float64, coef_seed: int | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """Rows from ``seed``; the linear coefficients from ``coef_seed`` when given (so that shards / train and
    test sets drawn with different seeds follow the same relationship), else from ``seed`` as well."""
    rng = np.random.RandomState(seed)
    X = rng.standard_normal((n, f)).astype(dtype)
    k = informative or max(3, f // 3)
    w = rng.standard_normal(k)
    if coef_seed is not None:
        w = np.random.RandomState(coef_seed).standard_normal(k)
    y = X[:, :k] @ w
    y = y + np.sin(2.0 * X[:, 0]) * 2.0 + X[:, 1] * X[:, 2] + (X[:, min(3, f - 1)] > 0.5) * 1.5
    y = y + noise * rng.standard_normal(n)
    return X, y.astype(np.float64)


def make_epsilon_like(n: int, f: int = 2000, seed: int = 0, device: str = "cpu"):
    """Dense weakly-informative features, binary label (Epsilon-shaped: 400 k x 2000)."""
    import torch
    dev = device
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    X = torch.randn(n, f, generator=g, device=dev, dtype=torch.float32)
    w = torch.randn(f, generator=g, device=dev, dtype=torch.float32) / math.sqrt(f)
    s = X @ w * 3.0 + 0.5 * torch.tanh(X[:, 0] * X[:, 1])
    y = (torch.rand(n, generator=g, device=dev) < torch.sigmoid(s)).float()
    return X, y
