"""Density network vs an oracle-independent known answer (tests/known_answer.py: exact rational arithmetic from tiny-cuda-nn's
published algorithm, one dense-indexed and one hashed level, 16 ReLU neurons, 3 points).  The CPU test pins oracle/network.py,
the GPU test pins the kernels: exactly in fp32 (all values are dyadic rationals with few bits), within fp16 storage error
(features and weights rounded to 11 bits) in the fp16 mode."""
from fractions import Fraction as Fr

import numpy as np
import pytest
import torch

from tests import known_answer as K


def _answer():
    r = K.evaluate()
    assert r["sigma"] == K.SIGMA_LITERAL                                     # the pasted literals are what the code computes
    assert all(r["grad"][i] == v for i, v in K.GRAD_PROBES_LITERAL.items())
    assert sum(1 for v in r["grad"] if v != 0) == K.N_NONZERO_GRAD
    assert r["hidden_on"] == [9, 9, 9]                                       # both branches of the ReLU are exercised
    f = lambda xs: np.array([float(x) for x in xs], np.float64)
    for x in r["params"] + r["sigma"] + r["grad"]:
        assert Fr(float(np.float32(float(x)))) == x                          # every value is exactly representable in fp32
    return r, f(r["params"]), f(r["sigma"]), f(r["grad"])


def _d_pts(r):
    return np.array([[float(v) for v in row] for row in r["d_pts"]], np.float64)


def test_known_answer_geometry_is_the_published_one():
    lv, n = K.levels()
    assert [(l["res"], l["entries"], l["offset"]) for l in lv] == [(2, 8, 0), (4, 32, 8)] and n == 40
    # level 0 is dense-indexed (x + 2y + 4z), level 1 hashed: (1,1,1) -> 0xAE352E25 % 32 = 5, (3,2,1) -> hand-computed below
    assert K.grid_index(lv[0], [1, 1, 1]) == 7 and K.grid_index(lv[0], [2, 1, 0]) == 4
    assert K.grid_index(lv[1], [1, 1, 1]) == 0xAE352E25 % 32 == 5
    assert K.grid_index(lv[1], [3, 2, 1]) == ((3 ^ (2 * 2654435761) ^ 805459861) & 0xFFFFFFFF) % 32


def test_oracle_network_reproduces_the_rational_known_answer():
    from oracle import network as NW
    r, params, sigma, grad = _answer()
    spec = NW.NetworkSpec.from_config(K.ENC, K.NET)
    assert (spec.n_params, spec.n_mlp_params) == (len(params), r["n_mlp"])
    p = torch.tensor(params, dtype=torch.float32, requires_grad=True)
    pts = torch.tensor([[float(c) for c in pw] for pw in K.POINTS_WORLD], dtype=torch.float32)
    s = NW.density(spec, p, pts)
    assert np.array_equal(s.detach().numpy().astype(np.float64), sigma)       # exact
    (s * torch.tensor([float(d) for d in K.D_SIGMA])).sum().backward()
    assert np.array_equal(p.grad.numpy().astype(np.float64), grad)
    # d sigma / d xyz (the pose-gradient route), inside the cell floor() selects
    x = pts.clone().requires_grad_(True)
    (NW.density(spec, p.detach(), x) * torch.tensor([float(d) for d in K.D_SIGMA])).sum().backward()
    assert np.array_equal(x.grad.numpy().astype(np.float64), _d_pts(r))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_kernels_reproduce_the_rational_known_answer(precision):
    from loner_amd import hip, ops
    r, params, sigma, grad = _answer()
    spec = hip.make_net_spec(K.ENC, dict(K.NET, precision=precision))
    assert (int(spec.n_params), spec.n_mlp_params) == (len(params), r["n_mlp"])
    p = torch.tensor(params, dtype=torch.float32, device="cuda")
    pts = torch.tensor([[float(c) for c in pw] for pw in K.POINTS_WORLD], dtype=torch.float32, device="cuda")
    exact = precision == "fp32"
    close = lambda a, b, tol: np.abs(a - b).max() <= tol * np.abs(b).max()
    s = ops.density_forward(spec, p, pts=pts)
    sg = s.cpu().numpy().astype(np.float64)
    assert np.array_equal(sg, sigma) if exact else close(sg, sigma, 2e-3), (sg, sigma)
    g = torch.zeros_like(p)
    ds = torch.tensor([float(d) for d in K.D_SIGMA], device="cuda")
    d_pts = ops.density_backward(spec, p, ds, g, pts=pts, want_d_pts=True)
    got = g.cpu().numpy().astype(np.float64)
    # the 16-row output matrix: only row 0 exists mathematically; rows 1..15 must not receive a gradient
    h, in_dim = spec.n_neurons, spec.in_dim
    assert not got[h * in_dim + h: r["n_mlp"]].any()
    # weight gradients are exact; table gradients pass through 26-bit records (relative 2^-18) in either mode
    if exact:
        assert np.array_equal(got[:r["n_mlp"]], grad[:r["n_mlp"]])
        assert close(got[r["n_mlp"]:], grad[r["n_mlp"]:], 2.0 ** -18)
        assert set(np.nonzero(got)[0]) == set(np.nonzero(grad)[0])           # exactly the hand-derived entries are touched
    else:
        assert close(got[:r["n_mlp"]], grad[:r["n_mlp"]], 4e-3) and close(got[r["n_mlp"]:], grad[r["n_mlp"]:], 4e-3)
        assert set(np.nonzero(got[r["n_mlp"]:])[0]) == set(np.nonzero(grad[r["n_mlp"]:])[0])
    # same through the rays form (the training loop's route): o + d z with z = 1 and d = 0 offsets
    rays = torch.zeros(3, 13, device="cuda"); rays[:, 0:3] = pts; rays[:, 3:6] = torch.tensor([1.0, 0.0, 0.0], device="cuda")
    z = torch.zeros(3, 64, device="cuda")
    s2 = ops.density_forward(spec, p, rays=rays, z=z)
    assert torch.equal(s2[:, 0], s) and torch.equal(s2[:, 0], s2[:, 63])
    assert close(d_pts.cpu().numpy().astype(np.float64), _d_pts(r), 1e-6 if exact else 1e-2)
