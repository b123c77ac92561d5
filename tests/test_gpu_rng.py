"""The in-kernel random numbers (counter-based Philox4x32-10, lnr_common.h) against what the reference draws:
torch.rand [N, S/2] twice and torch.randn [N, S] once per forward (ray_sampling.py:72, rendering_tcnn.py:48,104) and
torch.randint per keyframe (optimizer.py:288).  Two questions: are the draws distributed like those (uniform on [0,1) with 24
bits, N(0,1), independent along a ray, across rays, across streams and across seeds), and are they really what the kernels use
(a seeded call must equal the same call fed with the dumped draws, bit for bit)."""
import numpy as np
import pytest
import torch
from scipy import stats

pytestmark = pytest.mark.gpu

DEV = "cuda"
N_RAYS, N_PER = 4096, 256          # the default training shape: 4096 rays x 256 jitter / pdf draws (512 noise draws)


def _corr(a, b):
    a = a.double().flatten() - a.double().mean()
    b = b.double().flatten() - b.double().mean()
    return float((a * b).sum() / torch.sqrt((a * a).sum() * (b * b).sum()))


@pytest.mark.parametrize("which", ["jitter", "pdf", "ray_index0", "ray_index3"])
def test_uniform_draws_are_uniform_and_independent(which):
    from loner_amd import hip, ops
    code = {"jitter": hip.DRAW_JITTER, "pdf": hip.DRAW_PDF, "ray_index0": hip.DRAW_RAY_INDEX, "ray_index3": hip.DRAW_RAY_INDEX + 3}[which]
    u = ops.rng_draws(code, 0x1234567, N_RAYS, N_PER)
    n = u.numel()
    x = u.cpu().double().numpy().ravel()
    # torch's float32 uniform: 24 random bits / 2^24, never 1.0
    assert x.min() >= 0.0 and x.max() < 1.0 and np.all(x * 2 ** 24 == np.round(x * 2 ** 24))
    # first four moments of U(0,1): mean 1/2, variance 1/12, skewness 0, excess kurtosis -6/5; each within 4.5 standard errors
    assert abs(x.mean() - 0.5) < 4.5 * np.sqrt(1 / 12 / n)
    assert abs(x.var() - 1 / 12) < 4.5 * np.sqrt(1 / 180 / n)
    assert abs(stats.skew(x)) < 4.5 * np.sqrt(6 / n) and abs(stats.kurtosis(x) + 1.2) < 4.5 * np.sqrt(24 / n)
    # Kolmogorov-Smirnov against U(0,1) and a chi-square over 1024 equal bins
    assert stats.kstest(x, "uniform").pvalue > 1e-4
    counts = np.bincount(np.minimum((x * 1024).astype(np.int64), 1023), minlength=1024)
    assert stats.chisquare(counts).pvalue > 1e-4
    # independence: along a ray (lags 1, 2, 4: elements of one Philox call and of neighbouring calls), across rays, and of the low bits
    lim = 4.5 / np.sqrt(n)
    for lag in (1, 2, 3, 4, 5):
        assert abs(_corr(u[:, :-lag], u[:, lag:])) < lim, lag
    assert abs(_corr(u[:-1], u[1:])) < lim and abs(_corr(u[:-7], u[7:])) < lim
    low = torch.from_numpy((x * 2 ** 24).astype(np.int64) & 0xFF).double().reshape(N_RAYS, N_PER)
    assert abs(_corr(low[:, :-1], low[:, 1:])) < lim
    assert stats.chisquare(np.bincount(low.long().numpy().ravel(), minlength=256)).pvalue > 1e-4
    # other seeds, other streams: unrelated
    assert abs(_corr(u, ops.rng_draws(code, 0x1234568, N_RAYS, N_PER))) < lim
    other = hip.DRAW_PDF if code != hip.DRAW_PDF else hip.DRAW_JITTER
    assert abs(_corr(u, ops.rng_draws(other, 0x1234567, N_RAYS, N_PER))) < lim


def test_noise_draws_are_standard_normal_and_independent():
    from loner_amd import hip, ops
    S = 2 * N_PER
    g = ops.rng_draws(hip.DRAW_NOISE, 99, N_RAYS, S)
    n = g.numel()
    x = g.cpu().double().numpy().ravel()
    assert np.isfinite(x).all() and np.abs(x).max() < 5.8               # Box-Muller on 24-bit uniforms: |x| <= sqrt(2 ln 2^24) = 5.77
    assert abs(x.mean()) < 4.5 / np.sqrt(n) and abs(x.var() - 1.0) < 4.5 * np.sqrt(2 / n)
    assert abs(stats.skew(x)) < 4.5 * np.sqrt(6 / n) and abs(stats.kurtosis(x)) < 4.5 * np.sqrt(24 / n)
    assert stats.kstest(x, "norm").pvalue > 1e-4
    # tail mass where the loss is sensitive (raw2outputs adds the noise in front of a ReLU, rendering_tcnn.py:102-112)
    for t in (1.0, 2.0, 3.0):
        p = 2 * stats.norm.sf(t)
        assert abs((np.abs(x) > t).mean() - p) < 4.5 * np.sqrt(p * (1 - p) / n), t
    assert abs((x > 0).mean() - 0.5) < 4.5 * np.sqrt(0.25 / n)
    lim = 4.5 / np.sqrt(n)
    for lag in (1, 2, 3):                                                # elements 2k, 2k+1 come from ONE Philox call
        assert abs(_corr(g[:, :-lag], g[:, lag:])) < lim, lag
        assert abs(_corr(g[:, :-lag].abs(), g[:, lag:].abs())) < lim, lag
    assert abs(_corr(g[:-1], g[1:])) < lim
    assert abs(_corr(g, ops.rng_draws(hip.DRAW_NOISE, 100, N_RAYS, S))) < lim
    # unrelated to the sampler's uniforms of the same seed (the loop uses seed and seed + 1, optimizer._loss_and_grads)
    assert abs(_corr(g[:, :N_PER], ops.rng_draws(hip.DRAW_JITTER, 99, N_RAYS, N_PER))) < lim
    assert abs(_corr(g[:, :N_PER], ops.rng_draws(hip.DRAW_JITTER, 98, N_RAYS, N_PER))) < lim


def _rays(n, seed=0):
    gen = torch.Generator().manual_seed(seed)
    rays = torch.zeros(n, 13)
    rays[:, 0:3] = torch.rand(n, 3, generator=gen) * 0.2 - 0.1
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1)
    rays[:, 6:9] = -rays[:, 3:6]
    rays[:, 11] = 0.0117
    rays[:, 12] = 0.3 + 0.28 * torch.rand(n, generator=gen)
    return rays.to(DEV)


def test_seeded_kernels_use_exactly_these_draws():
    """lnr_rng_draws is only evidence if the kernels use the same numbers: every seeded call must equal, bit for bit, the same call
    with the dumped draws handed in (the explicit-draw paths are the ones the golden fixtures pin against the reference)."""
    from loner_amd import hip, ops
    n, S, seed = 300, 128, 0xABCDEF0123
    rays = _rays(n)
    gen = torch.Generator().manual_seed(4)
    grid = (torch.randn(32, 32, 32, generator=gen) * 2).to(DEV)
    u1, u2 = ops.rng_draws(hip.DRAW_JITTER, seed, n, S // 2), ops.rng_draws(hip.DRAW_PDF, seed, n, S // 2)
    z_seed = ops.sample_rays_occ(rays, grid, S, 1.0, seed=seed)
    z_draw = ops.sample_rays_occ(rays, grid, S, 1.0, u_jitter=u1, u_pdf=u2)
    assert torch.equal(z_seed, z_draw)
    assert not torch.equal(z_seed, ops.sample_rays_occ(rays, grid, S, 1.0, seed=seed + 1))
    uj = ops.rng_draws(hip.DRAW_JITTER, seed, n, S)
    assert torch.equal(ops.sample_rays_uniform(rays, S, 1.0, seed=seed), ops.sample_rays_uniform(rays, S, 1.0, u_jitter=uj))
    # density noise: lnr_render_forward and the fused loss add noise_std * N(0,1) to sigma (noise_std 1.0: default_model_config.yaml:18)
    sigma = (torch.randn(n, S, generator=gen) * 3).to(DEV)
    noise = ops.rng_draws(hip.DRAW_NOISE, seed, n, S)
    a = ops.render_forward(sigma, z_seed, rays, noise_std=1.0, seed=seed)
    b = ops.render_forward(sigma, z_seed, rays, noise=noise, noise_std=1.0)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())
    depths = torch.full((n,), 0.2, device=DEV)
    cfg = hip.LossConfig(selection=0, min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0, depth_lambda=0.005, min_eps=0.5, fixed_eps=3.0)
    counts = ops.count_opaque(rays, depths)
    la = ops.los_loss_fused(sigma, z_seed, rays, depths, 85.0, cfg, counts, noise_std=1.0, seed=seed)
    lb = ops.los_loss_fused(sigma, z_seed, rays, depths, 85.0, cfg, counts, noise=noise, noise_std=1.0)
    assert torch.equal(la[0], lb[0]) and torch.equal(la[1], lb[1]) and torch.equal(la[2], lb[2])
    # 2048-sample rays take the staged-row path of the compositing kernels (32 samples per lane)
    S2 = 2048
    z2 = torch.sort(torch.rand(64, S2, generator=gen) * 0.25 + 0.012, dim=1).values.to(DEV)
    sg2 = (torch.randn(64, S2, generator=gen) * 3).to(DEV)
    n2 = ops.rng_draws(hip.DRAW_NOISE, seed, 64, S2)
    for x, y in zip(ops.render_forward(sg2, z2, rays[:64], noise_std=1.0, seed=seed), ops.render_forward(sg2, z2, rays[:64], noise=n2, noise_std=1.0)):
        assert torch.equal(x, y)


def test_window_ray_indices_are_uniform_over_the_scan():
    """optimizer.py:288: torch.randint(len(scan), (512,)) per keyframe; here index = floor(u * n) with the draws of stream
    LNR_DRAW_RAY_INDEX + segment: the indices the kernel reports are exactly those, and they cover the scan evenly."""
    from loner_amd import hip, ops
    n_pts, n_seg, per = 65536, 3, 4096
    gen = torch.Generator().manual_seed(1)
    dirs = [torch.nn.functional.normalize(torch.randn(3, n_pts, generator=gen), dim=0).to(DEV) for _ in range(n_seg)]
    dist = [(torch.rand(n_pts, generator=gen) * 20 + 2).to(DEV) for _ in range(n_seg)]
    tab = ops.WindowTables(dirs, dist, [0.0] * n_seg, [per] * n_seg, list(range(n_seg)))
    T = torch.eye(4)[:3, :4].reshape(1, 12).repeat(n_seg, 1).to(DEV)
    seed = 777
    _, _, _, idx = ops.build_window_rays(tab, T, [1.0, 50.0], 85.0, [0.0, 0.0, 0.0], seed=seed)
    idx = idx.reshape(n_seg, per)
    for s in range(n_seg):
        u = ops.rng_draws(hip.DRAW_RAY_INDEX + s, seed, 1, per).reshape(-1)
        want = torch.clamp((u * float(n_pts)).long(), max=n_pts - 1)
        assert torch.equal(idx[s], want)
    # evenness over many draws: chi-square over 64 beams (index // 1024), and no index out of range
    big = torch.clamp((ops.rng_draws(hip.DRAW_RAY_INDEX, 5, 512, 2048).reshape(-1) * float(n_pts)).long(), max=n_pts - 1).cpu().numpy()
    assert big.min() >= 0 and big.max() < n_pts
    assert stats.chisquare(np.bincount(big // 1024, minlength=64)).pvalue > 1e-4
    assert stats.chisquare(np.bincount(big % 1024, minlength=1024)).pvalue > 1e-4
