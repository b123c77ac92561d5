"""HIP kernels vs the CPU oracle and the golden fixtures - needs an MI355X (pytest -m gpu).

Every test calls the product through the C ABI (loner_amd.ops -> ctypes -> libloner_hip.so) and uses
oracle/ only as the checker.  Integer / rounding-defined stages are asserted bit-exact; floating point
stages to the tolerance written next to each assert (north_star: 1e-4 relative on depth outputs).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss as OL
from oracle import mapping_step as MS
from oracle import network as NW
from oracle import occupancy as OC
from oracle import poses as OP
from oracle import rays as OR
from oracle import render as ORD
from oracle import sampling as SP

DEV = "cuda"


def dv(x, dtype=torch.float32):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(DEV, dtype).contiguous()


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_layers(spec_o, grad, ref):
    """The largest per-MATRIX relative error of an MLP gradient (each weight matrix against its own largest entry): `rel` over the whole
    vector is dominated by the matrix with the largest gradient - a first-layer gradient that was wrong in every entry passed it at
    3e-3 (round 5, fp16 mode of the 256 x n route) because the output row's entries were 1000 x larger."""
    worst, off = 0.0, 0
    for rows, cols in spec_o.mlp_shapes:
        n = rows * cols
        if float(ref[off:off + n].abs().max()) > 0.0:
            worst = max(worst, rel(grad[off:off + n], ref[off:off + n]))
        off += n
    return worst


@pytest.fixture(scope="module")
def ops():
    from loner_amd import ops as _ops
    from loner_amd import hip
    hip.load()
    return _ops


# ------------------------------------------------------------------------------------------- basics
def test_mfma_fragment_layout(ops):
    assert ops.selftest_mfma(DEV) == 0.0


def test_library_refuses_cpu_tensors(ops):
    from loner_amd import hip
    spec = hip.make_net_spec(dict(otype="Frequency", n_frequencies=4), dict(n_neurons=16, n_hidden_layers=1))
    with pytest.raises(RuntimeError):
        ops.density_forward(spec, torch.zeros(int(spec.n_params)), pts=torch.zeros(4, 3))


# ------------------------------------------------------------------------------------------- rays
@pytest.mark.parametrize("i", [0, 1, 2])
def test_build_lidar_rays_matches_golden_and_oracle(ops, golden, i):
    g = golden("g1_rays")
    dirs, dist = dv(g[f"dirs_g{i}"]), dv(g[f"dist_g{i}"])
    n = dirs.shape[1]
    T = torch.from_numpy(g[f"T{i}"])
    idx = torch.arange(n, device=DEV)
    rays, depths, keep = ops.build_lidar_rays(dirs, dist, idx, dv(T[:3, :4].reshape(12)), g["ray_range"], float(g["scale"]), g["shift"])
    # all candidates vs the reference's ignore_world_cube=True output, kept subset vs its default output
    assert rel(rays, g[f"rays_all{i}"]) < 1e-6
    k = keep.bool().cpu()
    assert int(k.sum()) == g[f"rays{i}"].shape[0]                 # same rays dropped
    assert rel(rays.cpu()[k], g[f"rays{i}"]) < 1e-6
    assert np.array_equal(depths.cpu()[k].numpy(), g[f"depths{i}"])  # bit-exact (a single IEEE division)
    # backward: dL/d[R|t] against torch autograd through the oracle
    Tt = T.clone().requires_grad_(True)
    r_o, _, keep_o = OR.lidar_ray_records(torch.from_numpy(g[f"dirs_g{i}"]), torch.from_numpy(g[f"dist_g{i}"]), torch.arange(n), Tt,
                                          torch.from_numpy(g["ray_range"]), torch.tensor(float(g["scale"])), torch.from_numpy(g["shift"]),
                                          keep_all=True)
    cot = torch.randn(n, 13, generator=torch.Generator().manual_seed(i))
    (r_o * cot).sum().backward()
    seg = torch.tensor([0, n], device=DEV, dtype=torch.int32)
    dT = ops.lidar_rays_backward(dv(cot), rays, idx, seg, [dirs], dv(T[:3, :4].reshape(1, 12)), float(g["scale"]))
    assert rel(dT.reshape(3, 4), Tt.grad[:3, :4]) < 2e-4


def test_compact_rays_preserves_order_and_segments(ops):
    gen = torch.Generator().manual_seed(3)
    n = 3000
    rays = torch.randn(n, 13, generator=gen)
    depths = torch.rand(n, generator=gen)
    keep = (torch.rand(n, generator=gen) > 0.3).to(torch.uint8)
    src = torch.randint(0, 65536, (n,), generator=gen)
    seg = [0, 700, 700, 1900, 3000]                      # includes an empty segment
    r, d, s, out_seg, n_out = ops.compact_rays(dv(rays), dv(depths), keep.to(DEV), src.to(DEV), seg)
    k = keep.bool()
    m = int(k.sum())
    assert int(n_out.item()) == m
    assert torch.equal(r.cpu()[:m], rays[k]) and torch.equal(d.cpu()[:m], depths[k]) and torch.equal(s.cpu()[:m], src[k])
    expect = [int(k[:b].sum()) for b in seg]
    assert out_seg.cpu().tolist() == expect


def test_compact_rays_with_counts_or_front_record_in_the_same_launch(ops):
    """lnr_compact_rays_front: the compaction plus, in the same launch, the loss normalisers of lnr_count_opaque or the front record of
    lnr_shard_front_pack - equal to the separate calls on the compacted batch, also when a leading segment keeps nothing, when nothing
    is kept at all, and with depths on both sides of far[0]."""
    from loner_amd import hip
    gen = torch.Generator().manual_seed(13)
    n = 2500
    rays = torch.randn(n, 13, generator=gen)
    rays[:, 12] = torch.rand(n, generator=gen) * 0.5 + 0.3                        # far
    depths = torch.rand(n, generator=gen) * 1.2 - 0.1                              # some <= 0, some beyond far[0]
    src = torch.randint(0, 65536, (n,), generator=gen)
    seg = [0, 400, 1100, 1100, 2500]
    for case in ("mixed", "first segment empty", "nothing kept"):
        keep = (torch.rand(n, generator=gen) > 0.4).to(torch.uint8)
        if case == "first segment empty":
            keep[:400] = 0
        if case == "nothing kept":
            keep[:] = 0
        r, d, s_, out_seg, n_out = ops.compact_rays(dv(rays), dv(depths), keep.to(DEV), src.to(DEV), seg)
        ref_counts = ops.count_opaque(r, d, n_rays_dev=n_out)
        r2, d2, s2, seg2, n2, counts = ops.compact_rays(dv(rays), dv(depths), keep.to(DEV), src.to(DEV), seg, want_counts=True)
        m = int(n_out.item())
        assert int(n2.item()) == m and torch.equal(seg2, out_seg) and torch.equal(r2[:m], r[:m]) and torch.equal(d2[:m], d[:m]) and torch.equal(s2[:m], s_[:m])
        assert torch.equal(counts, ref_counts), (case, counts, ref_counts)
        order, cap = [1, 3, 4, 6], n + 17
        ref_rec = ops.shard_front_pack(r, out_seg, order, d, n_out, cap)
        *_, rec = ops.compact_rays(dv(rays), dv(depths), keep.to(DEV), src.to(DEV), seg, front=(order, cap))
        assert rec.shape[0] == hip.FRONT_HEADER + cap and torch.equal(rec.view(torch.int32), ref_rec.view(torch.int32)), case
    with pytest.raises(RuntimeError, match="front record"):
        ops.compact_rays(dv(rays), dv(depths), keep.to(DEV), src.to(DEV), seg, front=([0, 1, 2, 3], n - 1))


# ------------------------------------------------------------------------------------------- occupancy / samplers
def test_occ_interpolate_bit_exact(ops, golden):
    g = golden("g2_occ_lookup")
    out = ops.occ_interpolate(dv(g["grid"]), dv(g["pts"]))
    assert np.array_equal(out.cpu().numpy(), g["out"])


@pytest.mark.parametrize("S", [128, 512])
def test_occ_sampler_zero_grid_bit_identical_to_reference(ops, golden, S):
    g = golden("g4_samplers")
    z = ops.sample_rays_occ(dv(g["rays"]), dv(g["zero_grid"]), S, 1.0, u_jitter=dv(g[f"zero_u1_{S}"]), u_pdf=dv(g[f"zero_u2_{S}"]))
    assert np.array_equal(z.cpu().numpy(), g[f"zero_z{S}"])


@pytest.mark.parametrize("S", [128, 512])
def test_occ_sampler_trained_grid_bit_identical_to_oracle(ops, golden, S):
    g = golden("g4_samplers")
    z, dbg = ops.sample_rays_occ(dv(g["rays"]), dv(g["trained_grid"]), S, 1.0, u_jitter=dv(g[f"trained_u1_{S}"]),
                                 u_pdf=dv(g[f"trained_u2_{S}"]), debug=True)
    zo, st = SP.sample_occupancy(g["rays"], g["trained_grid"], S, 1.0, g[f"trained_u1_{S}"], g[f"trained_u2_{S}"], return_stages=True)
    assert np.array_equal(dbg["probs"].cpu().numpy(), st["probs"])            # correctly rounded sigmoid on both sides
    assert np.array_equal(dbg["cdf"].cpu().numpy(), st["cdf"])                # cascade sum + float64 running cdf
    assert np.array_equal(dbg["inds"].cpu().numpy(), st["inds"])              # bit-identical sample indices
    assert np.array_equal(z.cpu().numpy(), zo)
    # and against the reference itself: identical wherever its float32 exp happened to round correctly
    ref_same = (st["probs"] == g[f"trained_probs{S}"]).all(axis=1)
    zk, zr = z.cpu().numpy(), g[f"trained_z{S}"]
    frac_rows, frac_z_rows, frac_z = float(ref_same.mean()), float((zk == zr).all(axis=1).mean()), float((zk == zr).mean())
    print(f"trained grid, S={S}: rays whose pdf equals the reference's bit for bit {frac_rows:.4f}; rays with all sample depths identical "
          f"{frac_z_rows:.4f}; sample depths identical {frac_z:.6f} (torch's float32 exp is not correctly rounded: SURVEY B.5)")
    # the observed fractions of this fixture (64 rays; 99.88 % / 99.84 % of the pdf VALUES are identical): a regression moves them
    floor_rows, floor_z_rows, floor_z = {128: (0.9375, 0.96875, 0.9954), 512: (0.65625, 0.75, 0.9979)}[S]
    assert frac_rows >= floor_rows and frac_z_rows >= floor_z_rows and frac_z >= floor_z
    assert np.array_equal(zk[ref_same], zr[ref_same])


def test_occ_sampler_at_the_training_shape_g15(ops, golden):
    """G15 (512 rays x 512 samples, a trained grid - the shape the mapping loop runs; G4 is 64 rays): the kernel is bit-identical to the
    oracle on every depth, and to the REFERENCE on every ray whose pdf torch's float32 exp happened to round correctly."""
    g = golden("g15_sampler_512x512")
    z = ops.sample_rays_occ(dv(g["rays"]), dv(g["grid"]), 512, 1.0, u_jitter=dv(g["u1"]), u_pdf=dv(g["u2"])).cpu().numpy()
    zo = SP.sample_occupancy(g["rays"], g["grid"], 512, 1.0, g["u1"], g["u2"])
    assert np.array_equal(z, zo)
    zr = g["z"]
    rows, same = float((z == zr).all(axis=1).mean()), float((z == zr).mean())
    print(f"G15: kernel vs reference: rays identical {rows:.4f}, depths identical {same:.6f}, max |dz| {float(np.abs(z - zr).max()):.2e}")
    assert rows >= 0.86 and same >= 0.9965 and float(np.abs(z - zr).max()) < 1e-6


def test_occ_sampler_large_and_ragged_sample_counts(ops, golden):
    g = golden("g4_samplers")
    rays = g["rays"][:9]
    gen = torch.Generator().manual_seed(5)
    for S in (20, 100, 2048):                      # K<64 rows, non-power-of-two, test-time S
        u1, u2 = torch.rand(9, S // 2, generator=gen).numpy(), torch.rand(9, S // 2, generator=gen).numpy()
        z = ops.sample_rays_occ(dv(rays), dv(g["trained_grid"]), S, 1.0, u_jitter=dv(u1), u_pdf=dv(u2))
        zo = SP.sample_occupancy(rays, g["trained_grid"], S, 1.0, u1, u2)
        assert np.array_equal(z.cpu().numpy(), zo), S


def test_uniform_sampler_bit_exact(ops, golden):
    g = golden("g4_samplers")
    z = ops.sample_rays_uniform(dv(g["rays"]), 128, 1.0, u_jitter=dv(g["uniform_u128"]))
    assert np.array_equal(z.cpu().numpy(), g["uniform_z128"])
    z = ops.sample_rays_uniform(dv(g["rays"]), 128, 0.0)
    assert np.array_equal(z.cpu().numpy(), g["uniform_z128_det"])


def test_samplers_internal_rng_properties(ops, golden):
    g = golden("g4_samplers")
    rays = dv(g["rays"])
    z1 = ops.sample_rays_occ(rays, dv(g["trained_grid"]), 512, 1.0, seed=11)
    z2 = ops.sample_rays_occ(rays, dv(g["trained_grid"]), 512, 1.0, seed=11)
    z3 = ops.sample_rays_occ(rays, dv(g["trained_grid"]), 512, 1.0, seed=12)
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    assert bool((z1[:, 1:] >= z1[:, :-1]).all())
    assert bool((z1 >= rays[:, 11:12] - 1e-7).all()) and bool((z1 <= rays[:, 12:13] + 1e-7).all())


# ------------------------------------------------------------------------------------------- density network
NETS = {
    "default": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16),
                dict(activation="ReLU", n_neurons=64, n_hidden_layers=1)),
    # the default shape class (32 encoded features -> <= 64 ReLU neurons -> 1): "fp32" runs its matrix products on the bf16 pipe with
    # three-term operand splits (lnr_density_bf3.hip), "fp32_chain" on v_mfma_f32_16x16x4_f32 (exact fma chains, lnr_density_impl.h)
    "default_chain": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16),
                      dict(activation="ReLU", n_neurons=64, n_hidden_layers=1, precision="fp32_chain")),
    "default_n32": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=14, base_resolution=8),
                    dict(activation="ReLU", n_neurons=32, n_hidden_layers=1)),
    "default_n16": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=13, base_resolution=4),
                    dict(activation="ReLU", n_neurons=16, n_hidden_layers=1)),
    "default_n32_chain": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=14, base_resolution=8),
                          dict(activation="ReLU", n_neurons=32, n_hidden_layers=1, precision="fp32_chain")),
    "small_hash": (dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8),
                   dict(activation="ReLU", n_neurons=32, n_hidden_layers=1)),
    "hash_f4_2hidden": (dict(otype="HashGrid", n_levels=6, n_features_per_level=4, log2_hashmap_size=14, base_resolution=4, per_level_scale=1.5),
                        dict(activation="ReLU", n_neurons=64, n_hidden_layers=2)),
    "hash_f1": (dict(otype="HashGrid", n_levels=10, n_features_per_level=1, log2_hashmap_size=13, base_resolution=8),
                dict(activation="LeakyReLU", n_neurons=16, n_hidden_layers=1)),
    "hash_f8": (dict(otype="HashGrid", n_levels=3, n_features_per_level=8, log2_hashmap_size=12, base_resolution=8),
                dict(activation="Tanh", n_neurons=32, n_hidden_layers=1)),
    "freq_siren": (dict(otype="Frequency", n_frequencies=8), dict(activation="Sine", n_neurons=64, n_hidden_layers=3)),
    "freq_relu128": (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2)),
    "freq_wide256": (dict(otype="Frequency", n_frequencies=6), dict(activation="ReLU", n_neurons=256, n_hidden_layers=1)),
    "freq_relu3": (dict(otype="Frequency", n_frequencies=4), dict(activation="ReLU", n_neurons=64, n_hidden_layers=3)),
    "freq_tanh2": (dict(otype="Frequency", n_frequencies=10), dict(activation="Tanh", n_neurons=32, n_hidden_layers=2)),
    "freq_sine16": (dict(otype="Frequency", n_frequencies=3), dict(activation="Sine", n_neurons=16, n_hidden_layers=2)),
    "freq12_small": (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=32, n_hidden_layers=1)),
    "freq16_h16": (dict(otype="Frequency", n_frequencies=16), dict(activation="Softplus", n_neurons=16, n_hidden_layers=1)),
    # (Squareplus / Softplus carry tiny-cuda-nn's K_ACT = 10: oracle/network.py)
    "hash_squareplus": (dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8),
                        dict(activation="Squareplus", n_neurons=32, n_hidden_layers=2)),
    "freq_tanh128": (dict(otype="Frequency", n_frequencies=5), dict(activation="Tanh", n_neurons=128, n_hidden_layers=2)),
    "freq_relu128x3": (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=3)),
    # 256 neurons x 2..3 hidden layers: the layer-by-layer route with the split-K weight gradient (lnr_density_wide.hip)
    "freq_wide256x2": (dict(otype="Frequency", n_frequencies=6), dict(activation="ReLU", n_neurons=256, n_hidden_layers=2)),
    "hash_wide256x3": (dict(otype="HashGrid", n_levels=8, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8),
                       dict(activation="LeakyReLU", n_neurons=256, n_hidden_layers=3)),
    "freq_sine256x2": (dict(otype="Frequency", n_frequencies=12), dict(activation="Sine", n_neurons=256, n_hidden_layers=2)),
    "freq_tanh256x3": (dict(otype="Frequency", n_frequencies=6), dict(activation="Tanh", n_neurons=256, n_hidden_layers=3)),
}


def _net(name, seed=0, table_gain=1.0):
    from loner_amd import hip
    enc, net = NETS[name]
    spec_o = NW.NetworkSpec.from_config(enc, net)
    spec_h = hip.make_net_spec(enc, net)
    assert spec_o.n_params == int(spec_h.n_params) and spec_o.n_mlp_params == spec_h.n_mlp_params
    for l, lv in enumerate(spec_o.levels):
        assert (lv.res, lv.size, lv.offset, int(lv.hashed)) == (spec_h.level_res[l], spec_h.level_size[l], spec_h.level_offset[l], spec_h.level_hashed[l])
        assert abs(lv.scale - spec_h.level_scale[l]) <= 2e-7 * lv.scale
    params = NW.init_params(spec_o, seed)
    if spec_o.n_enc_params:
        params[spec_o.n_mlp_params:] *= table_gain
    return spec_o, spec_h, params


@pytest.mark.parametrize("name", list(NETS))
def test_density_forward_matches_oracle(ops, name):
    spec_o, spec_h, params = _net(name, table_gain=3000.0)
    gen = torch.Generator().manual_seed(1)
    pts = torch.rand(1000, 3, generator=gen) * 1.98 - 0.99          # 1000: not a multiple of the 16-sample tile
    sig = ops.density_forward(spec_h, dv(params), pts=dv(pts))
    ref64 = NW.density(spec_o, params.double(), pts.double())
    ref32 = NW.density(spec_o, params, pts)
    scale = float(ref32.abs().max())
    err = float((sig.cpu() - ref32).abs().max()) / scale                    # parity: same fp32 arithmetic definition
    err64 = float((sig.cpu().double() - ref64).abs().max()) / scale         # accuracy: for information
    err32 = float((ref32.double() - ref64).abs().max()) / scale
    print(f"{name}: |sigma|max={scale:.3g}  hip-vs-oracle(fp32) {err:.2e}  hip-vs-fp64 {err64:.2e}  oracle-fp32-vs-fp64 {err32:.2e}")
    assert err < 1e-5


def test_all_dense_network_at_and_beyond_the_cube_boundary(ops):
    """An all-dense network (both levels index their grid directly: no hashed level behind them in the parameter buffer) evaluated ON
    and OUTSIDE the unit-cube boundary, where grid indices reach the level size and the reference's modulo wraps them - the wrapped
    index can be the level's last entry, and the dense levels' 16-byte pair gathers must not read past it (the last level's
    end is the end of the parameter buffer).  Forward and both gradients equal the oracle, and the guard bytes behind the buffer stay
    untouched by the backward."""
    from loner_amd import hip
    enc = dict(otype="HashGrid", n_levels=2, n_features_per_level=2, log2_hashmap_size=19, base_resolution=4, per_level_scale=2.0)
    net = dict(activation="ReLU", n_neurons=32, n_hidden_layers=1)
    spec_o, spec_h = NW.NetworkSpec.from_config(enc, net), hip.make_net_spec(enc, net)
    assert all(not lv.hashed for lv in spec_o.levels)
    params = NW.init_params(spec_o, 5)
    params[spec_o.n_mlp_params:] *= 3000.0
    gen = torch.Generator().manual_seed(11)
    corners = torch.tensor([[sx, sy, sz] for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)])
    faces = torch.rand(64, 3, generator=gen) * 2 - 1
    faces[torch.arange(64), torch.randint(0, 3, (64,), generator=gen)] = 1.0          # one coordinate exactly on the far face
    outside = torch.rand(64, 3, generator=gen) * 0.4 + 0.95                           # up to 35 % beyond it
    inside = torch.rand(120, 3, generator=gen) * 1.98 - 0.99
    pts = torch.cat([corners, faces, outside, inside])
    # the parameters sit at the very END of an allocation with a poisoned guard behind them
    n = int(spec_h.n_params)
    buf = torch.full((n + 64,), float("nan"), device=DEV)
    buf[:n] = dv(params)
    sig = ops.density_forward(spec_h, buf[:n], pts=dv(pts))
    ref = NW.density(spec_o, params, pts)
    assert torch.isfinite(sig).all()
    assert float((sig.cpu() - ref).abs().max()) / float(ref.abs().max()) < 1e-5
    d_sigma = torch.randn(pts.shape[0], generator=gen)
    gbuf = torch.zeros(n + 64, device=DEV)
    d_pts = ops.density_backward(spec_h, buf[:n], dv(d_sigma), gbuf[:n], pts=dv(pts), want_d_pts=True)
    p32, x32 = params.clone().requires_grad_(True), pts.clone().requires_grad_(True)
    (NW.density(spec_o, p32, x32) * d_sigma).sum().backward()
    assert rel(gbuf[:n], p32.grad) < 2e-5 and float(gbuf[n:].abs().sum()) == 0.0
    assert rel(d_pts, x32.grad) < 1e-4


@pytest.mark.parametrize("name", list(NETS))
def test_density_backward_matches_oracle_autograd(ops, name):
    spec_o, spec_h, params = _net(name, seed=2, table_gain=3000.0)
    gen = torch.Generator().manual_seed(4)
    n = 777
    pts = (torch.rand(n, 3, generator=gen) * 1.9 - 0.95)
    d_sigma = torch.randn(n, generator=gen)
    d_sigma[torch.rand(n, generator=gen) < 0.3] = 0.0                 # exact zeros exercise the skip paths
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_pts = ops.density_backward(spec_h, dv(params), dv(d_sigma), grad, pts=dv(pts), want_d_pts=True)
    # reference = the oracle in the same precision (fp32); fp64 only for information: at the finest hash levels
    # (scale 2^19) the float32 rounding of x*scale+0.5 is part of the function's definition.
    p32 = params.clone().requires_grad_(True)
    x32 = pts.clone().requires_grad_(True)
    (NW.density(spec_o, p32, x32) * d_sigma).sum().backward()
    p64 = params.double().requires_grad_(True)
    x64 = pts.double().requires_grad_(True)
    (NW.density(spec_o, p64, x64) * d_sigma.double()).sum().backward()
    e_p, e_x = rel(grad, p32.grad), rel(d_pts, x32.grad)
    print(f"{name}: dparams rel {e_p:.2e} (fp32-vs-fp64 {rel(p32.grad, p64.grad):.2e})  dpts rel {e_x:.2e} (fp32-vs-fp64 {rel(x32.grad, x64.grad):.2e})")
    assert e_p < 2e-5
    assert e_x < 2e-4
    assert rel_layers(spec_o, grad, p32.grad) < 2e-4                 # matrix by matrix (structural errors; the precision statement is e_p)
    # without input gradients the parameter gradient must be the same
    grad2 = torch.zeros_like(grad)
    assert ops.density_backward(spec_h, dv(params), dv(d_sigma), grad2, pts=dv(pts), want_d_pts=False) is None
    assert rel(grad2, grad) < 1e-6


@pytest.mark.parametrize("name", ["freq_relu128", "freq_relu128x3", "freq_wide256", "freq_siren", "hash_f4_2hidden", "freq_wide256x2", "freq_tanh256x3"])
def test_general_fp32_backward_over_many_steps(ops, name):
    """(the two 256 x n networks: the layer-by-layer route, lnr_density_wide.hip - 40 013 points are a partly filled chunk whose weight
    gradient is split over 128 sample ranges; test_wide_networks_across_chunks covers several chunks.  The three-layer one is smooth
    (Tanh): with 256 x 3 piecewise-linear units and 40 013 points the fp32 oracle itself is 8e-3 from its fp64 self - pre-activations
    within rounding of a kink change a unit's derivative - so a kinked network of that size cannot be held to 2e-5 by anyone)
    mlp_backward_regs_kernel (lnr_density_regs.h) beyond one step per workgroup: 256 workgroups x 64 samples per step, so 40 013 points
    are three steps with a ragged last tile, the inputs of step i + 1 requested during step i; a stretch of 20 000 points without gradient
    makes whole steps take the workgroup-uniform skip (their d_feature rows must still come out zero), single points without gradient sit
    inside live tiles.  The five networks cover the three homes of the weights (all in LDS, hidden matrices only, none), 4 / 8 / 16 row
    tiles and one to three hidden layers."""
    spec_o, spec_h, params = _net(name, seed=3, table_gain=3000.0)
    gen = torch.Generator().manual_seed(11)
    n = 40013
    pts = (torch.rand(n, 3, generator=gen) * 1.9 - 0.95)
    d_sigma = torch.randn(n, generator=gen)
    d_sigma[9000:29000] = 0.0
    d_sigma[torch.rand(n, generator=gen) < 0.2] = 0.0
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_pts = ops.density_backward(spec_h, dv(params), dv(d_sigma), grad, pts=dv(pts), want_d_pts=True)
    p32 = params.clone().requires_grad_(True)
    x32 = pts.clone().requires_grad_(True)
    (NW.density(spec_o, p32, x32) * d_sigma).sum().backward()
    e_p, e_x = rel(grad, p32.grad), rel(d_pts, x32.grad)
    print(f"{name}: {n} points, dparams rel {e_p:.2e}, dpts rel {e_x:.2e}")
    assert e_p < 2e-5 and e_x < 2e-4
    assert rel_layers(spec_o, grad, p32.grad) < 2e-4
    assert float(d_pts[9000:29000].abs().max()) == 0.0
    grad2 = torch.zeros_like(grad)
    assert ops.density_backward(spec_h, dv(params), dv(d_sigma), grad2, pts=dv(pts), want_d_pts=False) is None
    assert rel(grad2, grad) < 1e-6


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_wide_networks_across_chunks(ops, prec):
    """256 x 2 through the layer-by-layer route over MORE than two chunks of 131 072 samples (rays form, a device-side live count that
    ends inside the last chunk): sigma of a subset and the full gradient against the oracle - evaluated chunk by chunk on the CPU with
    the same arithmetic model - and the padding rays beyond the live count untouched."""
    from loner_amd import hip
    enc, net = NETS["freq_wide256x2"]
    net = dict(net, precision=prec, activation="Tanh")             # (smooth: see test_general_fp32_backward_over_many_steps)
    spec_o, spec_h = NW.NetworkSpec.from_config(enc, net), hip.make_net_spec(enc, net)
    params = NW.init_params(spec_o, 4)
    gen = torch.Generator().manual_seed(21)
    n_rays, S, live = 2200, 128, 2100                      # 281 600 sample slots, 268 800 live: two full chunks + 6 656 samples
    rays = torch.zeros(n_rays, 13)
    rays[:, 0:3] = torch.rand(n_rays, 3, generator=gen) * 0.6 - 0.3
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=1)
    z = torch.sort(torch.rand(n_rays, S, generator=gen) * 0.6, dim=1).values
    d_sigma = torch.randn(n_rays, S, generator=gen)
    d_sigma[torch.rand(n_rays, S, generator=gen) < 0.5] = 0.0
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    R, Z, P = dv(rays), dv(z), dv(params)
    sig = ops.density_forward(spec_h, P, rays=R, z=Z, n_rays_dev=n_dev)
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_rays = torch.zeros(n_rays, 13, device=DEV)
    ops.density_backward(spec_h, P, dv(d_sigma), grad, rays=R, z=Z, n_rays_dev=n_dev, reuse_features=True, d_rays=d_rays)
    p = params.clone().requires_grad_(True)
    ref_rows = torch.cat([torch.arange(0, 3), torch.arange(1022, 1026), torch.arange(2046, 2050), torch.arange(live - 3, live)])   # around the chunk seams (1024, 2048) and the end
    total = torch.zeros(())
    sig_ref = {}
    for lo in range(0, live, 300):                                       # the oracle in pieces (memory), the gradient accumulates
        hi = min(lo + 300, live)
        pts = (rays[lo:hi, None, 0:3] + rays[lo:hi, None, 3:6] * z[lo:hi, :, None]).reshape(-1, 3)
        s_ = NW.density(spec_o, p, pts).reshape(hi - lo, S)
        (s_ * d_sigma[lo:hi]).sum().backward()
        for r in ref_rows.tolist():
            if lo <= r < hi:
                sig_ref[r] = s_[r - lo].detach()
    tol_s, tol_g = (1e-5, 3e-5) if prec == "fp32" else (2e-3, 3e-3)
    scale = max(float(v.abs().max()) for v in sig_ref.values())
    for r, v in sig_ref.items():
        assert float((sig[r].cpu() - v).abs().max()) / scale < tol_s, r
    e_g = rel(grad, p.grad)
    print(f"256 x 2 over three chunks ({prec}): dparams rel {e_g:.2e}")
    assert e_g < tol_g
    assert rel_layers(spec_o, grad, p.grad) < 10 * tol_g
    assert float(d_rays[live:].abs().max()) == 0.0 and float(d_rays[:live, 0:6].abs().max()) > 0.0


@pytest.mark.parametrize("name", ["default", "freq_siren"])
def test_density_clips_non_finite_outputs_like_the_reference(ops, name):
    """nerf_tcnn.py:74-78: non-finite densities are replaced (nan_to_num: NaN -> 0, +-inf -> the dtype's extremes)."""
    spec_o, spec_h, params = _net(name, seed=1, table_gain=3000.0)
    gen = torch.Generator().manual_seed(3)
    pts = torch.rand(500, 3, generator=gen) * 1.9 - 0.95
    bad = params.clone()
    h, nh, ind = int(spec_h.n_neurons), int(spec_h.n_hidden), int(spec_h.in_dim)
    wo = h * ind + (nh - 1) * h * h                     # output layer, row 0 = sigma
    bad[wo:wo + h] = float("inf")
    ref = NW.density(spec_o, bad, pts)
    assert not torch.isfinite(ref).all()                # the oracle (like tinycudann) produces inf / NaN here
    sig = ops.density_forward(spec_h, dv(bad), pts=dv(pts)).cpu()
    fmax = torch.finfo(torch.float32).max
    assert torch.isfinite(sig).all()
    assert torch.equal(sig, torch.nan_to_num(ref, nan=0.0, posinf=fmax, neginf=-fmax))


def test_density_rays_form_equals_points_form(ops, golden):
    g = golden("g4_samplers")
    spec_o, spec_h, params = _net("small_hash", table_gain=3000.0)
    rays, z = dv(g["rays"]), dv(g["zero_z128"])
    s1 = ops.density_forward(spec_h, dv(params), rays=rays, z=z)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    s2 = ops.density_forward(spec_h, dv(params), pts=pts.reshape(-1, 3)).reshape(z.shape)
    print('rays-form vs points-form max diff', float((s1 - s2).abs().max()))
    assert rel(s1, s2) < 1e-5


@pytest.mark.parametrize("name", ["small_hash", "hash_f1", "freq_siren"])
@pytest.mark.parametrize("n", [1, 15, 17, 255, 257, 1000])
def test_density_ragged_point_counts(ops, name, n):
    """Tile (16), wave (64) and workgroup (256) boundaries of the level-major pipeline: forward, gradients, and the
    feature planes kept by the forward (reuse_features) against a fresh backward."""
    spec_o, spec_h, params = _net(name, seed=5, table_gain=3000.0)
    gen = torch.Generator().manual_seed(100 + n)
    pts = torch.rand(n, 3, generator=gen) * 1.9 - 0.95
    d_sigma = torch.randn(n, generator=gen)
    p_dev, x_dev = dv(params), dv(pts)
    sig = ops.density_forward(spec_h, p_dev, pts=x_dev)
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_pts = ops.density_backward(spec_h, p_dev, dv(d_sigma), grad, pts=x_dev, want_d_pts=True, reuse_features=True)
    p32 = params.clone().requires_grad_(True)
    x32 = pts.clone().requires_grad_(True)
    out = NW.density(spec_o, p32, x32)
    (out * d_sigma).sum().backward()
    assert rel(sig, out.detach()) < 1e-5
    assert rel(grad, p32.grad) < 2e-5 and rel(d_pts, x32.grad) < 2e-4
    grad2 = torch.zeros_like(grad)
    ops.density_forward(spec_h, p_dev, pts=dv(torch.zeros(max(n // 2, 1), 3)))          # clobber the kept feature planes
    d_pts2 = ops.density_backward(spec_h, p_dev, dv(d_sigma), grad2, pts=x_dev, want_d_pts=True, reuse_features=False)
    assert rel(grad2, grad) < 1e-6 and rel(d_pts2, d_pts) < 1e-6
    with pytest.raises(RuntimeError):                                                     # stale planes are refused, not used
        ops.density_backward(spec_h, p_dev, dv(d_sigma), grad2, pts=x_dev, reuse_features=True)


def test_density_dead_rays_and_empty_batches(ops, golden):
    """n_rays_dev < n_rays (rays dropped by the cube clip without a host sync): live rows equal a call on the live rows
    only, dead rows are never touched; zero live rays and zero points are no-ops."""
    g = golden("g4_samplers")
    spec_o, spec_h, params = _net("default", seed=3, table_gain=3000.0)
    rays, z = dv(g["rays"]), dv(g["zero_z128"])
    n, S = z.shape
    live = n // 3
    p_dev = dv(params)
    gen = torch.Generator().manual_seed(9)
    d_sigma = dv(torch.randn(n, S, generator=gen))
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    sig = torch.full((n, S), 7.0, device=DEV)
    sig_live = ops.density_forward(spec_h, p_dev, rays=rays, z=z, n_rays_dev=n_dev)
    ref = ops.density_forward(spec_h, p_dev, rays=rays[:live].contiguous(), z=z[:live].contiguous())
    assert torch.equal(sig_live[:live], ref)
    g1 = torch.zeros(int(spec_h.n_params), device=DEV); g2 = torch.zeros_like(g1)
    d1 = ops.density_backward(spec_h, p_dev, d_sigma, g1, rays=rays, z=z, n_rays_dev=n_dev, want_d_pts=True)
    d2 = ops.density_backward(spec_h, p_dev, d_sigma[:live].contiguous(), g2, rays=rays[:live].contiguous(), z=z[:live].contiguous(),
                              want_d_pts=True)
    assert rel(g1, g2) < 1e-6 and rel(d1[:live], d2) < 1e-6
    # nothing alive / nothing to do
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    g3 = torch.zeros_like(g1)
    ops.density_forward(spec_h, p_dev, rays=rays, z=z, n_rays_dev=zero)
    ops.density_backward(spec_h, p_dev, d_sigma, g3, rays=rays, z=z, n_rays_dev=zero, want_d_pts=True)
    assert float(g3.abs().max()) == 0.0
    assert ops.density_forward(spec_h, p_dev, pts=torch.zeros(0, 3, device=DEV)).numel() == 0
    del sig


@pytest.mark.parametrize("name,S", [("default", 128), ("default", 100), ("small_hash", 64), ("hash_f4_2hidden", 128), ("freq_siren", 128)])
def test_density_backward_ray_gradient_mode(ops, golden, name, S):
    """d_rays mode of lnr_density_backward (the per-ray reduction of dL/dxyz inside the encode kernels when a wave's 64
    samples share a ray, the planes + lnr_points_grad_to_rays route otherwise) == d_pts followed by lnr_points_grad_to_rays,
    with identical parameter gradients; rays dropped on the device (n_rays_dev) stay untouched."""
    g = golden("g4_samplers")
    spec_o, spec_h, params = _net(name, seed=7, table_gain=3000.0)
    rays = dv(g["rays"])
    n = rays.shape[0]
    gen = torch.Generator().manual_seed(S)
    z = dv(torch.sort(torch.rand(n, S, generator=gen) * 0.5 + 0.01, dim=1).values)
    d_sigma = dv(torch.randn(n, S, generator=gen))
    p_dev = dv(params)
    live = n - 5
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    g1 = torch.zeros(int(spec_h.n_params), device=DEV); g2 = torch.zeros_like(g1)
    base = dv(torch.randn(n, 13, generator=gen))                      # d_rays is accumulated into, not overwritten
    d_pts = ops.density_backward(spec_h, p_dev, d_sigma, g1, rays=rays, z=z, n_rays_dev=n_dev, want_d_pts=True)
    ref = base.clone()
    ops.points_grad_to_rays(d_pts, z, ref, n_rays_dev=n_dev)
    out = base.clone()
    assert ops.density_backward(spec_h, p_dev, d_sigma, g2, rays=rays, z=z, n_rays_dev=n_dev, d_rays=out) is None
    assert rel(g2, g1) < 1e-6
    assert rel(out[:live, :6], ref[:live, :6]) < 2e-5 and torch.equal(out[:, 6:], base[:, 6:]) and torch.equal(out[live:], base[live:])


# ------------------------------------------------------------------------------------------- rendering
def test_render_forward_backward_matches_golden(ops, golden):
    g = golden("g5_render")
    n, S = g["z"].shape
    rays = torch.zeros(n, 13)
    rays[:, 3:6] = torch.from_numpy(g["dirs"]); rays[:, 12] = torch.from_numpy(g["far"])[:, 0]
    sigma, z, noise = dv(g["sigma"]), dv(g["z"]), dv(g["noise"])
    depth, weights, opacity, variance = ops.render_forward(sigma, z, dv(rays), noise=noise, noise_std=1.0)
    assert rel(depth, g["depth"]) < 1e-5 and rel(weights, g["weights"]) < 1e-5
    assert rel(opacity, g["opacity"]) < 1e-5 and rel(variance, g["variance"]) < 1e-4
    d_sigma, d_rays = ops.render_backward(sigma, z, dv(rays), dv(g["cot_depth"]), dv(g["cot_weights"]), dv(g["cot_opacity"]),
                                          dv(g["cot_variance"]), noise=noise, noise_std=1.0)
    assert rel(d_sigma, g["dsigma"]) < 1e-4
    assert rel(d_rays[:, 3:6], g["ddirs"]) < 1e-4
    assert rel(d_rays[:, 12], g["dfar"][:, 0]) < 1e-4


def test_raw2outputs_is_differentiable_like_the_reference(ops, golden):
    """models.rendering_tcnn.raw2outputs (rendering_tcnn.py:71-147, plain autograd in the reference): called with the reference's
    signature, its outputs AND the gradients torch autograd delivers to raw, rays_d and far equal the reference's (G5); a z_vals that
    requires a gradient is refused (the kernels treat sample depths as constants, as the reference's detached samplers make them)."""
    from loner_amd.models.rendering_tcnn import raw2outputs
    g = golden("g5_render")
    raw = dv(g["sigma"])[..., None].clone().requires_grad_(True)
    dirs, far = dv(g["dirs"]).clone().requires_grad_(True), dv(g["far"]).clone().requires_grad_(True)
    _, depth, weights, opacity, variance = raw2outputs(raw, dv(g["z"]), dirs, raw_noise_std=1.0, sigma_only=True, far=far, ret_var=True,
                                                       noise=dv(g["noise"]))
    assert rel(depth, g["depth"]) < 1e-5 and rel(weights, g["weights"]) < 1e-5 and rel(variance, g["variance"]) < 1e-4
    ((depth * dv(g["cot_depth"])).sum() + (weights * dv(g["cot_weights"])).sum() + (opacity * dv(g["cot_opacity"])).sum() +
     (variance * dv(g["cot_variance"])).sum()).backward()
    assert rel(raw.grad[..., 0], g["dsigma"]) < 1e-4 and rel(dirs.grad, g["ddirs"]) < 1e-4 and rel(far.grad, g["dfar"]) < 1e-4
    with pytest.raises(NotImplementedError):
        raw2outputs(raw.detach(), dv(g["z"]).requires_grad_(True), dirs.detach(), sigma_only=True, far=far.detach())
    with torch.no_grad():          # and without a graph it is the plain forward
        out = raw2outputs(raw, dv(g["z"]), dirs, raw_noise_std=1.0, sigma_only=True, far=far, noise=dv(g["noise"]))
    assert out[4] is None and not out[1].requires_grad and rel(out[1], g["depth"]) < 1e-5


def test_render_ragged_and_long_rays_match_oracle(ops):
    gen = torch.Generator().manual_seed(9)
    for S in (2, 7, 100, 2048):
        n = 10
        z = torch.sort(torch.rand(n, S, generator=gen) * 0.5 + 0.01, dim=1).values
        sigma = torch.randn(n, S, generator=gen) * 20
        rays = torch.zeros(n, 13)
        rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1) * 1.1
        rays[:, 12] = 0.6
        out = ORD.composite(sigma.double(), z.double(), rays[:, 3:6].double(), rays[:, 12:13].double())
        depth, weights, opacity, variance = ops.render_forward(dv(sigma), dv(z), dv(rays))
        assert rel(depth, out["depth"]) < 1e-5 and rel(weights, out["weights"]) < 1e-5, S
        assert rel(opacity, out["opacity"]) < 1e-5 and rel(variance, out["variance"]) < 1e-4, S


def test_target_weights_and_logit_grad_match_golden(ops, golden):
    g = golden("g6_targets")
    s, gt, eps = dv(g["s"]), dv(g["g"]), dv(g["eps"])
    assert rel(ops.weights_gt(s, gt, 1.37), g["w_float"]) < 1e-5
    assert rel(ops.weights_gt(s, gt, eps), g["w_tensor"]) < 1e-5
    assert rel(ops.weights_gt(s, gt, eps, normalise=False), g["w_unnorm"]) < 1e-5
    assert np.array_equal(ops.logits_grad(s, gt).cpu().numpy(), g["logits_grad"])


# ------------------------------------------------------------------------------------------- fused loss
def _g8_setup(golden):
    from loner_amd import hip
    g = golden("g8_compute_loss")
    enc = dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8)
    net = dict(activation="ReLU", n_neurons=32, n_hidden_layers=1)
    return g, NW.NetworkSpec.from_config(enc, net), hip.make_net_spec(enc, net)


def test_fused_loss_matches_reference_compute_loss(ops, golden):
    """loss value, per-ray stats and every gradient of Optimizer.compute_loss (reference, G8), on the
    reference's own sample depths (sampler parity is asserted separately)."""
    from loner_amd import hip
    g, spec_o, spec_h = _g8_setup(golden)
    rays, z, depths, params = dv(g["rays"]), dv(g["z"]), dv(g["depths"]), dv(g["params"])
    noise = dv(g["noise"])
    counts = ops.count_opaque(rays, depths)
    cfg = hip.LossConfig(selection=0, min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0, depth_lambda=0.005, min_eps=0.5, fixed_eps=3.0)
    sigma = ops.density_forward(spec_h, params, rays=rays, z=z)
    loss, d_sigma, d_rays, stats, w = ops.los_loss_fused(sigma, z, rays, depths, float(g["scale"]), cfg, counts, noise=noise,
                                                        noise_std=1.0, want_stats=True, want_weights=True)
    n = rays.shape[0]
    far0 = g["rays"][0, 12]
    opaque = (g["depths"] > 0) & ~(g["depths"] > far0)
    assert counts.cpu().tolist() == [n, int(opaque.sum())] and 0 < int(opaque.sum()) < n
    assert rel(w, g["weights"]) < 1e-4
    assert rel(stats[:, 0], g["depth"]) < 1e-4                     # north_star: depth within 1e-4 relative
    assert rel(stats[:, 1], g["opacity"]) < 1e-4 and rel(stats[:, 2], g["variance"]) < 1e-3
    assert abs(float(loss[0]) - float(g["loss"])) / float(g["loss"]) < 1e-4
    assert abs(float(stats[:, 6].mean()) - float(g["depth_eps"])) < 1e-4
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_pts = ops.density_backward(spec_h, params, d_sigma, grad, rays=rays, z=z, want_d_pts=True)
    ops.points_grad_to_rays(d_pts, z, d_rays)
    assert rel(grad, g["dparams"]) < 2e-4
    assert rel(d_rays, g["drays"]) < 2e-4
    # pose tail: rays -> [R|t] (HIP) -> 6-vector (torch autograd), against the reference's pose gradients
    from loner_amd.common.pose_utils import tensor_to_transform
    seg, lo = [0], 0
    dirs, srcs, poses = [], [], []
    for k in range(2):
        n_k = g[f"dirs{k}"].shape[1]
        dirs.append(dv(g[f"dirs{k}"])); srcs.append(torch.arange(n_k, device=DEV))
        poses.append(torch.from_numpy(g[f"pose{k}"]).clone().requires_grad_(True))
        seg.append(seg[-1] + n_k)
    assert seg[-1] == n                                                 # no ray was dropped in this fixture
    T = torch.stack([tensor_to_transform(p)[:3, :4].reshape(12) for p in poses])
    dT = ops.lidar_rays_backward(d_rays, rays, torch.cat(srcs), torch.tensor(seg, device=DEV, dtype=torch.int32), dirs, dv(T.detach()), float(g["scale"]))
    T.backward(dT.cpu())
    assert rel(poses[0].grad, g["dpose0"]) < 5e-4 and rel(poses[1].grad, g["dpose1"]) < 5e-4


def test_fused_loss_on_shards_with_broadcast_far0_equals_whole_batch(ops, golden):
    """The keyframe-sharded window: each rank evaluates its rays with the GLOBAL normalisers and with far[0] of the WHOLE
    batch (the reference's `depth > far[0]` quirk, optimizer.py:460-461).  Two shards evaluated that way must add up to the
    single-batch loss and reproduce its gradients row for row; with a shard's own first ray they do not."""
    from loner_amd import hip
    g, spec_o, spec_h = _g8_setup(golden)
    rays, z, depths, params = dv(g["rays"]).clone(), dv(g["z"]), dv(g["depths"]), dv(g["params"])
    noise = dv(g["noise"])
    n = rays.shape[0]
    cut = n // 3
    rays[0, 12] = 0.3 * rays[0, 12] + 0.7 * rays[0, 11]              # first ray of the batch clipped short: many depths exceed far[0]
    cfg = hip.LossConfig(selection=0, min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0, depth_lambda=0.005, min_eps=0.5, fixed_eps=3.0)
    sigma = ops.density_forward(spec_h, params, rays=rays, z=z)
    counts = ops.count_opaque(rays, depths)
    whole = ops.los_loss_fused(sigma, z, rays, depths, float(g["scale"]), cfg, counts, noise=noise, noise_std=1.0)
    far0 = rays[0:1, 12].clone()
    parts, cnt = [], torch.zeros(2, device=DEV, dtype=torch.int32)
    sl = [slice(0, cut), slice(cut, n)]
    for s_ in sl:
        cnt += ops.count_opaque(rays[s_].contiguous(), depths[s_].contiguous(), far0=far0)
    assert torch.equal(cnt, counts) and 0 < int(counts[1]) < n
    own = ops.count_opaque(rays[sl[1]].contiguous(), depths[sl[1]].contiguous())          # shard 1 with ITS first ray: a different mask
    assert int(own[1]) != int(ops.count_opaque(rays[sl[1]].contiguous(), depths[sl[1]].contiguous(), far0=far0)[1])
    for s_ in sl:
        parts.append(ops.los_loss_fused(sigma[s_].contiguous(), z[s_].contiguous(), rays[s_].contiguous(), depths[s_].contiguous(),
                                        float(g["scale"]), cfg, counts, noise=noise[s_].contiguous(), noise_std=1.0, far0=far0))
    assert rel(parts[0][0][:5] + parts[1][0][:5], whole[0][:5]) < 1e-6
    assert torch.equal(torch.cat([parts[0][1], parts[1][1]]), whole[1])                   # d_sigma, bit for bit
    assert torch.equal(torch.cat([parts[0][2], parts[1][2]]), whole[2])                   # direct d_rays


@pytest.mark.parametrize("selection", ["L2_JS", "L1_LOS", "L2_LOS"])
def test_fused_loss_other_selections_match_oracle(ops, golden, selection):
    from loner_amd import hip
    g, spec_o, spec_h = _g8_setup(golden)
    rays, z, depths, params = dv(g["rays"]), dv(g["z"]), dv(g["depths"]), dv(g["params"])
    cfg_o = OL.LossConfig(selection=selection)
    it = 3
    eps_fixed = max(cfg_o.eps0 * cfg_o.eps_decay_rate ** (it / cfg_o.eps_decay_steps), cfg_o.min_eps)
    cfg = hip.LossConfig(selection=hip.LOSS_SELECTIONS[selection], min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0,
                         depth_lambda=0.005, min_eps=0.5, fixed_eps=eps_fixed)
    sigma = ops.density_forward(spec_h, params, rays=rays, z=z)
    counts = ops.count_opaque(rays, depths)
    loss, d_sigma, d_rays, _, _ = ops.los_loss_fused(sigma, z, rays, depths, float(g["scale"]), cfg, counts, noise=dv(g["noise"]), noise_std=1.0)
    sig_o = sigma.cpu().clone().requires_grad_(True)
    rays_o = torch.from_numpy(g["rays"]).clone().requires_grad_(True)
    zo = torch.from_numpy(g["z"])
    out = ORD.composite(sig_o, zo, rays_o[:, 3:6], rays_o[:, 12:13], torch.from_numpy(g["noise"]))
    lo, _ = OL.lidar_loss(out, zo, rays_o, torch.from_numpy(g["depths"]), torch.tensor(float(g["scale"])), cfg_o, it)
    lo.backward()
    assert abs(float(loss[0]) - float(lo)) / float(lo) < 1e-4
    assert rel(d_sigma, sig_o.grad) < 2e-4
    assert rel(d_rays[:, [3, 4, 5, 12]], rays_o.grad[:, [3, 4, 5, 12]]) < 2e-4


# ------------------------------------------------------------------------------------------- optimisers
def test_adam_step_matches_oracle(ops):
    gen = torch.Generator().manual_seed(0)
    n = 10007
    p0 = torch.randn(n, generator=gen)
    p_o = p0.clone()
    adam = MS.AdamState([p_o], [0.01])
    p, m, v = dv(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(n, generator=gen) * (10.0 ** (step - 3))
        gr[::7] = 0.0
        gd = dv(gr)
        ops.adam_step(p, gd, m, v, 0.01, step, zero_grad=True)
        adam.step([gr])
        assert float(gd.abs().max()) == 0.0
    assert rel(p, p_o) < 1e-6


def test_occ_grid_step_matches_oracle(ops, golden):
    g = golden("g4_samplers")
    rays, z = g["rays"], g["trained_z128"]
    gen = torch.Generator().manual_seed(2)
    V = 24
    grid0 = torch.randn(1, 1, V, V, V, generator=gen)
    depths = torch.from_numpy(g["depths"])
    scale = 85.76
    pts = ORD.sample_points(torch.from_numpy(rays), torch.from_numpy(z))
    expect = OC.grid_step(grid0, pts, torch.from_numpy(z) * scale, depths[:, None] * scale, 1e-2)
    grid = dv(grid0.clone())
    ops.occ_grid_step(grid, dv(rays), dv(z), dv(depths), scale, 1e-2)
    assert float((expect - grid0).abs().max()) > 0
    assert rel(grid.cpu() - grid0, expect - grid0) < 1e-4
    # two-stage form used when the window is sharded
    # the form the optimiser uses: 64-bit fixed-point accumulators, then apply - reproducible bit for bit
    grid2, buf = dv(grid0.clone()), torch.zeros(V ** 3, device=DEV, dtype=torch.int64)
    ops.occ_grid_step(grid2, dv(rays), dv(z), dv(depths), scale, 1e-2, grad_buf=buf)
    ops.occ_grid_apply(grid2, buf, 1e-2)
    assert rel(grid2.cpu() - grid0, expect - grid0) < 1e-4 and int(buf.abs().max()) == 0
    grid3 = dv(grid0.clone())
    ops.occ_grid_step(grid3, dv(rays), dv(z), dv(depths), scale, 1e-2, grad_buf=buf)
    ops.occ_grid_apply(grid3, buf, 1e-2)
    assert torch.equal(grid3, grid2)


# ------------------------------------------------------------------------------------------- pose kernels / window build
def test_pose_forward_backward_match_oracle_autograd(ops):
    gen = torch.Generator().manual_seed(6)
    p = torch.randn(9, 6, generator=gen)
    p[0, 3:] = 0.0                      # identity rotation: exercises the small-angle branch and |aa| = 0
    p[1, 3:] *= 1e-8
    T = ops.pose_forward(dv(p))
    po = p.clone().requires_grad_(True)
    To = torch.stack([OP.transform_from_pose6(po[i])[:3, :4].reshape(12) for i in range(9)])
    assert rel(T, To) < 1e-6
    cot = torch.randn(9, 12, generator=gen)
    To.backward(cot)
    mask = torch.tensor([1, 1, 0, 1, 1, 1, 1, 0, 1], dtype=torch.uint8)
    g = ops.pose_backward(dv(p), dv(cot), mask=mask.to(DEV))
    expect = po.grad * mask[:, None].float()
    assert rel(g, expect) < 1e-5
    assert torch.isfinite(g).all() and float(g[2].abs().max()) == 0.0
    g2 = ops.pose_backward(dv(p), dv(cot), mask=mask.to(DEV), out=g.clone(), accumulate=True)
    assert rel(g2, 2 * expect) < 1e-5


def test_build_window_rays_equals_per_keyframe_build(ops, golden):
    g = golden("g1_rays")
    dirs = [dv(g[f"dirs_g{i}"]) for i in range(3)]
    dist = [dv(g[f"dist_g{i}"]) for i in range(3)]
    T = torch.stack([torch.from_numpy(g[f"T{i}"])[:3, :4].reshape(12) for i in range(3)])
    counts = [300, 64, 500]
    sky = dv(torch.nn.functional.normalize(torch.randn(3, 40, generator=torch.Generator().manual_seed(1)), dim=0))
    tab = ops.WindowTables(dirs + [sky], dist + [None], [0.0, 0.0, 0.0, 51.0], counts + [32], [0, 1, 2, 2])
    gen = torch.Generator().manual_seed(2)
    idx = torch.cat([torch.randint(0, 700, (c,), generator=gen) for c in counts] + [torch.randint(0, 40, (32,), generator=gen)])
    rays, depths, keep, src = ops.build_window_rays(tab, dv(T), g["ray_range"], float(g["scale"]), g["shift"], index=idx.to(DEV))
    assert torch.equal(src.cpu(), idx)
    lo = 0
    for s in range(3):
        r1, d1, k1 = ops.build_lidar_rays(dirs[s], dist[s], idx[lo:lo + counts[s]].to(DEV), dv(T[s]), g["ray_range"], float(g["scale"]), g["shift"])
        assert torch.equal(rays[lo:lo + counts[s]], r1) and torch.equal(depths[lo:lo + counts[s]], d1) and torch.equal(keep[lo:lo + counts[s]], k1)
        lo += counts[s]
    assert torch.allclose(depths[lo:], torch.full((32,), 51.0 / float(g["scale"]), device=DEV))     # sky: constant distance
    # in-kernel index draw: in range, reproducible per seed, different across seeds
    r_a = ops.build_window_rays(tab, dv(T), g["ray_range"], float(g["scale"]), g["shift"], seed=5)
    r_b = ops.build_window_rays(tab, dv(T), g["ray_range"], float(g["scale"]), g["shift"], seed=5)
    r_c = ops.build_window_rays(tab, dv(T), g["ray_range"], float(g["scale"]), g["shift"], seed=6)
    assert torch.equal(r_a[3], r_b[3]) and not torch.equal(r_a[3], r_c[3])
    assert int(r_a[3][:864].min()) >= 0 and int(r_a[3][:864].max()) < 700 and int(r_a[3][864:].max()) < 40
    assert len(torch.unique(r_a[3][:300])) > 150


# ------------------------------------------------------------------------------------------- precision / rounding modes
def _default_pair(precision="fp32", pos_rounding="fma", seed=0, gain=2000.0):
    from loner_amd import hip
    enc, net = dict(NETS["default"][0]), dict(NETS["default"][1])
    enc["pos_rounding"] = pos_rounding
    net["precision"] = precision
    spec_o = NW.NetworkSpec.from_config(enc, net)
    spec_h = hip.make_net_spec(enc, net)
    params = NW.init_params(spec_o, seed)
    params[spec_o.n_mlp_params:] *= gain
    return spec_o, spec_h, params


@pytest.mark.parametrize("mode", ["fma", "mul_add"])
def test_grid_position_rounding_convention(ops, mode):
    """tiny-cuda-nn computes the lookup position with one fused multiply-add; the kernels follow the spec's switch and agree
    with the oracle in EITHER convention to fp32 noise, while the two conventions differ from each other by much more at the
    finest levels (one ulp of the position = 1/32 cell at scale 5.2e5)."""
    spec_o, spec_h, params = _default_pair(pos_rounding=mode)
    other_o, _, _ = _default_pair(pos_rounding="mul_add" if mode == "fma" else "fma")
    gen = torch.Generator().manual_seed(11)
    pts = torch.rand(20000, 3, generator=gen) * 1.9 - 0.95
    sig = ops.density_forward(spec_h, dv(params), pts=dv(pts)).cpu()
    ref, ref_other = NW.density(spec_o, params, pts), NW.density(other_o, params, pts)
    scale = float(ref.abs().max())
    same, cross = float((sig - ref).abs().max()) / scale, float((sig - ref_other).abs().max()) / scale
    print(f"pos_rounding={mode}: kernel vs oracle (same convention) {same:.2e}, vs the other convention {cross:.2e}")
    assert same < 2e-6
    assert cross > 10 * same            # the switch is observable: the conventions are not interchangeable at fp32 parity


def test_hash_known_answer_entries_on_device(ops):
    """The kernels' index rules against the hand-computed literals of tests/test_host.py: the table gradient of a single point
    is non-zero exactly at the 8 hand-computed entries of its cell (hashed level 15, dense level 1)."""
    from tests.test_host import LEVEL15_CELL, LEVEL15_ENTRIES, LEVEL1_CELL, LEVEL1_ENTRIES
    spec_o, spec_h, params = _default_pair()
    for level, cell, want in ((15, LEVEL15_CELL, LEVEL15_ENTRIES), (1, LEVEL1_CELL, LEVEL1_ENTRIES)):
        lv = spec_o.levels[level]
        x_unit = torch.tensor([[(c + 0.25 - 0.5) / lv.scale for c in cell]], dtype=torch.float32)
        pts = x_unit * 2 - 1                                         # world cube; (pts + 1) / 2 is exact here
        assert torch.equal((pts + 1) / 2, x_unit)
        grad = torch.zeros(int(spec_h.n_params), device=DEV)
        ops.density_backward(spec_h, dv(params), torch.ones(1, device=DEV), grad, pts=dv(pts))
        tab = grad[spec_h.n_mlp_params:].reshape(-1, 2).cpu()
        lo, hi = lv.offset, lv.offset + lv.size
        touched = sorted(set((tab[lo:hi].abs().sum(1).nonzero().flatten()).tolist()))
        assert touched == want, (level, touched)


def test_fp16_mode_config5_4096x256(ops):
    """BASELINE configs[4]: 4096 rays x 256 samples, fp16 MLP on MFMA.  sigma, rendered depth and all gradients of the fp16
    mode against (a) the oracle with the same storage rounding (kernel arithmetic: fp32-accumulating MFMA, per-tile scaled
    fp16 dZ) and (b) the fp32 definition (error budget of the storage types themselves).  The oracle runs on a 96-ray subset
    (d_sigma is zero elsewhere, so the gradients only depend on those rays); the launch is full size."""
    from loner_amd import hip
    from loner_amd.utils import synthetic as SY
    o16, h16, params = _default_pair("fp16")
    o32, h32, _ = _default_pair("fp32")
    N, S, SUB = 4096, 256, 96
    gen = torch.Generator().manual_seed(5)
    dirs, _ = SY.lidar_pattern()
    scale, shift = SY.world_cube()
    T = OP.transform_from_pose6(SY.trajectory_pose6(2)[1])
    dist = SY.scene_ranges(dirs, T)
    idx = torch.randint(0, dirs.shape[1], (N,), generator=gen)
    rays, depths, keep = ops.build_lidar_rays(dv(dirs), dv(dist), idx.to(DEV), dv(T[:3, :4].reshape(12)), (1.0, 50.0), scale, shift)
    assert int(keep.sum()) == N
    z = ops.sample_rays_occ(rays, dv(torch.randn(100, 100, 100, generator=gen)), S, 1.0, seed=3)
    P = dv(params)
    sig16 = ops.density_forward(h16, P, rays=rays, z=z)
    sub = torch.randperm(N, generator=gen)[:SUB]
    d_sigma = torch.zeros(N, S)
    d_sigma[sub] = torch.randn(SUB, S, generator=gen) * torch.logspace(-9, 0, SUB)[:, None]    # nine decades of magnitudes: no underflow
    d_sigma[sub[:8], ::3] = 0.0
    g16 = torch.zeros(int(h16.n_params), device=DEV)
    d_rays16 = torch.zeros(N, 13, device=DEV)
    ops.density_backward(h16, P, dv(d_sigma), g16, rays=rays, z=z, reuse_features=True, d_rays=d_rays16)
    depth16 = ops.render_forward(sig16, z, rays)[0]
    sig32 = ops.density_forward(h32, P, rays=rays, z=z)
    depth32 = ops.render_forward(sig32, z, rays)[0]
    # oracle on the subset
    r_s, z_s = rays[sub.to(DEV)].cpu(), z[sub.to(DEV)].cpu()
    res = {}
    for name, so in (("fp16", o16), ("fp32", o32)):
        p = params.clone().requires_grad_(True)
        rr = r_s.clone().requires_grad_(True)
        pts = rr[:, None, 0:3] + rr[:, None, 3:6] * z_s[:, :, None]
        sg = NW.density(so, p, pts.reshape(-1, 3)).reshape(SUB, S)
        (sg * d_sigma[sub]).sum().backward()
        res[name] = (sg.detach(), p.grad, rr.grad)
    s_scale = float(res["fp32"][0].abs().max())
    e_sig_model = float((sig16[sub.to(DEV)].cpu() - res["fp16"][0]).abs().max()) / s_scale
    e_sig_fp32 = float((sig16[sub.to(DEV)].cpu() - res["fp32"][0]).abs().max()) / s_scale
    e_gp_model, e_gp_fp32 = rel(g16, res["fp16"][1]), rel(g16, res["fp32"][1])
    e_gr_model, e_gr_fp32 = rel(d_rays16[sub.to(DEV)][:, :6], res["fp16"][2][:, :6]), rel(d_rays16[sub.to(DEV)][:, :6], res["fp32"][2][:, :6])
    e_depth = float(((depth16 - depth32).abs() / depth32.abs().clamp(min=1e-6)).max())
    print(f"fp16 mode 4096x256: sigma vs fp16-model {e_sig_model:.2e}, vs fp32 {e_sig_fp32:.2e}; dparams {e_gp_model:.2e} / {e_gp_fp32:.2e}; "
          f"drays {e_gr_model:.2e} / {e_gr_fp32:.2e}; rendered depth fp16-vs-fp32 rel {e_depth:.2e}")
    # (a) kernel vs the same storage model: forward is fp32-accumulated either way -> fp32 noise; backward adds the fp16 rounding
    # of dZ (2^-11 relative per element, averaged down by the sums it enters)
    # (sigma: the encoded features are rounded to fp16, so an fp32 feature one ulp off the oracle's - the forward kernel accumulates
    # the eight corner terms with fmaf like tiny-cuda-nn, on the fine levels as two chains of four; the oracle multiplies and adds -
    # lands on the neighbouring fp16 value once in ~2^13 features: 2^-11 of one feature's contribution, ~1e-4 of max |sigma| in the
    # worst sample of 25 k; one sample in ~256 holds a flipped feature.  Hence: fp32 noise for the bulk, the one-flip bound for the worst.)
    err = (sig16[sub.to(DEV)].cpu() - res["fp16"][0]).abs().flatten() / s_scale
    print(f"  sigma error quantiles 0.9 / 0.98 / 0.999 / max: {float(torch.quantile(err, 0.9)):.1e} {float(torch.quantile(err, 0.98)):.1e} "
          f"{float(torch.quantile(err, 0.999)):.1e} {float(err.max()):.1e}")
    assert float(torch.quantile(err, 0.9)) < 2e-6 and e_sig_model < 5e-4
    assert e_gp_model < 1e-3 and e_gr_model < 2e-3
    e_layers = rel_layers(o16, g16, res["fp16"][1])
    print(f"  worst matrix of the MLP gradient vs the fp16 model: {e_layers:.2e}")
    assert e_layers < 1e-2
    # (b) storage error of fp16 features and weights: 2^-11 per rounded operand; sigma sums 32 + 64 rounded products.  The ray
    # gradient is the most sensitive output: d/dx multiplies each level's d_feature (perturbed by ~5e-4) with differences of
    # neighbouring table entries times the level scale (up to 5e5) - large terms of both signs (measured 1.4e-2; the same
    # storage model on the CPU, (a), agrees with the kernel to 6e-4, so this is the price of fp16 storage, not of the kernel)
    assert e_sig_fp32 < 5e-3 and e_gp_fp32 < 1e-2 and e_gr_fp32 < 5e-2
    # north_star: depth outputs within 1e-4 relative of the (fp16) reference path - here fp16 mode vs the fp32 definition
    assert e_depth < 1e-3
    # gradients of rays without upstream gradient stay exactly zero, tiny d_sigma rows survive (per-tile power-of-two scaling)
    mask = torch.ones(N, dtype=torch.bool); mask[sub] = False
    assert float(d_rays16[mask.to(DEV)].abs().max()) == 0.0
    tiny = sub[:8].to(DEV)
    assert float(d_rays16[tiny][:, :6].abs().max()) > 0.0


@pytest.mark.parametrize("name", ["hash_f4_2hidden", "hash_f8", "freq_siren", "freq_relu128", "small_hash", "freq_wide256", "freq_relu3",
                                  "freq_tanh2", "freq_sine16", "freq12_small", "freq16_h16", "freq_tanh128", "freq_wide256x2", "hash_wide256x3"])
def test_fp16_mode_general_networks(ops, name):
    """precision fp16 beyond the reference's default shape: frequency encoding + SIREN / wide ReLU MLPs, several hidden layers,
    4 or 8 features per level - forward and every gradient against the oracle with the same storage rounding (fp16 features,
    weights and inter-layer activations; fp32 accumulation)."""
    from loner_amd import hip
    enc, net = NETS[name]
    net16 = dict(net, precision="fp16")
    spec_o, spec_h = NW.NetworkSpec.from_config(enc, net16), hip.make_net_spec(enc, net16)
    params = NW.init_params(spec_o, 3)
    if spec_o.n_enc_params:
        params[spec_o.n_mlp_params:] *= 3000.0
    gen = torch.Generator().manual_seed(8)
    n = 1000                                                       # not a multiple of the 32-sample tile
    pts = torch.rand(n, 3, generator=gen) * 1.9 - 0.95
    d_sigma = torch.randn(n, generator=gen) * torch.logspace(-6, 0, n)
    d_sigma[torch.rand(n, generator=gen) < 0.2] = 0.0
    sig = ops.density_forward(spec_h, dv(params), pts=dv(pts)).cpu()
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    d_pts = ops.density_backward(spec_h, dv(params), dv(d_sigma), grad, pts=dv(pts), want_d_pts=True)
    p = params.clone().requires_grad_(True)
    x = pts.clone().requires_grad_(True)
    ref = NW.density(spec_o, p, x)
    (ref * d_sigma).sum().backward()
    scale = float(ref.detach().abs().max())
    e_s = float((sig - ref.detach()).abs().max()) / scale
    nm = spec_o.n_mlp_params
    e_w, e_x = rel(grad[:nm], p.grad[:nm]), rel(d_pts, x.grad)
    e_t = rel(grad[nm:], p.grad[nm:]) if spec_o.n_enc_params else 0.0
    print(f"fp16 {name}: sigma {e_s:.2e}  dW {e_w:.2e}  dtable {e_t:.2e}  dpts {e_x:.2e}")
    # forward: identical storage rounding, fp32 accumulation on both sides; an activation that lands on a rounding boundary of fp16 may
    # round differently after different summation orders (2^-11 of that activation), hence not 1e-6.  backward: + fp16 rounding of dZ.
    assert e_s < 2e-3 and e_w < 3e-3 and e_t < 3e-3 and e_x < 5e-3
    e_l = rel_layers(spec_o, grad, p.grad)
    print(f"fp16 {name}: worst matrix of the MLP gradient {e_l:.2e}")
    assert e_l < 3e-2
    # linearity in d_sigma (exact power-of-two scale) and frozen parameters
    g2 = torch.zeros_like(grad)
    ops.density_backward(spec_h, dv(params), dv(d_sigma * 4.0), g2, pts=dv(pts))
    assert rel(g2, 4.0 * grad) < 1e-6
    assert torch.equal(ops.density_backward(spec_h, dv(params), dv(d_sigma), None, pts=dv(pts), want_d_pts=True), d_pts)


@pytest.mark.parametrize("name,S,live", [("freq_relu128", 64, 37), ("freq_relu128", 20, 50), ("freq12_small", 96, 50), ("freq_relu3", 33, 41)])
def test_fused_frequency_rays_form_equals_points_form(ops, name, S, live):
    """The fused frequency kernels take their points from rays + depths (o + d z, rounded as the reference rounds it) through three
    request routes: one ray per wave step (S a multiple of 32: its index from a shift / one scalar division, the record loaded once),
    a ray per lane (any S), and explicit points.  All three are the same function of the same points, bit for bit - forward, the
    weight gradient and the input gradient - with a device-side live-ray count that leaves the last tile ragged."""
    from loner_amd import hip
    enc, net = NETS[name]
    net16 = dict(net, precision="fp16")
    spec_o, spec_h = NW.NetworkSpec.from_config(enc, net16), hip.make_net_spec(enc, net16)
    params = dv(NW.init_params(spec_o, 4))
    gen = torch.Generator().manual_seed(21)
    n = 50
    rays = torch.zeros(n, 13); rays[:, 0:3] = torch.rand(n, 3, generator=gen) * 0.3 - 0.15
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1); rays[:, 11] = 0.02; rays[:, 12] = 0.6
    z = torch.sort(torch.rand(n, S, generator=gen) * 0.55 + 0.02, dim=1).values
    d_sigma = torch.randn(n, S, generator=gen) * torch.logspace(-4, 0, n)[:, None]
    R, Z, DS = dv(rays), dv(z), dv(d_sigma)
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    pts = (R[:live, None, 0:3] + R[:live, None, 3:6] * Z[:live, :, None]).reshape(-1, 3).contiguous()    # (separate multiply and add: no fma)
    s_rays = ops.density_forward(spec_h, params, rays=R, z=Z, n_rays_dev=n_dev)
    s_pts = ops.density_forward(spec_h, params, pts=pts)
    assert torch.equal(s_rays.reshape(n, S)[:live].reshape(-1), s_pts.reshape(-1)) and float(s_pts.abs().max()) > 0
    g_rays, g_pts = torch.zeros(int(spec_h.n_params), device=DEV), torch.zeros(int(spec_h.n_params), device=DEV)
    p_rays = ops.density_backward(spec_h, params, DS, g_rays, rays=R, z=Z, n_rays_dev=n_dev, want_d_pts=True)
    p_pts = ops.density_backward(spec_h, params, DS[:live].reshape(-1).contiguous(), g_pts, pts=pts, want_d_pts=True)
    assert torch.equal(g_rays, g_pts) and float(g_rays.abs().max()) > 0
    assert torch.equal(p_rays.reshape(n * S, 3)[:live * S], p_pts.reshape(-1, 3))
    # and against the oracle (same storage rounding) on the live samples
    ref = NW.density(spec_o, params.cpu(), pts.cpu())
    assert float((s_pts.cpu().reshape(-1) - ref).abs().max()) < 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("name", ["freq_relu128", "hash_f4_2hidden", "freq_wide256"])
def test_fp16_mlp_weight_fill_with_a_parameter_vector_that_is_not_16_byte_aligned(ops, name):
    """The fp16 MLP kernels read the network into LDS with 8- and 16-byte loads when the parameter vector allows it and element by
    element when it does not (a view that starts 4 bytes into an allocation): the same network either way, bit for bit."""
    from loner_amd import hip
    enc, net = NETS[name]
    net16 = dict(net, precision="fp16")
    spec_o, spec_h = NW.NetworkSpec.from_config(enc, net16), hip.make_net_spec(enc, net16)
    params = dv(NW.init_params(spec_o, 6))
    shifted = torch.empty(params.numel() + 1, device=DEV)[1:]
    shifted.copy_(params)
    assert params.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    gen = torch.Generator().manual_seed(2)
    n = 777
    pts = dv(torch.rand(n, 3, generator=gen) * 1.9 - 0.95)
    d_sigma = dv(torch.randn(n, generator=gen))
    assert torch.equal(ops.density_forward(spec_h, params, pts=pts), ops.density_forward(spec_h, shifted, pts=pts))
    g_a, g_s = torch.zeros(int(spec_h.n_params), device=DEV), torch.zeros(int(spec_h.n_params), device=DEV)
    p_a = ops.density_backward(spec_h, params, d_sigma, g_a, pts=pts, want_d_pts=True)
    p_s = ops.density_backward(spec_h, shifted, d_sigma, g_s, pts=pts, want_d_pts=True)
    assert torch.equal(g_a, g_s) and torch.equal(p_a, p_s) and float(g_a.abs().max()) > 0


def test_fp16_mode_refuses_what_it_does_not_cover(ops):
    from loner_amd import hip
    many_inputs = (dict(otype="Frequency", n_frequencies=24), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2))
    for enc, net in (NETS["hash_f1"], many_inputs):               # odd feature count per level; 144 inputs into a 128-wide network
        bad = hip.make_net_spec(enc, dict(net, precision="fp16"))
        with pytest.raises(RuntimeError, match="fp16"):
            ops.density_forward(bad, torch.zeros(int(bad.n_params), device=DEV), pts=torch.zeros(64, 3, device=DEV))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_density_backward_with_frozen_parameters(ops, prec):
    """grad_params=None (tracking phase, optimizer.py:239-259): only the input gradient, identical to the one of a full backward."""
    _, spec_h, params = _default_pair(prec)
    gen = torch.Generator().manual_seed(9)
    n, S = 24, 128
    rays = torch.zeros(n, 13); rays[:, 0:3] = torch.rand(n, 3, generator=gen) * 0.2 - 0.1
    d = torch.randn(n, 3, generator=gen); rays[:, 3:6] = d / d.norm(dim=1, keepdim=True); rays[:, 11] = 0.02; rays[:, 12] = 0.55
    z = torch.sort(torch.rand(n, S, generator=gen) * 0.5 + 0.02, dim=1).values
    ds = torch.randn(n, S, generator=gen)
    P, R, Z = dv(params), dv(rays), dv(z)
    ops.density_forward(spec_h, P, rays=R, z=Z)
    full, frozen = torch.zeros(n, 13, device=DEV), torch.zeros(n, 13, device=DEV)
    grad = torch.zeros(int(spec_h.n_params), device=DEV)
    ops.density_backward(spec_h, P, dv(ds), grad, rays=R, z=Z, reuse_features=True, d_rays=full)
    ops.density_backward(spec_h, P, dv(ds), None, rays=R, z=Z, reuse_features=True, d_rays=frozen)
    assert torch.equal(full, frozen) and float(full.abs().max()) > 0
    p_full = ops.density_backward(spec_h, P, dv(ds), grad, rays=R, z=Z, want_d_pts=True)
    p_frozen = ops.density_backward(spec_h, P, dv(ds), None, rays=R, z=Z, want_d_pts=True)
    assert torch.equal(p_full, p_frozen)
    with pytest.raises(RuntimeError):
        ops.density_backward(spec_h, P, dv(ds), None, rays=R, z=Z)


@pytest.mark.parametrize("net", ["default", "hash_f4_2hidden", "hash_f1", "hash_f8", "small_hash"])
def test_record_partition_equals_global_accumulation(ops, net):
    """Every route a table-gradient contribution can take - 8-byte pair records, 12-byte x-pair records, run-length combined
    records, the split reduce of dense-indexed levels, region overflow - ends in the same 64-bit fixed-point sum as the
    LNR_BWD_TABLE_ATOMICS route (every contribution straight into the overflow accumulators): the gradients are EQUAL, bit for
    bit, at a size where several reduce workgroups share an owner (76 800 samples), with a ragged live-ray count and samples
    bunched on few cells (the regions of the coarse levels overflow)."""
    from loner_amd import hip
    enc, net_cfg = NETS[net]
    spec = hip.make_net_spec(enc, net_cfg)
    gen = torch.Generator().manual_seed(11)
    params = dv(NW.init_params(NW.NetworkSpec.from_config(enc, net_cfg), 3))
    params[spec.n_mlp_params:] *= 500
    N, S, live = 300, 256, 277
    o = torch.rand(N, 3, generator=gen) * 0.2 - 0.1
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=1)
    rays = torch.zeros(N, 13); rays[:, 0:3] = o; rays[:, 3:6] = d; rays[:, 6:9] = -d; rays[:, 11] = 0.01; rays[:, 12] = 0.8
    z = torch.sort(torch.rand(N, S, generator=gen) ** 3 * 0.75 + 0.01, dim=1).values          # bunched near the origin
    d_sigma = torch.randn(N, S, generator=gen) * torch.logspace(-3, 2, N)[:, None]
    n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
    ops.density_forward(spec, params, rays=dv(rays), z=dv(z), n_rays_dev=n_dev)
    g_rec = torch.zeros(int(spec.n_params), device=DEV); g_acc = torch.zeros_like(g_rec)
    r_rec = torch.zeros(N, 13, device=DEV); r_acc = torch.zeros_like(r_rec)
    ops.density_backward(spec, params, dv(d_sigma), g_rec, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=r_rec)
    ops.density_backward(spec, params, dv(d_sigma), g_acc, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=r_acc, table_atomics=True)
    table = slice(int(spec.n_mlp_params), int(spec.n_params))
    assert float(g_rec[table].abs().max()) > 0
    print(net, "entries touched", int((g_rec[table] != 0).sum()), "differing", int((g_rec[table] != g_acc[table]).sum()))
    assert torch.equal(g_rec[table], g_acc[table])
    assert torch.equal(g_rec[:table.start], g_acc[:table.start]) and torch.equal(r_rec, r_acc)
    assert float(r_rec[live:].abs().max()) == 0.0                                  # dropped rays receive nothing
    # the binned partition of the hashed levels (LNR_BWD_BINS, both workgroup sizes) is the same function too
    for w8 in (False, True):
        g_bin = torch.zeros_like(g_rec); r_bin = torch.zeros_like(r_rec)
        ops.density_backward(spec, params, dv(d_sigma), g_bin, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=r_bin, bins=True, bins_w8=w8)
        assert torch.equal(g_bin, g_rec) and torch.equal(r_bin, r_rec)
    # [r4] the 64-bit overflow accumulators are kept all-zero BETWEEN calls (the reduce re-zeroes what it reads; no per-call clear): a call
    # behind one that filled every level's accumulators (the all-atomics route) must see none of it
    g_acc2 = torch.zeros_like(g_rec)
    ops.density_backward(spec, params, dv(d_sigma), g_acc2, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=torch.zeros_like(r_rec), table_atomics=True)
    g_again = torch.zeros_like(g_rec); r_again = torch.zeros_like(r_rec)
    ops.density_backward(spec, params, dv(d_sigma), g_again, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=r_again)
    assert torch.equal(g_again, g_rec) and torch.equal(r_again, r_rec) and torch.equal(g_acc2, g_rec)
    # [r4] LNR_BWD_OVERWRITE_GRAD: the gradient is STORED (every float of the table and of the MLP matrices), whatever the buffer held
    g_over = torch.full_like(g_rec, 7.0)
    ops.density_backward(spec, params, dv(d_sigma), g_over, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, d_rays=torch.zeros_like(r_rec), overwrite_grad=True)
    assert torch.equal(g_over, g_rec)


@pytest.mark.parametrize("net", ["default", "hash_f4_2hidden"])
def test_binned_partition_overflow_and_region_close(ops, net):
    """The binned partition under the worst skew: every sample of the batch sits in the same few cells, so a handful of owners
    receive everything - their LDS bins overflow in every batch and their regions fill up and close - while the rest stay empty.
    Whatever route a record takes (bin -> line -> region, bin tail at the kernel's end, bin-full overflow, closed-region overflow),
    the gradient equals the all-atomics accumulation bit for bit."""
    from loner_amd import hip
    enc, net_cfg = NETS[net]
    spec = hip.make_net_spec(enc, net_cfg)
    gen = torch.Generator().manual_seed(12)
    params = dv(NW.init_params(NW.NetworkSpec.from_config(enc, net_cfg), 4))
    params[spec.n_mlp_params:] *= 500
    N, S = 256, 128
    rays = torch.zeros(N, 13)
    rays[:, 0:3] = torch.tensor([0.05, -0.02, 0.01]); rays[:, 3:6] = torch.tensor([0.6, 0.0, 0.8]); rays[:, 6:9] = -rays[:, 3:6]
    rays[:, 11] = 0.01; rays[:, 12] = 0.8
    # three clusters of samples 1e-6 apart (same cells on every level), a few stragglers elsewhere
    z = torch.full((N, S), 0.2) + torch.rand(N, S, generator=gen) * 1e-6
    z[:, 40:80] += 0.1; z[:, 80:] += 0.25
    z[::17, ::9] = torch.rand(len(range(0, N, 17)), len(range(0, S, 9)), generator=gen) * 0.7 + 0.02
    z = torch.sort(z, dim=1).values
    d_sigma = torch.randn(N, S, generator=gen)
    for n_live in (N, 3):                                                          # also a batch with three live rays (one partial batch)
        n_dev = torch.tensor([n_live], dtype=torch.int32, device=DEV)
        ops.density_forward(spec, params, rays=dv(rays), z=dv(z), n_rays_dev=n_dev)
        g_bin = torch.zeros(int(spec.n_params), device=DEV); g_acc = torch.zeros_like(g_bin); g_scan = torch.zeros_like(g_bin)
        ops.density_backward(spec, params, dv(d_sigma), g_bin, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, bins=True)
        ops.density_backward(spec, params, dv(d_sigma), g_acc, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, table_atomics=True)
        ops.density_backward(spec, params, dv(d_sigma), g_scan, rays=dv(rays), z=dv(z), n_rays_dev=n_dev)
        assert float(g_bin.abs().max()) > 0
        assert torch.equal(g_bin, g_acc), int((g_bin != g_acc).sum())
        assert torch.equal(g_bin, g_scan)
        g_w8 = torch.zeros_like(g_bin)
        ops.density_backward(spec, params, dv(d_sigma), g_w8, rays=dv(rays), z=dv(z), n_rays_dev=n_dev, bins=True, bins_w8=True)
        assert torch.equal(g_bin, g_w8)


# ------------------------------------------------------------------------------------------- full-size properties
def test_full_size_iteration_properties(ops):
    """BASELINE size (4096 rays x 512 samples, default network): size-independent properties of every stage, and the
    record-partition table gradient against the independent global-atomic accumulation path."""
    import os
    from loner_amd import hip
    from loner_amd.utils import synthetic as SY
    enc, net = NETS["default"]
    spec = hip.make_net_spec(enc, net)
    gen = torch.Generator().manual_seed(0)
    params = dv(NW.init_params(NW.NetworkSpec.from_config(enc, net), 0))
    params[spec.n_mlp_params:] *= 2000
    N, S = 4096, 512
    dirs, _ = SY.lidar_pattern()
    scale, shift = SY.world_cube()
    T = OP.transform_from_pose6(SY.trajectory_pose6(2)[1])
    dist = SY.scene_ranges(dirs, T)
    idx = torch.randint(0, dirs.shape[1], (N,), generator=gen)
    rays, depths, keep = ops.build_lidar_rays(dv(dirs), dv(dist), idx.to(DEV), dv(T[:3, :4].reshape(12)), (1.0, 50.0), scale, shift)
    assert int(keep.sum()) == N
    grid = dv(torch.randn(100, 100, 100, generator=gen))
    z = ops.sample_rays_occ(rays, grid, S, 1.0, seed=3)
    assert bool((z[:, 1:] >= z[:, :-1]).all()) and bool((z >= rays[:, 11:12]).all()) and bool((z <= rays[:, 12:13]).all())
    sigma = ops.density_forward(spec, params, rays=rays, z=z)
    assert torch.isfinite(sigma).all()
    # values at full size against the oracle on a 64-ray subset (the launch is the full 4096 x 512; the oracle only visits the subset)
    sub = torch.randperm(N, generator=gen)[:64]
    spec_o = NW.NetworkSpec.from_config(enc, net)
    pts_sub = (rays[sub.to(DEV), None, 0:3] + rays[sub.to(DEV), None, 3:6] * z[sub.to(DEV), :, None]).cpu()
    ref_sub = NW.density(spec_o, params.cpu(), pts_sub.reshape(-1, 3)).reshape(64, S)
    assert rel(sigma[sub.to(DEV)], ref_sub) < 1e-5
    depth, w, opac, var = ops.render_forward(sigma, z, rays, noise_std=1.0, seed=4)
    assert bool((w >= 0).all()) and float(opac.max()) <= 1.0 + 1e-5 and bool((var >= 0).all())
    assert bool((depth >= rays[:, 11] - 1e-6).all()) and bool((depth <= rays[:, 12] + 1e-6).all())    # convex combination of z and far
    counts = ops.count_opaque(rays, depths)
    cfg = hip.LossConfig(selection=0, min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0, depth_lambda=0.005, min_eps=0.5, fixed_eps=3.0)
    loss, d_sigma, d_rays, stats, _ = ops.los_loss_fused(sigma, z, rays, depths, scale, cfg, counts, noise_std=1.0, seed=4, want_stats=True)
    assert torch.isfinite(loss).all() and abs(float(loss[0]) - float(loss[1] + loss[2] + loss[3])) < 1e-3 * float(loss[0])
    assert rel(stats[:, 0], depth) < 1e-6                                     # same rendering code, same noise stream
    # linearity of the backward in d_sigma, and partition path == atomic path
    g1 = torch.zeros(int(spec.n_params), device=DEV); g2 = torch.zeros_like(g1); g3 = torch.zeros_like(g1)
    p1 = ops.density_backward(spec, params, d_sigma, g1, rays=rays, z=z, want_d_pts=True)
    ops.density_backward(spec, params, 2.0 * d_sigma, g2, rays=rays, z=z, want_d_pts=False)
    assert rel(g2, 2.0 * g1) < 1e-5
    p3 = ops.density_backward(spec, params, d_sigma, g3, rays=rays, z=z, want_d_pts=True, table_atomics=True)
    # the full-size gradient against the oracle: d_sigma restricted to the subset rays, so that the oracle only needs those
    ds_sub = torch.zeros_like(d_sigma); ds_sub[sub.to(DEV)] = d_sigma[sub.to(DEV)]
    g_sub = torch.zeros_like(g1)
    ops.density_backward(spec, params, ds_sub, g_sub, rays=rays, z=z)
    p_o = params.cpu().clone().requires_grad_(True)
    (NW.density(spec_o, p_o, pts_sub.reshape(-1, 3)).reshape(64, S) * d_sigma[sub.to(DEV)].cpu()).sum().backward()
    print("full-size launch, subset gradient vs oracle: rel", rel(g_sub, p_o.grad))
    assert rel(g_sub, p_o.grad) < 5e-5
    print("partition vs atomic path: table grad rel", rel(g1, g3), " checksum", float(g1.double().sum()), float(g3.double().sum()))
    assert rel(g1, g3) < 1e-5 and torch.equal(p1, p3)
    assert abs(float(g1.double().sum()) - float(g3.double().sum())) < 1e-6 * float(g1.double().abs().sum())
    # run-to-run reproducibility: weight gradient (ordered slab reduction), input gradient and table gradient (64-bit fixed-point
    # sums, incl. the overflow accumulators of the coherent levels) are bit-identical; only a statistically skewed hashed level
    # could still spill into float atomics
    g1b = torch.zeros_like(g1)
    p1b = ops.density_backward(spec, params, d_sigma, g1b, rays=rays, z=z, want_d_pts=True)
    print("reproducibility: differing gradient entries", int((g1b != g1).sum()))
    assert torch.equal(g1b, g1) and torch.equal(p1b, p1)
    # the training loop's route for the pose gradient: per-ray reduction inside the backward == d_pts -> lnr_points_grad_to_rays
    ref_rays = d_rays.clone(); ops.points_grad_to_rays(p1, z, ref_rays)
    out_rays = d_rays.clone(); g4 = torch.zeros_like(g1)
    ops.density_backward(spec, params, d_sigma, g4, rays=rays, z=z, d_rays=out_rays)
    assert rel(g4, g1) < 1e-6 and rel(out_rays[:, :6], ref_rays[:, :6]) < 2e-5 and torch.equal(out_rays[:, 6:], d_rays[:, 6:])
    out_rays2 = d_rays.clone(); g5 = torch.zeros_like(g1)
    ops.density_backward(spec, params, d_sigma, g5, rays=rays, z=z, d_rays=out_rays2)
    assert torch.equal(out_rays2, out_rays) and torch.equal(g5, g4)          # fixed-point ray sums: reproducible as well


# ------------------------------------------------------------------------------------------- failure guard (poison word)
def test_failure_guard_kernels(ops, golden):
    """include/loner_hip.h "Failure guard": the loss reduce marks a NaN total, pose_backward a non-finite pose gradient / pose,
    first event wins with the caller's tag; lnr_adam_step and lnr_occ_grid_apply do nothing once the word is set."""
    from loner_amd import hip
    g, spec_o, spec_h = _g8_setup(golden)
    rays, z, depths, params = dv(g["rays"]), dv(g["z"]), dv(g["depths"]), dv(g["params"])
    cfg = hip.LossConfig(selection=0, min_js=1.0, max_js=10.0, js_alpha=1.0, los_lambda=1000.0, depth_lambda=0.005, min_eps=0.5, fixed_eps=3.0)
    counts = ops.count_opaque(rays, depths)
    sigma = ops.density_forward(spec_h, params, rays=rays, z=z)
    poison = torch.zeros(2, device=DEV, dtype=torch.int32)
    loss = ops.los_loss_fused(sigma, z, rays, depths, float(g["scale"]), cfg, counts, noise_std=0.0, poison=poison, poison_tag=7)[0]
    assert np.isfinite(float(loss[0])) and poison.cpu().tolist() == [0, 0]
    # (a NaN density is harmless: relu(sigma + noise) maps it to 0, in the reference's F.relu on the GPU as here; a NaN sample depth is not)
    bad = z.clone(); bad[3, 5] = float("nan")
    loss = ops.los_loss_fused(sigma, bad, rays, depths, float(g["scale"]), cfg, counts, noise_std=0.0, poison=poison, poison_tag=7)[0]
    assert np.isnan(float(loss[0])) and poison.cpu().tolist() == [hip.POISON_NAN_LOSS, 7]
    # first event wins
    p6 = torch.tensor([[0.1, 0.2, 0.3, 0.01, -0.02, 0.03], [0.0, 0.0, 0.0, 0.3, 0.1, -0.2]], device=DEV)
    dT = torch.randn(2, 12, device=DEV); dT[1, 4] = float("inf")
    ops.pose_backward(p6, dT, poison=poison, poison_tag=9)
    assert poison.cpu().tolist() == [hip.POISON_NAN_LOSS, 7]
    # pose gradient / pose checks on a fresh word; a fixed pose's gradient is exactly zero even if its cotangent is not finite
    poison.zero_()
    mask = torch.tensor([1, 0], device=DEV, dtype=torch.uint8)
    d = ops.pose_backward(p6, dT, mask=mask, poison=poison, poison_tag=2)
    assert poison.cpu().tolist() == [0, 0] and float(d[1].abs().max()) == 0.0 and torch.isfinite(d).all()
    ops.pose_backward(p6, dT, poison=poison, poison_tag=3)
    assert poison.cpu().tolist() == [hip.POISON_POSE_GRAD, 3]
    poison.zero_()
    p_bad = p6.clone(); p_bad[0, 1] = float("nan")
    ops.pose_backward(p_bad, torch.zeros(2, 12, device=DEV), mask=torch.tensor([0, 1], device=DEV, dtype=torch.uint8), poison=poison, poison_tag=4)
    assert poison.cpu().tolist() == [hip.POISON_POSE, 4]
    # the step kernels obey the word
    n = 1003
    p, gr, m, v = torch.randn(n, device=DEV), torch.randn(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p0, g0 = p.clone(), gr.clone()
    ops.adam_step(p, gr, m, v, 0.01, 1, zero_grad=True, poison=poison)
    assert torch.equal(p, p0) and torch.equal(gr, g0) and float(m.abs().max()) == 0.0
    grid, buf = torch.randn(8, 8, 8, device=DEV), torch.randint(-1000, 1000, (512,), device=DEV, dtype=torch.int64) << 30
    grid0, buf0 = grid.clone(), buf.clone()
    ops.occ_grid_apply(grid, buf, 1e-2, poison=poison)
    assert torch.equal(grid, grid0) and torch.equal(buf, buf0)
    poison.zero_()
    ops.adam_step(p, gr, m, v, 0.01, 1, zero_grad=True, poison=poison)
    ops.occ_grid_apply(grid, buf, 1e-2, poison=poison)
    assert not torch.equal(p, p0) and not torch.equal(grid, grid0)
