"""CPU-side checks: the C-ABI library loads and exports every declared symbol, host logic (network spec
derivation, settings, pose maths, schedules) and the no-fallback rule."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as ge
    ge.build()
    from loner_amd import hip
    return hip.LIB_PATH


def test_library_exports_every_symbol_in_the_header(lib_path):
    header = open(os.path.join(ROOT, "include", "loner_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lnr_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/loner_hip.h but not exported"
    from loner_amd import hip
    assert sorted(hip.declared_symbols()) == declared            # the ctypes table binds exactly the header
    assert hip.load().lnr_version() >= 100


def test_net_spec_matches_oracle_level_geometry(lib_path):
    from loner_amd import hip
    from oracle import network as NW
    cases = [
        (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16), dict(n_neurons=64, n_hidden_layers=1)),
        (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16), dict(n_neurons=64, n_hidden_layers=4)),
        (dict(otype="HashGrid", n_levels=7, n_features_per_level=4, log2_hashmap_size=15, base_resolution=5, per_level_scale=1.38), dict(n_neurons=128, n_hidden_layers=2)),
        (dict(otype="Frequency", n_frequencies=12), dict(n_neurons=256, n_hidden_layers=1, activation="Sine")),
    ]
    for enc, net in cases:
        h = hip.make_net_spec(enc, net)
        o = NW.NetworkSpec.from_config(enc, net)
        assert (h.enc_dim, h.in_dim, h.n_mlp_params, int(h.n_params)) == (o.enc_dim, o.in_dim, o.n_mlp_params, o.n_params)
        for l, lv in enumerate(o.levels):
            assert (h.level_res[l], h.level_size[l], h.level_offset[l], h.level_hashed[l]) == (lv.res, lv.size, lv.offset, int(lv.hashed))
            assert abs(h.level_scale[l] - lv.scale) <= 2e-7 * lv.scale      # exp2f/log2f: libm vs numpy, <= 1 ulp
    # the default density network of the reference: 3072 MLP weights + 3 706 880 x 2 table entries
    h = hip.make_net_spec(*cases[0])
    assert h.n_mlp_params == 3072 and int(h.n_params) == 7416832
    assert list(h.level_size[:3]) == [4096, 32768, 262144] and list(h.level_hashed[:4]) == [0, 0, 0, 1]


# Hand-computed known answers for the two index rules of the grid encoding (tiny-cuda-nn's published grid_index):
#   hashed: (x ^ y*2654435761 ^ z*805459861) mod 2^32 mod T;  e.g. (1,1,1): 0x9E3779B1 ^ 0x30025795 ^ 1 = 0xAE352E25 = 2922720805
#   dense:  x + y*res + z*res^2
HASH_KA = {(1, 1, 1): 2922720805, (0, 0, 0): 0, (5, 0, 0): 5, (0, 1, 0): 2654435761, (0, 0, 1): 805459861,
           (127, 255, 511): 1307501915, (300000, 1, 2): 4265035131, (300000, 200001, 400002): 3250711227, (524287, 524287, 524287): 2750599643}
LEVEL15_CELL = (300000, 200001, 400002)                      # hash of the cell's own corner: 3250711227
LEVEL15_ENTRIES = [87272, 87273, 125626, 125627, 212092, 212093, 229934, 229935]          # its 8 corners, mod 2^18
LEVEL1_CELL = (13, 14, 15)
LEVEL1_ENTRIES = [15821, 15822, 15853, 15854, 16845, 16846, 16877, 16878]                # res 32: x + 32 y + 1024 z
DEFAULT_SCALES = [16.0 * 2 ** l - 1.0 for l in range(16)]                                 # 15, 31, ..., 524287 (exact in fp32)


def test_hash_and_level_geometry_known_answers(lib_path):
    """The oracle's index rules and level geometry against literals computed by hand (not by the oracle's own code), and the
    oracle's table gradient of single points landing exactly on the hand-computed entries."""
    import torch
    from loner_amd import hip
    from oracle import network as NW
    m = 0xFFFFFFFF
    for (x, y, z), want in HASH_KA.items():
        assert ((x ^ (y * NW.PRIME_Y & m) ^ (z * NW.PRIME_Z & m)) & m) == want
    enc, net = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16), dict(n_neurons=64, n_hidden_layers=1)
    o, h = NW.NetworkSpec.from_config(enc, net), hip.make_net_spec(enc, net)
    assert [lv.scale for lv in o.levels] == DEFAULT_SCALES == [float(h.level_scale[l]) for l in range(16)]
    assert [lv.res for lv in o.levels] == [16 * 2 ** l for l in range(16)]
    assert [lv.offset for lv in o.levels] == [0, 4096, 36864] + [36864 + 262144 * k for k in range(1, 14)]
    # which table entries does one point touch?  (gradient of sum(features) w.r.t. the table = the 8 corner weights)
    for level, cell, want in ((15, LEVEL15_CELL, LEVEL15_ENTRIES), (1, LEVEL1_CELL, LEVEL1_ENTRIES)):
        lv = o.levels[level]
        x = torch.tensor([[(c + 0.25 - 0.5) / lv.scale for c in cell]], dtype=torch.float32)     # pos = cell + 0.25
        table = torch.zeros(o.n_enc_params // 2, 2, requires_grad=True)
        feats = NW.encode_hashgrid(o, table, x)
        feats[:, 2 * level].sum().backward()
        touched = sorted((table.grad[:, 0].nonzero().flatten() - lv.offset).tolist())
        assert touched == want, (level, touched)
        w = table.grad[lv.offset + torch.tensor(want), 0]
        assert abs(float(w.sum()) - 1.0) < 1e-6 and abs(float(w.max()) - 0.75 ** 3) < 1e-3   # weights of frac = 0.25: (3/4)^3 at the cell's own corner


def test_bad_configs_fail_loudly(lib_path):
    from loner_amd import hip
    with pytest.raises(RuntimeError):
        hip.make_net_spec(dict(otype="HashGrid"), dict(n_neurons=48))
    with pytest.raises(RuntimeError):
        hip.make_net_spec(dict(otype="SphericalHarmonics"), dict(n_neurons=64))
    with pytest.raises(RuntimeError):
        hip.make_net_spec(dict(otype="HashGrid", n_features_per_level=3), dict(n_neurons=64))
    with pytest.raises(RuntimeError):
        hip.make_net_spec(dict(otype="HashGrid"), dict(n_neurons=64, output_activation="Sigmoid"))


def test_no_cpu_fallback_and_product_never_imports_the_oracle(lib_path):
    from loner_amd import hip, ops
    spec = hip.make_net_spec(dict(otype="Frequency", n_frequencies=4), dict(n_neurons=16, n_hidden_layers=1))
    with pytest.raises(RuntimeError):
        ops.density_forward(spec, torch.zeros(int(spec.n_params)), pts=torch.zeros(4, 3))
    if not torch.cuda.is_available():
        from loner_amd.common.pose_utils import WorldCube
        from loner_amd.common.settings import default_optimizer_settings
        from loner_amd.mapping.optimizer import Optimizer
        with pytest.raises(RuntimeError):
            Optimizer(default_optimizer_settings(), None, WorldCube(torch.tensor(85.0), torch.zeros(3)), 0)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "loner_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src


def test_missing_library_is_an_error(lib_path, tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from loner_amd import hip\n"
            "hip.LIB_PATH = %r\n"
            "try:\n    hip.load()\nexcept RuntimeError as e:\n    print('RAISED')\n") % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "RAISED" in out.stdout


def test_product_initialiser_equals_the_oracles(lib_path):
    """SigmaNetwork(seed=s) - the product's own initialiser (Xavier-uniform matrices, uniform(-1e-4, 1e-4) tables, tinycudann's
    layout) - reproduces oracle.network.init_params(spec, s) bit for bit, for the default and for a frequency-encoded network: the
    HIP legs of bench.py start from the product's initialiser, the oracle legs from the oracle's, and they are the same tensor."""
    from loner_amd.common.settings import default_nerf_config
    from loner_amd.models.nerf_tcnn import SigmaNetwork
    from oracle import network as NW
    nc = default_nerf_config()
    cases = [(nc["pos_encoding_sigma"], nc["sigma_network"]),
             (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2))]
    for enc, net in cases:
        for seed in (0, 7):
            a = SigmaNetwork(3, 1, enc, net, seed=seed).params.detach()
            b = NW.init_params(NW.NetworkSpec.from_config(dict(enc), dict(net)), seed)
            assert a.dtype == b.dtype and torch.equal(a, b)


def test_pose_maths_matches_oracle_and_scipy():
    from scipy.spatial.transform import Rotation
    from loner_amd.common.pose_utils import axis_angle_to_matrix, matrix_to_axis_angle, tensor_to_transform
    from oracle import poses as OP
    gen = torch.Generator().manual_seed(0)
    aa = torch.randn(50, 3, generator=gen) * 1.2
    aa[:3] *= 1e-9
    R = axis_angle_to_matrix(aa)
    assert torch.equal(R, OP.rotation_from_axis_angle(aa))
    assert np.abs(R.numpy() - Rotation.from_rotvec(aa.double().numpy()).as_matrix()).max() < 1e-6
    for i in range(50):
        back = matrix_to_axis_angle(R[i])
        assert np.abs(axis_angle_to_matrix(back).numpy() - R[i].numpy()).max() < 1e-5
    p = torch.randn(6, generator=gen).requires_grad_(True)
    q = p.detach().clone().requires_grad_(True)
    cot = torch.randn(4, 4, generator=gen)
    (tensor_to_transform(p) * cot).sum().backward()
    (OP.transform_from_pose6(q) * cot).sum().backward()
    assert torch.allclose(p.grad, q.grad, atol=1e-6)
    assert tensor_to_transform(torch.zeros(3, 6)).shape == (3, 4, 4)
    z = torch.zeros(6, requires_grad=True)                       # identity pose: gradient must be finite
    tensor_to_transform(z).sum().backward()
    assert torch.isfinite(z.grad).all()


def test_settings_schema_and_schedule_selection():
    from loner_amd.common.settings import Settings, default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings
    s = default_optimizer_settings()
    assert s.model_config.model.render.N_samples_train == 512 and s.model_config.model.occ_model.voxel_size == 100
    assert s.num_samples.lidar == 512 and s.samples_selection.strategy == "OGM"
    assert isinstance(s.model_config.model.ray_range, tuple)          # lists become tuples on attribute access
    assert s["keyframe_schedule"][0]["iteration_schedule"][0]["num_iterations"] == 1000
    assert s.model_config.loss.JS_loss.max_js_score == 10.0
    o = OptimizationSettings.from_dict({"num_iterations": 7, "freeze_poses": True})
    assert (o.num_iterations, o.freeze_poses, o.latest_kf_only, o.freeze_sigma_mlp) == (7, True, False, False)
    import pickle
    assert pickle.loads(pickle.dumps(s)).num_samples.lidar == 512


def test_synthetic_scene_is_exact_and_has_transparent_rays():
    from loner_amd.utils import synthetic as SY
    from loner_amd.common.pose_utils import tensor_to_transform
    dirs, ts = SY.lidar_pattern()
    assert dirs.shape == (3, 65536) and abs(float(dirs.norm(dim=0).mean()) - 1) < 1e-6
    r = SY.scene_ranges(dirs, tensor_to_transform(SY.trajectory_pose6(3)[2]))
    assert float(r.min()) > 1.0 and int((r > 50).sum()) > 100          # window rays exceed the 50 m range
    scale, shift = SY.world_cube()
    assert abs(scale - 85.7614) < 1e-3 and np.allclose(shift, [7.5, 5.0, 0.0])


def test_bench_gpus_argument_spawns_that_many_ranks():
    """`python bench.py --gpus N` with no launcher around it must become N ranks (--gpus was parsed and ignored).
    --launch-check runs the argument -> torch.distributed.run -> process group path alone, over gloo on the CPU."""
    import json
    env = dict(os.environ, LNR_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and sorted(line["ranks"]) == [0, 1]
    # a launcher that started a different number of ranks than --gpus is an error, not a silently wrong n_gpus
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_reference_shaped_stand_ins_behave_like_the_classes_they_stand_for():
    """The boundary test (tests/test_gpu_mapping.py::test_optimizer_is_driven_by_reference_shaped_callers) is only as good as its
    stand-ins: AttrDict semantics (nested dicts wrapped, lists -> tuples on ATTRIBUTE access only), a Pose whose matrix is cached at
    construction and re-derived from the 6-vector only while that vector requires a gradient (pose.py:140-144), and surfaces that
    refuse members the reference classes do not have."""
    import pytest
    import torch
    from oracle import poses as OP
    from tests.support import RefShapedKeyFrame, RefShapedFrame, RefShapedLidarScan, RefShapedPose, RefShapedSettings
    s = RefShapedSettings({"a": {"b": [1, 2, {"c": 3}]}, "k": [{"n": 1}]})
    assert isinstance(s.a, RefShapedSettings) and s.a.b[:2] == (1, 2) and s.a.b[2].c == 3 and isinstance(s["a"]["b"], list)
    assert isinstance(s.k, tuple) and s.k[0].n == 1
    with pytest.raises(AttributeError):
        s.missing
    p6 = torch.tensor([0.3, -0.2, 0.1, 0.02, -0.4, 0.25])
    pose = RefShapedPose(pose_tensor=p6.clone(), fixed=True)
    assert torch.allclose(pose.get_transformation_matrix(), OP.transform_from_pose6(p6), atol=1e-6)
    with torch.no_grad():
        pose.get_pose_tensor()[0] += 1.0                                   # an optimiser stepped the vector ...
    assert float(pose.get_transformation_matrix()[0, 3]) == pytest.approx(0.3)      # ... a FIXED pose still hands out the cached matrix
    pose.set_fixed(False)
    assert float(pose.get_transformation_matrix()[0, 3]) == pytest.approx(1.3)      # a free one re-derives it
    scan = RefShapedLidarScan(torch.zeros(3, 5), torch.ones(5), torch.arange(5.0), sky_rays=torch.Tensor())
    assert len(scan) == 5
    with pytest.raises(AssertionError):
        scan._lnr_dev = 1                                                  # not a member of the reference's LidarScan
    fr = RefShapedFrame(None, scan, RefShapedPose())
    fr._lidar_pose = pose
    kf = RefShapedKeyFrame(fr)
    assert kf.get_lidar_scan() is scan and kf.get_lidar_pose() is pose and float(kf.get_time()) == 0.0
    with pytest.raises(AssertionError):
        kf.extra = 1


def test_squareplus_and_softplus_carry_tiny_cuda_nn_k_act():
    """tiny-cuda-nn evaluates Squareplus and Softplus on K_ACT x and divides by K_ACT, K_ACT = 10; the oracle's forms
    against closed forms in float64, and their derivatives (what the kernels' act_bwd restates) against autograd."""
    import torch
    from oracle import network as NW
    x = torch.linspace(-3.0, 3.0, 601, dtype=torch.float64, requires_grad=True)
    sp = NW._activate(x, "Softplus")
    assert torch.allclose(sp, torch.log1p(torch.exp(10.0 * x)) / 10.0, atol=1e-12)
    assert torch.allclose(torch.autograd.grad(sp.sum(), x)[0], torch.sigmoid(10.0 * x), atol=1e-12)
    sq = NW._activate(x, "Squareplus")
    y = 10.0 * x
    assert torch.allclose(sq, 0.5 * (y + torch.sqrt(y * y + 4.0)) / 10.0, atol=1e-12)
    assert torch.allclose(torch.autograd.grad(sq.sum(), x)[0], 0.5 * (1.0 + y / torch.sqrt(y * y + 4.0)), atol=1e-12)
    assert abs(float(NW._activate(torch.zeros(1), "Squareplus")) - 0.1) < 1e-7 and abs(float(NW._activate(torch.zeros(1), "Softplus")) - 0.0693147) < 1e-6
