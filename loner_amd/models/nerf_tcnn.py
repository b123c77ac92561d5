"""DecoupledNeRF with the reference's attribute layout (src/models/nerf_tcnn.py:18-95).

The density branch (`_model_sigma`) is the hand-written HIP network (hash-grid / frequency
encoding + bias-free MLP, fp32) behind a module that looks like tinycudann's
NetworkWithInputEncoding: one flat float32 `params` Parameter (MLP matrices first, then the
encoding tables), `n_output_dims`, `dtype`.  The colour branch is never evaluated for LiDAR
(sigma_only=True, optimizer.py:466 with camera=False; frozen at optimizer.py:235); its parameter
tensors exist so that state_dict keys match, but no kernel touches them.
"""
import torch
import torch.nn as nn

from .. import hip, ops


class _DensityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, params, spec):
        sigma = ops.density_forward(spec, params.detach(), pts=pts.detach())
        ctx.spec = spec
        ctx.save_for_backward(pts, params)
        return sigma

    @staticmethod
    def backward(ctx, d_sigma):
        pts, params = ctx.saved_tensors
        want_pts = ctx.needs_input_grad[0]
        grad_params = torch.zeros_like(params)
        d_pts = ops.density_backward(ctx.spec, params.detach(), d_sigma.contiguous(), grad_params, pts=pts.detach(),
                                     want_d_pts=want_pts)
        if d_pts is not None:
            d_pts = d_pts.reshape(pts.shape)
        return d_pts, (grad_params if ctx.needs_input_grad[1] else None), None


class SigmaNetwork(nn.Module):
    """tinycudann.NetworkWithInputEncoding stand-in for the density branch."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=None):
        super().__init__()
        if n_input_dims != 3 or n_output_dims != 1:
            raise RuntimeError("SigmaNetwork supports 3 inputs and 1 output")
        self.encoding_config = dict(encoding_config)
        self.network_config = dict(network_config)
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.dtype = torch.float32
        self._spec = None
        n_mlp, n_total, shapes = _param_layout(self.spec)
        gen = torch.Generator()
        gen.manual_seed(int(torch.randint(0, 2 ** 31, (1,)).item()) if seed is None else int(seed))
        chunks = []
        for (o, i) in shapes:                              # Xavier-uniform matrices
            bound = (6.0 / (i + o)) ** 0.5
            chunks.append((torch.rand(o * i, generator=gen) * 2 - 1) * bound)
        if n_total > n_mlp:                                # uniform(-1e-4, 1e-4) tables
            chunks.append((torch.rand(n_total - n_mlp, generator=gen) * 2 - 1) * 1e-4)
        self.params = nn.Parameter(torch.cat(chunks).float())

    @property
    def spec(self) -> hip.NetSpec:
        if self._spec is None:
            self._spec = hip.make_net_spec(self.encoding_config, self.network_config)
        return self._spec

    def __getstate__(self):           # ctypes structs do not pickle; rebuild lazily in the child process
        state = self.__dict__.copy()
        state["_spec"] = None
        return state

    def density(self, pts_world: torch.Tensor) -> torch.Tensor:
        """pts in the world cube [-1,1]^3, any leading shape -> sigma with that shape."""
        shape = pts_world.shape[:-1]
        return _DensityFn.apply(pts_world.reshape(-1, 3).float(), self.params, self.spec).reshape(shape)

    def forward(self, x_unit: torch.Tensor) -> torch.Tensor:
        """tinycudann call convention: inputs in [0,1]^3 -> [B, n_output_dims]."""
        return self.density(x_unit * 2 - 1)[..., None]


def _param_layout(spec):
    h, nh = spec.n_neurons, spec.n_hidden
    shapes = [(h, spec.in_dim)] + [(h, h)] * (nh - 1) + [(16, h)]
    return spec.n_mlp_params, int(spec.n_params), shapes


class _FrozenParams(nn.Module):
    """Parameter holder for the (unused) colour branch."""

    def __init__(self, count, n_output_dims=0):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(count), requires_grad=False)
        self.n_output_dims = n_output_dims
        self.dtype = torch.float32


class DecoupledNeRF(nn.Module):
    def __init__(self, cfg, num_colors=3):
        super().__init__()
        self._num_colors = num_colors
        self.cfg = cfg
        self._enable_view_dependence = cfg["enable_view_dependence"]
        self._model_sigma = SigmaNetwork(3, 1, cfg["pos_encoding_sigma"], cfg["sigma_network"])
        # colour branch: sized like the reference's modules, never evaluated on the lidar path
        pe = hip.make_net_spec(dict(cfg["pos_encoding_intensity"]), dict(n_neurons=16, n_hidden_layers=1))
        self._pos_encoding = _FrozenParams(int(pe.n_params) - pe.n_mlp_params, pe.enc_dim)
        self._dir_encoding = _FrozenParams(0, 16) if self._enable_view_dependence else None
        net = dict(cfg["intensity_network"])
        w, nh = int(net.get("n_neurons", 64)), int(net.get("n_hidden_layers", 4))
        in_dim = -(-(pe.enc_dim + (16 if self._enable_view_dependence else 0)) // 16) * 16
        self._model_intensity = _FrozenParams(w * in_dim + (nh - 1) * w * w + 16 * w, num_colors)
        # the clip limits are those of the network's dtype (nerf_tcnn.py:51-52): fp32, or fp16 in the reference's precision
        half = hip.PRECISIONS[str(dict(cfg["sigma_network"]).get("precision", "fp32"))] == 1
        self._max_float = torch.finfo(torch.float16 if half else torch.float32).max
        self._min_float = torch.finfo(torch.float16 if half else torch.float32).min
        self._warn_infinite = True
        self._clipped_seen = 0

    def forward(self, pos, dir=None, sigma_only=False, detach_sigma=True):
        """pos [N,3] in the world cube [-1,1] -> sigma [N,1] (nerf_tcnn.py:59-82).  Only sigma_only=True
        (the lidar path) is implemented; the colour branch is out of scope."""
        if not sigma_only:
            raise NotImplementedError("colour rendering (sigma_only=False) is not part of the LiDAR mapping path")
        sigma = self._model_sigma.density(pos)[..., None]
        self.warn_if_clipped(pos.device)
        return sigma

    def warn_if_clipped(self, device):
        """The kernels clip non-finite densities themselves (nan_to_num semantics, nerf_tcnn.py:70-78) and count them in a
        status word of the workspace; this prints the reference's warning the first time the count moves.  Reads one device
        word (a sync - the reference's `torch.isfinite(sigma).all()` is one too); the training loop calls it once per phase."""
        if not self._warn_infinite:
            return
        n = ops.density_clipped_count(device)
        if n > self._clipped_seen:
            print("Warning: Clipping infinite outputs. Will not warn about this again (but it will happen again)")
            self._warn_infinite = False
        self._clipped_seen = n
