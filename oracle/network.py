"""Density network: input encoding + bias-free MLP -- CPU oracle, torch fp32/fp64.

Test infrastructure.  PARITY UNPINNED: the reference instantiates
tinycudann.NetworkWithInputEncoding (src/models/nerf_tcnn.py:35-38) and calls
it at nerf_tcnn.py:63-72; tinycudann (NVlabs/tiny-cuda-nn, un-versioned git
HEAD, docker/container_dockerhub.Dockerfile:64-65) is CUDA-only and absent.
This module restates its published algorithm and is the definition the HIP
kernels are tested against:

* config schema = the reference's cfg/nerf_config/default_nerf_hash.yaml
  (`pos_encoding_sigma`, `sigma_network`);
* one flat float32 parameter vector, MLP matrices first then encoding tables;
* MLP: no biases; matrices are [out, in] row-major; the input width is padded
  to a multiple of 16 and the output to 16 rows (row 0 is sigma); activation
  after every hidden layer, none on the output;
* HashGrid: level l has scale = base*pls^l - 1, res = ceil(scale)+1,
  min(roundup8(res^3), 2^log2_T) entries of F features; lookup position
  fma(x, scale, 0.5) - ONE rounding, as tiny-cuda-nn's `fmaf(scale, input,
  0.5f)` (`pos_rounding="mul_add"` rounds product and sum separately, what
  un-contracted code would do; the two differ by an fp32 ulp of the position,
  1/32 cell on the finest default level) - trilinear over the 8 surrounding
  entries; entry index is the
  dense x + y*res + z*res^2 while res^3 <= table size, else the
  coherent-prime hash x ^ y*2654435761 ^ z*805459861 (uint32), always modulo
  the table size;
* Frequency: feature [dim][k][sin,cos] = sin(2^k*pi*x_dim + {0, pi/2}).
Inputs are the unit-cube coordinates (xyz+1)/2 (nerf_tcnn.py:63).

precision="fp16" models the storage types the reference actually runs with
(tinycudann in half precision): encoded features and MLP weights are rounded
to fp16 (round-to-nearest-even, straight-through for the gradient, i.e. fp32
master parameters as tinycudann keeps them), products accumulate in fp32 (the
HIP kernels use v_mfma_f32_16x16x32_f16; tinycudann accumulates in fp16), the
last hidden activation feeds the 1-row output product unrounded.  It exists so
that the error of the HIP fp16 mode against the fp32 definition can be split
into "storage rounding" (this model vs fp32) and "kernel arithmetic" (kernel
vs this model).

Remaining assumptions about tinycudann that nothing here can verify (the
package is absent): fused position rounding, padding inputs fed the constant
1, parameter order "MLP matrices, then tables", `mod T` applied to dense
levels too, Xavier-uniform / uniform(-1e-4, 1e-4) initialisation, K_ACT = 10
inside Squareplus and Softplus (non-default activations).
"""
import math
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861


@dataclass
class GridLevel:
    scale: float
    res: int
    size: int
    offset: int      # in entries, relative to the start of the encoding block
    hashed: bool


@dataclass
class NetworkSpec:
    """Parsed (encoding_config, network_config) pair."""
    enc_type: str = "HashGrid"            # HashGrid | Frequency
    n_levels: int = 16
    n_features: int = 2
    log2_table: int = 18
    base_res: int = 16
    per_level_scale: float = 2.0
    n_frequencies: int = 12
    activation: str = "ReLU"              # ReLU | Sine | None | ...
    n_neurons: int = 64
    n_hidden: int = 1
    precision: str = "fp32"               # fp32 | fp16 (storage rounding of features and weights, see the module docstring)
    pos_rounding: str = "fma"             # fma | mul_add
    levels: List[GridLevel] = field(default_factory=list)

    @staticmethod
    def from_config(enc: dict, net: dict) -> "NetworkSpec":
        s = NetworkSpec()
        s.enc_type = enc.get("otype", "HashGrid")
        if s.enc_type in ("HashGrid", "Grid"):
            s.enc_type = "HashGrid"
            s.n_levels = int(enc.get("n_levels", 16))
            s.n_features = int(enc.get("n_features_per_level", 2))
            s.log2_table = int(enc.get("log2_hashmap_size", 19))
            s.base_res = int(enc.get("base_resolution", 16))
            s.per_level_scale = float(enc.get("per_level_scale", 2.0))
            s.pos_rounding = str(enc.get("pos_rounding", "fma"))
        elif s.enc_type == "Frequency":
            s.n_frequencies = int(enc.get("n_frequencies", 12))
        else:
            raise ValueError(f"unsupported encoding {s.enc_type}")
        s.activation = str(net.get("activation", "ReLU"))
        s.n_neurons = int(net.get("n_neurons", 64))
        s.n_hidden = int(net.get("n_hidden_layers", 1))
        # ("fp32_chain": the kernels' exact-fma-chain variant of fp32 - the same arithmetic definition here)
        s.precision = {"fp32": "fp32", "float32": "fp32", "fp32_chain": "fp32", "fp16": "fp16", "half": "fp16", "float16": "fp16"}[str(net.get("precision", "fp32"))]
        s._build_levels()
        return s

    def _build_levels(self):
        self.levels = []
        if self.enc_type != "HashGrid":
            return
        cap = 1 << self.log2_table
        off = 0
        for l in range(self.n_levels):
            # float32 arithmetic as in the published implementation
            scale = float(np.float32(np.exp2(np.float32(l) * np.log2(np.float32(self.per_level_scale)))
                                     * np.float32(self.base_res) - np.float32(1.0)))
            res = int(math.ceil(scale)) + 1
            dense = res ** 3
            size = min(((dense + 7) // 8) * 8, cap)
            self.levels.append(GridLevel(scale, res, size, off, hashed=dense > size))
            off += size

    @property
    def enc_dim(self) -> int:
        return self.n_levels * self.n_features if self.enc_type == "HashGrid" else 3 * 2 * self.n_frequencies

    @property
    def in_dim(self) -> int:
        return ((self.enc_dim + 15) // 16) * 16

    @property
    def mlp_shapes(self):
        shapes = [(self.n_neurons, self.in_dim)]
        shapes += [(self.n_neurons, self.n_neurons)] * (self.n_hidden - 1)
        shapes += [(16, self.n_neurons)]
        return shapes

    @property
    def n_mlp_params(self) -> int:
        return sum(a * b for a, b in self.mlp_shapes)

    @property
    def n_enc_params(self) -> int:
        return sum(l.size for l in self.levels) * self.n_features if self.enc_type == "HashGrid" else 0

    @property
    def n_params(self) -> int:
        return self.n_mlp_params + self.n_enc_params


def init_params(spec: NetworkSpec, seed: int = 0) -> torch.Tensor:
    """Xavier-uniform matrices, uniform(-1e-4, 1e-4) tables (float32)."""
    g = torch.Generator().manual_seed(seed)
    chunks = []
    for (o, i) in spec.mlp_shapes:
        bound = math.sqrt(6.0 / (i + o))
        chunks.append((torch.rand(o * i, generator=g) * 2 - 1) * bound)
    if spec.n_enc_params:
        chunks.append((torch.rand(spec.n_enc_params, generator=g) * 2 - 1) * 1e-4)
    return torch.cat(chunks).float()


K_ACT = 10.0


def _activate(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "ReLU":
        return torch.relu(x)
    if kind == "Sine":
        return torch.sin(x)
    if kind == "None":
        return x
    if kind == "LeakyReLU":
        return torch.where(x > 0, x, 0.01 * x)
    if kind == "Exponential":
        return torch.exp(x)
    if kind == "Sigmoid":
        return torch.sigmoid(x)
    # tiny-cuda-nn evaluates these two on K_ACT x and divides the result by K_ACT, K_ACT = 10 (its published activation code, read from
    # memory like the rest of this file - the package is absent; rounds 1-5 used the unscaled forms)
    if kind == "Squareplus":
        y = K_ACT * x
        return 0.5 * (y + torch.sqrt(y * y + 4.0)) / K_ACT   # hyper-parameter b=2 -> b^2 = 4
    if kind == "Softplus":
        return torch.nn.functional.softplus(x, beta=K_ACT)   # log(1 + exp(K_ACT x)) / K_ACT
    if kind == "Tanh":
        return torch.tanh(x)
    raise ValueError(kind)


class _RoundF16(torch.autograd.Function):
    """round to fp16 and back (nearest even); the gradient passes straight through (fp32 master copy)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def round_f16(x: torch.Tensor) -> torch.Tensor:
    return _RoundF16.apply(x)


def grid_position(spec: NetworkSpec, x: torch.Tensor, scale: float) -> torch.Tensor:
    """x*scale + 0.5 with the rounding `spec.pos_rounding` asks for (float32 inputs; float64 inputs are exact enough
    either way).  fma: the product of two fp32 numbers is exact in fp64 and so is adding 0.5 to it at these magnitudes
    (48 significant bits), so rounding the fp64 result once to fp32 IS the fused multiply-add."""
    if x.dtype == torch.float32 and spec.pos_rounding == "fma":
        return (x.double() * float(scale) + 0.5).to(torch.float32)
    return x * scale + 0.5


def encode_hashgrid(spec: NetworkSpec, table: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """table [E, F] (all levels), x [B,3] in [0,1] -> [B, L*F]; differentiable in both."""
    feats = []
    F = spec.n_features
    for lv in spec.levels:
        pos = grid_position(spec, x, lv.scale)
        cell = torch.floor(pos)
        frac = pos - cell
        cell = cell.detach().to(torch.int64)
        acc = torch.zeros(x.shape[0], F, dtype=x.dtype, device=x.device)
        for corner in range(8):
            bx, by, bz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            cx = cell[:, 0] + bx
            cy = cell[:, 1] + by
            cz = cell[:, 2] + bz
            w = (frac[:, 0] if bx else 1 - frac[:, 0]) \
                * (frac[:, 1] if by else 1 - frac[:, 1]) \
                * (frac[:, 2] if bz else 1 - frac[:, 2])
            if lv.hashed:
                m = 0xFFFFFFFF
                idx = ((cx & m) ^ ((cy * PRIME_Y) & m) ^ ((cz * PRIME_Z) & m)) % lv.size
            else:
                idx = (cx + cy * lv.res + cz * lv.res * lv.res) % lv.size
            acc = acc + w[:, None] * table[lv.offset + idx]
        feats.append(acc)
    return torch.cat(feats, dim=1)


def encode_frequency(spec: NetworkSpec, x: torch.Tensor) -> torch.Tensor:
    """x [B,3] -> [B, 3*2*n_freq] ordered [dim][k][sin,cos]."""
    k = torch.arange(spec.n_frequencies, dtype=x.dtype, device=x.device)
    phase = x[:, :, None] * torch.exp2(k)[None, None, :] * math.pi        # [B,3,K]
    both = torch.stack([torch.sin(phase), torch.sin(phase + math.pi / 2)], dim=-1)
    return both.reshape(x.shape[0], -1)


def density(spec: NetworkSpec, params: torch.Tensor, xyz: torch.Tensor) -> torch.Tensor:
    """params flat [P]; xyz [B,3] in the world cube [-1,1] -> sigma [B]."""
    return density_unit(spec, params, (xyz + 1) / 2)


def density_unit(spec: NetworkSpec, params: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Same, for inputs already mapped to the unit cube [0,1]^3."""
    mats = []
    cur = 0
    for (o, i) in spec.mlp_shapes:
        mats.append(params[cur:cur + o * i].reshape(o, i))
        cur += o * i
    if spec.enc_type == "HashGrid":
        table = params[cur:].reshape(-1, spec.n_features)
        h = encode_hashgrid(spec, table, x)
    else:
        h = encode_frequency(spec, x)
    if h.shape[1] < spec.in_dim:   # padded inputs are fed the constant 1 (tiny-cuda-nn pads with ones)
        h = torch.cat([h, torch.ones(h.shape[0], spec.in_dim - h.shape[1], dtype=h.dtype, device=h.device)], dim=1)
    half = spec.precision == "fp16"
    if half:
        mats = [round_f16(m) for m in mats]
    for i, m in enumerate(mats[:-1]):
        h = _activate((round_f16(h) if half else h) @ m.T, spec.activation)
    out = h @ mats[-1].T
    return out[:, 0]
