"""Exact known answers for the density network, computed with rational arithmetic from tiny-cuda-nn's PUBLISHED algorithm -
written without importing oracle/network.py, so that the oracle (CPU) and the kernels (GPU) can both be checked against
something neither of them produced (tinycudann is absent, the oracle of the network is otherwise unpinned).

Algorithm restated (tiny-cuda-nn, GridEncoding / FullyFusedMLP as documented in its README and `grid.h`):
  level l:   scale_l = 2^(l * log2(per_level_scale)) * base_resolution - 1;   res_l = ceil(scale_l) + 1
             entries_l = min(round_up_8(res_l^3), 2^log2_hashmap_size)
  position:  pos = x * scale_l + 0.5 (x in [0,1]^3);  cell = floor(pos);  frac = pos - cell
  index:     stride = 1, index = 0; for each dim: if stride <= entries_l: index += cell[dim] * stride; stride *= res_l
             if entries_l < stride (the dense index does not fit): index = cell.x ^ cell.y * 2654435761 ^ cell.z * 805459861 (uint32)
             index %= entries_l
  feature:   sum over the 8 corners of prod_d (corner bit d ? frac_d : 1 - frac_d) * table_l[index(cell + corner)][f]
  network:   input = [features (level-major), 1, 1, ... up to a multiple of 16];  no biases;  weights [out][in] row-major,
             network parameters first, then the tables;  h = relu(W1 input);  sigma = (W_out h)[0]   (W_out padded to 16 rows)
Inputs of the reference are world coordinates in [-1,1]: x = (xyz + 1) / 2  (src/models/nerf_tcnn.py:63).

Everything here is a dyadic rational with few bits, so fp32 arithmetic reproduces the exact values: the comparison needs no
tolerance beyond accumulation-order effects (none for these magnitudes).
"""
from fractions import Fraction as Fr
from math import ceil

ENC = dict(otype="HashGrid", n_levels=2, n_features_per_level=2, log2_hashmap_size=5, base_resolution=2, per_level_scale=2.0)
NET = dict(activation="ReLU", n_neurons=16, n_hidden_layers=1, otype="FullyFusedMLP", output_activation="None")
POINTS_WORLD = [(Fr(-1, 2), Fr(0), Fr(1, 4)), (Fr(3, 4), Fr(-3, 4), Fr(1, 2)), (Fr(-1, 4), Fr(7, 8), Fr(-5, 8))]
D_SIGMA = [Fr(1), Fr(1, 2), Fr(-2)]

P_Y, P_Z = 2654435761, 805459861


def levels():
    out, offset = [], 0
    for l in range(ENC["n_levels"]):
        scale = Fr(2) ** l * ENC["base_resolution"] - 1              # per_level_scale = 2: exact
        res = ceil(scale) + 1
        entries = min((res ** 3 + 7) // 8 * 8, 2 ** ENC["log2_hashmap_size"])
        out.append(dict(scale=scale, res=res, entries=entries, offset=offset))
        offset += entries
    return out, offset


def grid_index(lv, cell):
    stride, index = 1, 0
    for d in range(3):
        if stride <= lv["entries"]:
            index += cell[d] * stride
            stride *= lv["res"]
    if lv["entries"] < stride:
        index = (cell[0] ^ (cell[1] * P_Y) ^ (cell[2] * P_Z)) & 0xFFFFFFFF
    return index % lv["entries"]


def lcg(seed):
    state = seed
    while True:
        state = (state * 1103515245 + 12345) % (1 << 31)
        yield (state >> 16) % 17 - 8                                  # integers in [-8, 8]


def parameters():
    """-> (flat list of Fractions: W1 [16][16], W_out [16][16], table level 0, table level 1), n_mlp"""
    lv, n_entries = levels()
    gen = lcg(12345)
    h, in_dim = NET["n_neurons"], 16
    w1 = [Fr(next(gen), 8) for _ in range(h * in_dim)]
    wo = [Fr(next(gen), 8) for _ in range(16 * h)]
    tab = [Fr(next(gen), 16) for _ in range(n_entries * 2)]
    return w1 + wo + tab, len(w1) + len(wo)


def corner_terms(lv, x):
    """-> list of (entry index inside the level, weight) for the 8 corners"""
    pos = [xd * lv["scale"] + Fr(1, 2) for xd in x]
    cell = [int(p // 1) for p in pos]
    frac = [p - c for p, c in zip(pos, cell)]
    out = []
    for k in range(8):
        bits = [(k >> d) & 1 for d in range(3)]
        w = Fr(1)
        for d in range(3):
            w *= frac[d] if bits[d] else 1 - frac[d]
        out.append((grid_index(lv, [cell[d] + bits[d] for d in range(3)]), w))
    return out


def corner_slopes(lv, x):
    """-> per axis d: list of (entry index, d weight / d x_d) - the derivative inside the cell floor() selected"""
    pos = [xd * lv["scale"] + Fr(1, 2) for xd in x]
    cell = [int(p // 1) for p in pos]
    frac = [p - c for p, c in zip(pos, cell)]
    out = [[], [], []]
    for k in range(8):
        bits = [(k >> d) & 1 for d in range(3)]
        idx = grid_index(lv, [cell[d] + bits[d] for d in range(3)])
        for axis in range(3):
            w = lv["scale"] * (1 if bits[axis] else -1)
            for d in range(3):
                if d != axis:
                    w *= frac[d] if bits[d] else 1 - frac[d]
            out[axis].append((idx, w))
    return out


def evaluate():
    """-> dict(sigma=[...], grad=[...flat, like the parameters...], d_pts=[[3] per point]) as Fractions: sigma per point,
    d(sum_i D_SIGMA_i sigma_i)/dparams, and D_SIGMA_i * d sigma_i / d xyz_world (xyz = 2x - 1)"""
    params, n_mlp = parameters()
    lv, _ = levels()
    h, in_dim, F = NET["n_neurons"], 16, 2
    w1, wo, tab = params[:h * in_dim], params[h * in_dim:n_mlp], params[n_mlp:]
    grad = [Fr(0)] * len(params)
    sigmas, hidden_on, d_pts = [], [], []
    for pw, ds in zip(POINTS_WORLD, D_SIGMA):
        x = [(c + 1) / 2 for c in pw]
        feats, terms = [], []
        for l in lv:
            ct = corner_terms(l, x)
            terms.append(ct)
            for f in range(F):
                feats.append(sum(w * tab[(l["offset"] + e) * F + f] for e, w in ct))
        inp = feats + [Fr(1)] * (in_dim - len(feats))
        zpre = [sum(w1[j * in_dim + i] * inp[i] for i in range(in_dim)) for j in range(h)]
        hid = [max(z, Fr(0)) for z in zpre]
        sigmas.append(sum(wo[j] * hid[j] for j in range(h)))                    # row 0 of W_out
        hidden_on.append(sum(1 for z in zpre if z > 0))
        # backward of ds * sigma
        for j in range(h):
            grad[h * in_dim + j] += ds * hid[j]
            if zpre[j] > 0:
                for i in range(in_dim):
                    grad[j * in_dim + i] += ds * wo[j] * inp[i]
        dinp = [sum(ds * wo[j] * w1[j * in_dim + i] for j in range(h) if zpre[j] > 0) for i in range(in_dim)]
        dx = [Fr(0)] * 3
        for li, l in enumerate(lv):
            for e, w in terms[li]:
                for f in range(F):
                    grad[n_mlp + (l["offset"] + e) * F + f] += w * dinp[li * F + f]
            for axis, sl in enumerate(corner_slopes(l, x)):
                for e, w in sl:
                    for f in range(F):
                        dx[axis] += w * tab[(l["offset"] + e) * F + f] * dinp[li * F + f]
        d_pts.append([v / 2 for v in dx])
    return dict(sigma=sigmas, grad=grad, hidden_on=hidden_on, n_mlp=n_mlp, params=params, d_pts=d_pts)


# Literals (computed once with the code above and pasted here, so that an edit of this file cannot silently move the target):
SIGMA_LITERAL = [Fr(-10899, 16384), Fr(-30851, 32768), Fr(-1582995, 2097152)]            # -0.66522216796875, -0.941497802734375, -0.7548308372497559
GRAD_PROBES_LITERAL = {32: Fr(-3539, 16384), 161: Fr(-51, 16384), 194: Fr(-11391, 65536), 512: Fr(-533, 65536), 513: Fr(-1001, 65536), 514: Fr(20869, 65536), 585: Fr(165, 512), 590: Fr(2585, 32768), 591: Fr(55, 512)}
N_NONZERO_GRAD = 197            # of 592 parameters (44 of them table entries)
