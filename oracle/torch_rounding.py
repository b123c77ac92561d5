"""NumPy emulation of the rounding behaviour of two torch CPU reductions.

Test infrastructure (see oracle/__init__.py).

The reference's importance sampler (src/models/rendering_tcnn.py:33-37) calls
``torch.sum(w, -1)`` and ``torch.cumsum(pdf, -1)`` on float32 rows; the bits of
their results decide which bin ``searchsorted`` picks.  To make sample indices
bit-identical the exact association order has to be reproduced:

* ``sum`` over a contiguous last dim (ATen SumKernel, "cascade sum"): the row is
  read as 8-lane vectors, 4 interleaved vector accumulators, and a 4-level
  cascade that is flushed every ``2**max(4, ceil_log2(n)//4)`` steps; rows
  shorter than one vector use the same scheme on scalars.
* ``cumsum``: running sum kept in float64, every prefix rounded to float32.

Both were verified bit-for-bit against torch 2.10 CPU in the build container
for row lengths 1..69, 126..8190 and 20000 (tests/test_oracle_rounding.py
re-checks against whatever torch is installed).
"""
import numpy as np

_VEC = 8      # lanes of the float32 vector the kernel was compiled for
_ILP = 4      # interleaved accumulators
_LEVELS = 4   # cascade depth


def _ceil_log2(n: int) -> int:
    return 0 if n <= 1 else int(n - 1).bit_length()


def _cascade(fetch, count, zero):
    """Sum ``fetch(0..count-1)`` with ATen's multi-level cascade."""
    power = max(4, _ceil_log2(count) // _LEVELS)
    step = 1 << power
    mask = step - 1
    acc = [zero.copy() for _ in range(_LEVELS)]
    i = 0
    while i + step <= count:
        for _ in range(step):
            acc[0] = acc[0] + fetch(i)
            i += 1
        for lvl in range(1, _LEVELS):
            acc[lvl] = acc[lvl] + acc[lvl - 1]
            acc[lvl - 1] = zero.copy()
            if i & (mask << (lvl * power)):
                break
    while i < count:
        acc[0] = acc[0] + fetch(i)
        i += 1
    for lvl in range(1, _LEVELS):
        acc[0] = acc[0] + acc[lvl]
    return acc[0]


def _row_sum(x: np.ndarray, lanes: int) -> np.ndarray:
    """x: [N, M*lanes] -> [N, lanes] using the 4-way interleave + cascade."""
    n = x.shape[0]
    m = x.shape[1] // lanes
    xv = x[:, : m * lanes].reshape(n, m, lanes)
    groups = m // _ILP
    zero = np.zeros((n, _ILP, lanes), np.float32)
    if groups > 0:
        part = _cascade(lambda i: xv[:, _ILP * i:_ILP * i + _ILP, :], groups, zero)
    else:
        part = zero
    for i in range(groups * _ILP, m):
        part[:, 0, :] = part[:, 0, :] + xv[:, i, :]
    out = part[:, 0, :]
    for k in range(1, _ILP):
        out = out + part[:, k, :]
    return out


def sum_lastdim_f32(x: np.ndarray) -> np.ndarray:
    """Bit-faithful ``torch.sum(x, -1)`` for a contiguous float32 [N, K] array."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, k = x.shape
    if k < _VEC:
        return _row_sum(x, 1)[:, 0]
    nvec = k // _VEC
    vacc = _row_sum(x[:, : nvec * _VEC], _VEC)
    total = np.zeros(n, np.float32)
    for j in range(nvec * _VEC, k):
        total = total + x[:, j]
    for lane in range(_VEC):
        total = total + vacc[:, lane]
    return total


def cumsum_lastdim_f32(x: np.ndarray) -> np.ndarray:
    """Bit-faithful ``torch.cumsum(x, -1)`` for float32 [N, K]."""
    x = np.asarray(x, dtype=np.float32)
    run = np.zeros(x.shape[0], np.float64)
    out = np.empty_like(x)
    for j in range(x.shape[1]):
        run = run + x[:, j].astype(np.float64)
        out[:, j] = run.astype(np.float32)
    return out
