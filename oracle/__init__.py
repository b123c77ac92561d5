"""CPU oracle for the LONER mapping hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's mapping
iteration (umautobots/LONER, src/mapping/optimizer.py:194-626 and what it
calls).  It exists to *check* the HIP path; it is never the thing shipped or
measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``loner_amd/`` imports it, and the product fails loudly without its HIP
library.

Pinning
-------
Every stage here is checked against fixtures in ``tests/golden/*.npz`` that
were produced by *importing the reference itself* in the build container
(``tests/golden/make_golden.py``; the reference cannot travel to the GPU box).
Stages and their status:

* rays / far-clip / occupancy lookup / samplers / sample_pdf / volume render /
  target weights / JS divergence / LOS loss / occupancy step / Adam loop:
  **pinned** (bit-exact where the stage is integer- or rounding-defined,
  see ``torch_rounding.py``; <=1e-6 rel elsewhere).
* positional encoding + MLP (``network.py``): **parity unpinned**.  The
  reference delegates this to tinycudann (NVlabs/tiny-cuda-nn, installed
  un-versioned from git HEAD by docker/container_dockerhub.Dockerfile:64-65),
  a CUDA-only dependency that is absent from /root/reference and cannot be
  built here.  ``network.py`` restates tiny-cuda-nn's published algorithm
  (multiresolution hash grid, frequency encoding, bias-free MLP) in fp32 and
  is the *definition* the HIP kernels are held to.
* axis-angle -> matrix (``poses.py``): restates pytorch3d 0.7.2
  (docker/container_dockerhub.Dockerfile:67), also absent; **unpinned**,
  self-consistency only.
"""
