#!/bin/bash
export TMPDIR=/tmp
for v in 1 0; do
  echo "== side priority $v"
  LNR_SIDE_PRIORITY=$v timeout 300 python - <<'PY' 2>/dev/null | python tools/bench_kernels.py | head -2
import os, sys
sys.argv = ["bench.py", "--steps", "60", "--warmup", "10", "--no-cpu-baseline"]
sys.path.insert(0, ".")
from loner_amd.mapping import optimizer as OM
pri = os.environ["LNR_SIDE_PRIORITY"] == "1"
orig = OM.Optimizer.__init__
def init(self, *a, **k):
    orig(self, *a, **k); self._side_priority = pri
OM.Optimizer.__init__ = init
import bench
bench.main()
PY
done
