#!/bin/bash
# does the timed window's position matter?  (the driver runs --steps 20 --warmup 5)
for w in 5 10 40; do timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('warmup $w steps 20:', round(d['ms_per_step'],4), {k:d['kernels_ms'][k] for k in ('encode_forward','encode_backward','table_grad_reduce','mlp_backward')}, d['kernels_ms'].get('occ_grid_step'))"; done
