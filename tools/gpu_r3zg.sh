#!/bin/bash
# what does the event sampling inside the timed region cost?
for e in 4 0 4 0 10; do timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-every $e 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('profile-every $e:', round(d['ms_per_step'],4))
except Exception as ex: print('profile-every $e: failed', ex)"; done
