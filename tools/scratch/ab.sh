#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 20 --warmup 3 $@ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_step'])"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do echo -n "full: "; run; done
for sc in atrium_like temple_like street_like alpha_test texture_test; do echo -n "$sc: "; run --scene $sc --steps 5; done
echo -n "shard: "; run --emulate-shard 0/8
