#!/bin/bash
# GPU box: one bench.py run reduced to "value (M2) / fwd+bwd only (M1) / ms per step";  usage: tools/bv.sh [bench args]
python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --no-cpu-baseline --no-extras --no-roofline --sustained-seconds 0 --steps 30 --warmup 5 "$@" 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('M2 %.1f  M1 %.1f  ms/step %.3f' % (d['value'], d['value_fwd_bwd_only'], d['ms_per_step']))"
