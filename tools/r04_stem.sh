#!/bin/bash
# (record of a measured-and-removed experiment: the switch / code path it exercised is no longer in the library -- see DESIGN 4 "streams inside one step")
# GPU box: block 0's weight gradients beside the stem's backward (default) against the stem behind every weight gradient (5)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_overlap.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3 4; do
  for on in 5 4; do echo -n "fp32 configs1 wgrad_overlap=$on  "; bash tools/bv.sh --wgrad-overlap $on; done
done
