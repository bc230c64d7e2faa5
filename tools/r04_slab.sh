#!/bin/bash
# (record of a measured-and-removed experiment: the switch / code path it exercised is no longer in the library -- see DESIGN 4 "streams inside one step")
# GPU box: the image-tile weight gradient's slab reduction on the side stream (bf16; --wgrad-overlap 0 = behind it on the main stream)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_overlap.py tests/test_gpu_bf16_points.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3 4; do
  for on in 0 4; do echo -n "bf16 configs2 wgrad_overlap=$on  "; bash tools/bv.sh --workload configs2 --wgrad-overlap $on; done
done
