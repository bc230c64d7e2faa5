#!/bin/bash
# (record of a measured-and-removed experiment: the switch / code path it exercised is no longer in the library -- see DESIGN 4 "streams inside one step")
# GPU box: stream priorities -- the policy's no-grad forward (third stream) and the side stream at low priority, against equal priorities
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'P'
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
for p in (-1, 0, 1, 2):
    print(p, torch.cuda.Stream(priority=p).priority)
P
for rep in 1 2 3; do
  echo -n "fp32 equal priorities               "; bash tools/bv.sh
  echo -n "fp32 third low                      "; bash tools/bv.sh --fwd-overlap 3
  echo -n "fp32 third low + side low           "; bash tools/bv.sh --fwd-overlap 3 --side-priority 1
  echo -n "fp32 side low                       "; bash tools/bv.sh --side-priority 1
done
for rep in 1 2; do
  echo -n "bf16 equal priorities               "; bash tools/bv.sh --workload configs2
  echo -n "bf16 third low + side low           "; bash tools/bv.sh --workload configs2 --fwd-overlap 3 --side-priority 1
done
