#!/bin/bash
# several recipes in ONE lease (a box takes minutes to get):  tools/lease.sh multi 1500 "pytest tests/x.py -k y" "ab 2 a:--x b:--y" ...
#   every argument is one recipe invocation, split on blanks (quote-free arguments only); output of each under its usual gpurun_out/ file
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for inv in "$@"; do
  set -- $inv
  r=$1; shift
  echo "=== $r $*"
  bash tools/recipes/$r.sh "$@"
done
