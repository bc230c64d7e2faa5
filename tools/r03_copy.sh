#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in fixed gather; do
  rm -rf /tmp/cp_$m; rocprofv3 --kernel-trace -d /tmp/cp_$m -o cp -- python $R/tools/copy_probe.py $m > /tmp/cp_$m.out 2>&1
  echo "== $m"; python $R/tools/rocprof_summary.py $(find /tmp/cp_$m -name "cp_results.db") 10 | grep -i "copyBuffer\|fillBuffer\|replay_gather\|total kernel"
done | tee $R/gpurun_out/copy_probe.log
