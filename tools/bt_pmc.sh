#!/bin/bash
# usage (GPU box): tools/bt_pmc.sh <out-name> "<counters>" <bf16_tiles args...>
out=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr -d /root/repo/gpurun_out/btp_$out -o bt -- python /root/repo/tools/bf16_tiles.py "$@" > /root/repo/gpurun_out/btp_$out.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $(find gpurun_out/btp_$out -name "*.db" | head -1) > gpurun_out/btp_$out.txt
rm -rf gpurun_out/btp_$out
