#!/bin/bash
# usage (on the GPU box): tools/bt.sh <out-name> <B> <shapes...>   -- kernel-trace summary of tools/bf16_tiles.py
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/bt_$out -o bt -- python /root/repo/tools/bf16_tiles.py "$@" > /root/repo/gpurun_out/bt_$out.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $(find gpurun_out/bt_$out -name "*.db" | head -1) > gpurun_out/bt_$out.txt
rm -rf gpurun_out/bt_$out
