#!/bin/bash
# GPU box: the measurements behind profiles/rNN_* (run from the repo root through gpurun):
#   1. python bench.py                      -> gpurun_out/final/bench_fp32.json   (the judged line, with cpu_baseline)
#   2. rocprofv3 --kernel-trace             -> kernel_trace.txt                   (same command, fewer steps)
#   3. rocprofv3 --pmc SQ_* (own pass)      -> pmc_sq.txt                         (matrix-pipe busy cycles)
#   4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two own passes) -> pmc_traffic.txt + pmc_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_fp32.json 2> $O/bench_fp32.err
B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --steps 5 --warmup 2"   # 7 full steps, nothing else
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o kt -- $B > $O/kt.out 2> $O/kt.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O -o ps -- $B > $O/ps.out 2> $O/ps.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o pf -- $B > $O/pf.out 2> $O/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o pw -- $B > $O/pw.out 2> $O/pw.err
cd $R
python tools/rocprof_summary.py $(find $O -name "kt_results.db") 7 > $O/kernel_trace.txt
python tools/rocprof_summary.py $(find $O -name "ps_results.db") 7 > $O/pmc_sq.txt
python tools/pmc_traffic.py $(find $O -name "pf_results.db") $(find $O -name "pw_results.db") $O/pmc_traffic.json > $O/pmc_traffic.txt
find $O -name "*.db" -size +30M -delete
tail -c 600 $O/bench_fp32.json
