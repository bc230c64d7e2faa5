#!/bin/bash
# The measurements behind profiles/rNN_* (copy gpurun_out/final/* to profiles/ with the round's prefix afterwards):
#   1. python bench.py                      -> bench.json   (the judged line: fp32 configs[1] headline + bf16 configs[2] leg, roofline, cpu_baseline)
#   per workload W in {b32 = fp32 configs[1], bf16_b128 = bf16 configs[2]}:
#   2. rocprofv3 --kernel-trace             -> bench_W_kernel_trace.txt          (same bench.py, 7 full steps, nothing else)
#   3. rocprofv3 --pmc SQ_* (own pass)      -> bench_W_pmc_sq.txt                (matrix-pipe busy cycles)
#   4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two own passes) -> bench_W_pmc_traffic.txt + pmc_traffic_W.json
#   5. the kernel trace again on plans with every stream overlap off -> bench_W_kernel_trace_serial.txt (each kernel alone on the device:
#      the view that matches bench.py's event timing of the roofline kernels)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
prof() {   # name, extra bench args...
  local name=$1; shift
  local B="python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 $*"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace -d $O -o kt_$name -- $B > $O/kt_$name.out 2> $O/kt_$name.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O -o ps_$name -- $B > $O/ps_$name.out 2> $O/ps_$name.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o pf_$name -- $B > $O/pf_$name.out 2> $O/pf_$name.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o pw_$name -- $B > $O/pw_$name.out 2> $O/pw_$name.err
  cd $R
  python tools/rocprof_summary.py $(find $O -name "kt_${name}_results.db") 7 > $O/bench_${name}_kernel_trace.txt
  python tools/rocprof_summary.py $(find $O -name "ps_${name}_results.db") 7 > $O/bench_${name}_pmc_sq.txt
  python tools/pmc_traffic.py $(find $O -name "pf_${name}_results.db") $(find $O -name "pw_${name}_results.db") $O/pmc_traffic_${name}.json > $O/bench_${name}_pmc_traffic.txt
  python tools/rocprof_phases.py $(find $O -name "kt_${name}_results.db") > $O/phases_${name}.txt
  python tools/rocprof_timeline.py $(find $O -name "kt_${name}_results.db") 0.5 > $O/timeline_${name}.txt
  python tools/rocprof_by_shape.py $(find $O -name "kt_${name}_results.db") 3.5 0.5 > $O/launch_shapes_${name}.txt
}
serial() {   # name, extra bench args...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $O -o ks_$name -- python $R/bench.py --no-cpu-baseline --no-extras --no-m1 --no-roofline --sustained-seconds 0 --steps 5 --warmup 2 --no-overlap --wgrad-overlap 0 --fwd-overlap 0 $* > $O/ks_$name.out 2> $O/ks_$name.err )
  python tools/rocprof_summary.py $(find $O -name "ks_${name}_results.db") 7 > $O/bench_${name}_kernel_trace_serial.txt
  python tools/rocprof_by_shape.py $(find $O -name "ks_${name}_results.db") 3.5 0.5 > $O/launch_shapes_serial_${name}.txt
}
prof b32; serial b32
prof bf16_b128 --workload configs2; serial bf16_b128 --workload configs2
# round 6: the fp32 step with the transform-domain GEMMs on the fp32 matrix pipe (the form of rounds 1-5), serial trace only -- the comparison leg of bench.py
serial b32_mfma --plan-option gemm_split=0
# ... and BASELINE configs[3] on one GPU (two nets x 256, a launch stream per robot group): how many kernels run at once
bash $R/tools/recipes/overlap.sh configs3 --workload configs3 --steps 4 --warmup 2 > /dev/null; cp $R/gpurun_out/configs3_overlap.txt $O/phases_configs3.txt
find $O -name "*.db" -delete
find $O -type d -empty -delete
tail -c 1500 $O/bench.json
