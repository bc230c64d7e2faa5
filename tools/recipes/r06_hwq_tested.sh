cd $GRAFT_REPO_ROOT
for rep in 1 2; do for q in 4 5 6 8; do export GPU_MAX_HW_QUEUES=$q
 echo -n "rep $rep hwq $q configs1 "; bash tools/bv.sh
 echo -n "rep $rep hwq $q configs2 "; bash tools/bv.sh --workload configs2
done; done 2>&1 | grep -v amdgpu | tee gpurun_out/ab_hwq_tested_streams.log
