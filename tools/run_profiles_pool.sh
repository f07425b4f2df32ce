#!/bin/bash
# k_integrate against k_integrate_pool under the profiler: kernel trace + the counter passes that matter for the comparison.
# Run on the GPU box from the repo root:  bash tools/run_profiles_pool.sh <tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02pool}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PROFILE_POOLED=1
run() {
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/tools/profile_kernels.py" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
run trace --kernel-trace --stats &&
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run tatd --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE &&
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum
echo "profiles done rc=$?"
