#!/bin/bash
# Where do the integrator's L2 misses come from: read/write split at the L2 and at its memory side, and the L1's miss stream.
# Run on the GPU box from the repo root:  bash tools/run_profiles_l2.sh <tag>   -> gpurun_out/prof_<tag>/{l2rw,ea,tcp}
set -u
TAG=${1:-l2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/tools/profile_kernels.py" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
run l2rw --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum &&
run ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum &&
run tcp --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
echo "l2 profiles done rc=$?"
