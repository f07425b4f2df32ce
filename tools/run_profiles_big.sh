#!/bin/bash
# rocprofv3 passes over tools/bench_bigscenes.py (one counter group per run, never --pmc together with a trace domain).
# Run on the GPU box from the repo root:  bash tools/run_profiles_big.sh <tag> <atrium|tetra>  -> gpurun_out/prof_<tag>/{trace,fetch,write,l2}
set -u
TAG=${1:-big}
WHICH=${2:-tetra}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprofv3 args...
    local name=$1; shift
    timeout -s KILL 280 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/tools/bench_bigscenes.py" "$WHICH" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
run trace --kernel-trace --stats &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
run l2 --pmc TCC_HIT_sum TCC_MISS_sum &&
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY &&
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE &&
run tatd --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE &&
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum &&
run utcl1 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
echo "profiles done rc=$?"
