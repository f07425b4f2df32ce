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
run l2 --pmc TCC_HIT_sum TCC_MISS_sum
echo "profiles done rc=$?"
