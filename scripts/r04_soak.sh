# round-4 soak: every randomised differential fuzzer at the round's HEAD (new seeds) -- the
# whole-tile MFMA gradient kernel, per-transition stream positions, the registry dispatch.
# A step that times out ends the script: no GPU step is started after it.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04_soak}
mkdir -p $O
step() {                                  # step <seconds> <name> <args...>
    local secs=$1 name=$2; shift 2
    timeout -k 10 $secs python3 tests/soak/$name.py "$@" > $O/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a $O/rc.txt
    tail -1 $O/$name.log
    return $rc
}
step 300 fuzz_models 700 81 && step 300 fuzz_gauss 9000 82 && step 120 fuzz_reductions 5000 83 && step 200 fuzz_gibbs_n 10000 84 && step 200 fuzz_graph 8000 85
rc=$?
grep -c MISMATCH $O/*.log || true
exit $rc
