# third soak of the round, at HEAD after the two-chains-per-workgroup chi^2 and the leaner sqrt
# (other seeds).  A step that times out ends the script: no GPU step is started after it.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_soak3
mkdir -p $O
step() {                                  # step <seconds> <name> <args...>
    local secs=$1 name=$2; shift 2
    timeout -k 10 $secs python3 tests/soak/$name.py "$@" > $O/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a $O/rc.txt
    tail -1 $O/$name.log
    return $rc
}
step 300 fuzz_models 900 52 && step 300 fuzz_gauss 10000 51 && step 120 fuzz_reductions 6000 53 && step 200 fuzz_gibbs_n 15000 54
rc=$?
grep -c MISMATCH $O/*.log || true
exit $rc
