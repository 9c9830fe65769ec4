# fourth soak of the round: the model cases again at HEAD (one-launch energy, two-entry memo,
# packed targets, one-barrier block reduction), other seeds.  A step that times out ends the script.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_soak4
mkdir -p $O
step() {
    local secs=$1 name=$2; shift 2
    timeout -k 10 $secs python3 tests/soak/$name.py "$@" > $O/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a $O/rc.txt
    tail -1 $O/$name.log
    return $rc
}
step 500 fuzz_models 1200 62 && step 200 fuzz_gibbs_n 12000 64
rc=$?
grep -c MISMATCH $O/*.log || true
exit $rc
