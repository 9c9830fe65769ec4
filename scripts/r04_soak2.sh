# second round-4 soak: other seeds, more cases (about 15 minutes of GPU time)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04_soak2}
mkdir -p $O
step() {                                  # step <seconds> <name> <args...>
    local secs=$1 name=$2; shift 2
    timeout -k 10 $secs python3 tests/soak/$name.py "$@" > $O/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a $O/rc.txt
    tail -1 $O/$name.log
    return $rc
}
step 400 fuzz_models 1500 ${2:-91} && step 400 fuzz_gauss 20000 $((${2:-91}+1)) && step 200 fuzz_reductions 12000 $((${2:-91}+2)) && step 300 fuzz_gibbs_n 25000 $((${2:-91}+3)) && step 300 fuzz_graph 20000 $((${2:-91}+4)) && step 400 fuzz_pairdist_big 600 $((${2:-91}+5)) && step 300 fuzz_contract 200 $((${2:-91}+6))
rc=$?
grep -c MISMATCH $O/*.log || true
exit $rc
