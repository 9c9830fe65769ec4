# Round-3 soak: the randomised differential tests over the kernels this round touched
# (DPP / lane-swap reduction trees in every np.sum tree, batched leaf sums, the lane-group
# polynomial transition, the multi-sweep Gibbs launch, the contraction).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_soak
mkdir -p $O
timeout -k 10 420 python3 tests/soak/fuzz_gauss.py 12000 31 > $O/fuzz_gauss.log 2>&1; echo "fuzz_gauss rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_gauss.log
timeout -k 10 300 python3 tests/soak/fuzz_models.py 800 32 > $O/fuzz_models.log 2>&1; echo "fuzz_models rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_models.log
timeout -k 10 200 python3 tests/soak/fuzz_reductions.py 8000 33 > $O/fuzz_reductions.log 2>&1; echo "fuzz_reductions rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_reductions.log
timeout -k 10 200 python3 tests/soak/fuzz_gibbs_n.py 8000 34 > $O/fuzz_gibbs_n.log 2>&1; echo "fuzz_gibbs_n rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_gibbs_n.log
grep -c MISMATCH $O/*.log || true
