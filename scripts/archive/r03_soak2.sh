# second, longer soak at HEAD (other seeds)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_soak2
mkdir -p $O
timeout -k 10 400 python3 tests/soak/fuzz_gauss.py 14000 41 > $O/fuzz_gauss.log 2>&1; echo "fuzz_gauss rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_gauss.log
timeout -k 10 250 python3 tests/soak/fuzz_models.py 700 42 > $O/fuzz_models.log 2>&1; echo "fuzz_models rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_models.log
timeout -k 10 150 python3 tests/soak/fuzz_reductions.py 8000 43 > $O/fuzz_reductions.log 2>&1; echo "fuzz_reductions rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_reductions.log
timeout -k 10 300 python3 tests/soak/fuzz_gibbs_n.py 30000 44 > $O/fuzz_gibbs_n.log 2>&1; echo "fuzz_gibbs_n rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_gibbs_n.log
grep -c MISMATCH $O/*.log || true
