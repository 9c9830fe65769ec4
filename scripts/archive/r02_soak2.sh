set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_soak2
mkdir -p $O
timeout -k 10 560 python3 tests/soak/fuzz_gauss.py 30000 21 > $O/fuzz_gauss.log 2>&1; echo "fuzz_gauss rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_gauss.log
timeout -k 10 300 python3 tests/soak/fuzz_models.py 1500 22 > $O/fuzz_models.log 2>&1; echo "fuzz_models rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_models.log
timeout -k 10 200 python3 tests/soak/fuzz_reductions.py 10000 23 > $O/fuzz_reductions.log 2>&1; echo "fuzz_reductions rc=$?" | tee -a $O/rc.txt; tail -1 $O/fuzz_reductions.log
grep -c MISMATCH $O/*.log || true
