set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_f
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest_tail.txt || exit 1
python3 scripts/bench_generic_kernels.py | tee $O/generic_kernels.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_generic -- python3 $R/scripts/bench_generic_kernels.py > /dev/null 2>$O/rocprof.err
