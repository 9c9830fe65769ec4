# Round 2, first GPU call: GPU suite, the driver's exact bench command, the
# launcher's device-count error, kernel trace of the bench, generic-tier trace.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_a
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_tail.txt || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
python3 bench.py --gpus 2 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" | tee $O/bench_gpus2.rc; cat $O/bench_gpus2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-other-mode --no-extra --no-pmc > $O/bench_under_rocprof.json 2>$O/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_generic -- python3 $R/scripts/bench_generic.py > $O/bench_generic.json 2>$O/rocprof_generic.err
cd $R
python3 scripts/latency_probe.py > $O/latency.txt 2>&1
cat $O/bench_generic.json $O/latency.txt
