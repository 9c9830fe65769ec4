# the self-launching bench with two ranks sharing the one GPU of the box (gloo for
# the control plane: RCCL refuses two ranks on one device)
BINF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r02_k_gpus2.json 2> gpurun_out/r02_k_gpus2.err; echo rc=$?
cat gpurun_out/r02_k_gpus2.json | cut -c1-900; tail -3 gpurun_out/r02_k_gpus2.err
