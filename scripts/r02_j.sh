set -o pipefail
for NT in 1 2; do for W in 768 1024; do
BINF_POLY_GRAD_NT=$NT BINF_POLY_GRAD_WGS=$W python3 scripts/probe_poly_grad.py 2>/dev/null | sed "s/^/NT=$NT /"
done; done
timeout -k 10 600 python -m pytest tests/test_gpu_poly.py tests/test_gpu_guards.py tests/test_gpu_statistics.py -m gpu -x -q 2>&1 | tail -4
