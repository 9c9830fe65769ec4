set -o pipefail
python3 examples/gaussian_chains.py 2>&1 | tail -4 || exit 1
python3 examples/polynomial_fit.py --chains 512 --iterations 400 --burn-in 100 --thin 10 2>&1 | tail -4 || exit 1
python3 examples/distance_restraints.py 2>&1 | tail -4 || exit 1
