# Evidence for a round, in ONE GPU call: full GPU suite, smoke, the driver's bench
# command, the same under rocprofv3 --kernel-trace --stats, HBM counters in
# separate --pmc passes (also measured live by bench.py), bench variants, the C3 /
# C5 kernels with their own kernel stats and counters, the fused-generator path.
# Usage: bash scripts/final_profile.sh <tag>     then: python scripts/collect_profile.py <tag>
set -o pipefail
TAG=${1:-r02_z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_tail.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json
python3 bench.py --fuse 1 --steps 400 --warmup 100 --no-cpu-baseline --no-extra --no-pmc --no-other-mode > $O/bench_fuse1.json 2>/dev/null
python3 bench.py --mode fma --no-cpu-baseline --no-extra --no-other-mode > $O/bench_fma.json 2>/dev/null
python3 bench.py --thin 64 --no-cpu-baseline --no-extra --no-other-mode > $O/bench_thin64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-other-mode --no-extra --no-pmc --no-single-call --sustain-ms 0 > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --pmc-child --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --pmc-child --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_valu -- python3 $R/bench.py --pmc-child --steps 100 --warmup 20 > /dev/null 2>&1
# fused generator (draws generated in the kernel): kernel stats + VALU counters
cat > /tmp/e2e.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from binf_amd.pdf import IsotropicGaussian
from binf_amd.samplers.hmc import HMCSampler
from binf_amd.samplers.rng import DeviceRNG
dev = torch.device('cuda:0')
s = HMCSampler(IsotropicGaussian(), torch.zeros((4096, 1024), dtype=torch.float64, device=dev), 0.05, 20,
               variable_name='x', rng=DeviceRNG(0, dev))
buf = torch.empty((64, 4096, 1024), dtype=torch.float64, device=dev)
for _ in range(120): s.sample_n(64, out=buf)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -- python3 /tmp/e2e.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_e2e -- python3 /tmp/e2e.py > /dev/null 2>&1
# C3 / C4 (polynomial model)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_poly -- python3 $R/scripts/bench_poly.py > $O/bench_poly.json 2>/dev/null
cat > /tmp/polyk.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from binf_amd import _native
from binf_amd.example.likelihood import POLYVAL, ForwardModel
dev = torch.device('cuda:0'); C, K, N = 8192, 33, 16384
xs = np.linspace(-1, 1, N); ys = np.random.RandomState(9).standard_normal(N)
q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
fwm = ForwardModel(xs, POLYVAL); A = fwm.design_matrix(K, dev); ty = torch.from_numpy(ys).to(dev)
for _ in range(10): _native.poly_gauss_grad(q0, A, ty, 2.5)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_poly -- python3 /tmp/polyk.py > /dev/null 2>&1
# the same counters for the GENERAL gradient kernel (what the whole-tile kernel trimmed away)
BINF_POLY_GRAD_GENERAL=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_poly_general -- python3 /tmp/polyk.py > /dev/null 2>&1
# what the chip sustains on FP64 MFMA, and what an instruction beside one costs (C3's ceiling)
[ -x $R/scripts/mfma64_duty ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/mfma64_duty.hip -o $R/scripts/mfma64_duty
timeout -k 10 300 $R/scripts/mfma64_duty 1.0 > $O/mfma64_duty.jsonl
# the multi-rank control flow with the C4 / C5 legs: 2 gloo ranks sharing this GPU (a rehearsal)
(cd $R && BINF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err)
(cd $R && BINF_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --scaling strong --no-cpu-baseline > $O/bench_gloo2_strong.json 2>> $O/bench_gloo2.err)
# ... and the same control flow through RCCL itself with the one rank this box allows
(cd $R && BINF_BENCH_FORCE_DIST=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_force_dist_rccl.json 2> $O/bench_force_dist.err)
# the per-step tier on a user's torch PDF, eager launches and HIP-graph replay
(cd $R && python3 scripts/bench_generic.py > $O/bench_generic.json 2>/dev/null)
# C5 (pair-distance model)
for C in 256 2048; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dist_$C -- python3 $R/scripts/bench_distance.py $C > $O/bench_distance_$C.json 2>/dev/null
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/pmc_dist_$C -- python3 $R/scripts/bench_distance.py $C > /dev/null 2>&1
done
cd $R
python3 scripts/summarize_profile.py $O
