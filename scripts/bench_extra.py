"""Short sub-results carried under the `extra` key of bench.py's JSON line (so
they are timed by the driver's run, not only by builder runs): BASELINE configs
C3 (polynomial model, MFMA design-matrix gradient), C5 (pair-distance model,
FP64 VALU) and the C2 shape with the draws generated on the device inside the
timed region.  Not the headline; each takes a second or two.

All timings are HIP events on torch's current stream (the stream the C-ABI
launches go to)."""
import numpy as np
import torch

# 78.6 TFLOP/s: fp64 matrix (MFMA) datasheet peak; 39.3e12: FP64 vector
# lane-operations/s without FMA contraction (bench.py)
MFMA_F64_PEAK_TFLOPS = 78.6
VALU_PEAK_LANEOPS = 39.3e12
# What a bare loop of v_mfma_f64_16x16x4_f64 on random operands SUSTAINS on this chip
# (scripts/mfma64_duty.hip, profiles/r04_b_mfma64_duty.jsonl: 70.0 - 70.9 TFLOP/s at 2 - 4
# waves per SIMD, 2.39 GHz in-kernel clock, 70.9 - 71.6 cycles per MFMA and SIMD instead of
# the 64 the datasheet figure assumes; every VALU instruction beside it costs another 3.4 - 5.9)
MFMA_F64_SUSTAINED_TFLOPS = 70.8
# FP64 operations of one UNORDERED pair of the restraint force as the algorithm defines them
# (sqrt and divide one operation each): 3 sub (d = xi - xj), 3 mul + 2 add (r^2), 1 sqrt,
# 1 sub + 1 div + 1 mul (w = tau (r - y) / r), 3 mul (w d), 6 add (both beads) = 21
C5_ALGORITHMIC_OPS_PER_PAIR = 21.0
C5_ISA_OPS_PER_PAIR = 24.0        # VALU instructions the n <= 256 scheme issues per pair and lane


def _timed(fn, n, warm=2, settle_s=0.0):
    """settle_s: keep launching (untimed) for that long first -- the chip's clock /
    power controller needs ~50 ms of sustained load to converge."""
    import time
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < settle_s:
        fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def c3_polynomial(dev, C=8192, K=33, N=16384, L=20):
    """C3: K=33 coefficients, N=16384 data, 8192 chains (SURVEY 8(d):
    xs = linspace(-1, 1), tau = 2.5)."""
    from binf_amd import _native
    from binf_amd.example.likelihood import POLYVAL, ForwardModel, make_likelihood
    from binf_amd.example.priors import GammaPrior, GaussianPrior
    from binf_amd.pdf.posteriors import Posterior
    from binf_amd.samplers.hmc import HMCSampler
    from binf_amd.samplers.rng import DeviceRNG
    xs = np.linspace(-1, 1, N)
    c_true = np.random.RandomState(7).standard_normal(K)
    ys = POLYVAL(xs, c_true) + np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
    q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
    fwm = ForwardModel(xs, POLYVAL)
    A = fwm.design_matrix(K, dev)
    tx = fwm.xs_device(dev)
    ty = torch.from_numpy(ys).to(dev)
    t_grad = _timed(lambda: _native.poly_gauss_grad(q0, A, ty, 2.5), 20, settle_s=0.1)
    t_logp = _timed(lambda: _native.poly_gauss_logp(q0, tx, ty, 2.5), 20)
    flops = 4.0 * K * N * C                      # 4 K N per chain and gradient
    lik = make_likelihood(xs, ys, POLYVAL)
    post = Posterior({lik.name: lik},
                     {'precision_prior': GammaPrior(1.0, 0.2),
                      'coefficients_prior': GaussianPrior(np.zeros(K), np.ones(K) * 5)})
    cond = post.conditional_factory(precision=2.5)
    s = HMCSampler(cond, q0, 2e-4, L, variable_name='coefficients', rng=DeviceRNG(1, dev))
    t_hmc = _timed(s.sample, 10, warm=3)
    return {'workload': 'C3: polynomial K=%d, N=%d, %d chains, L=%d' % (K, N, C, L),
            'grad_kernel_ms': t_grad * 1e3,
            'grad_TFLOPs': flops / t_grad / 1e12,
            'mfma_frac': flops / t_grad / 1e12 / MFMA_F64_PEAK_TFLOPS,
            'mfma_frac_of_sustained': flops / t_grad / 1e12 / MFMA_F64_SUSTAINED_TFLOPS,
            'logp_kernel_ms': t_logp * 1e3,
            'hmc_sample_ms': t_hmc * 1e3,
            'chain_leapfrog_steps_per_s': C * L / t_hmc,
            'gradient_share_of_sample': (L + 1) * t_grad / t_hmc,
            'acceptance': float(s.acceptance_rate.mean())}


def c4_gibbs(dev, C=4096):
    """C4 per-GPU share (32768 chains over 8 GPUs): Gibbs-within-HMC, HMC on the
    coefficients + the conjugate precision update, through the class stack -- the
    SAME leg `bench.py --gpus N` runs on every rank (scripts/bench_legs.py: sharded
    start state and generators, every 5th state into a SampleStore), here with one rank."""
    from scripts import bench_legs
    comm = bench_legs.Comm(device=dev)
    r = bench_legs.run_leg(bench_legs.C4Leg(dev, comm, chains_per_gpu=C), comm, 20, thin=5)
    r['gibbs_sweep_ms'] = r['sweep_ms']
    return r


def c5_distance(dev, C=256, n=256, L=20):
    """C5: 3 x 256 coordinates; 256 chains = the per-GPU share of its 2048 chains on 8 GPUs
    (2048: the whole configuration on one GPU)."""
    from binf_amd import _native
    from binf_amd.example.distance import make_distance_likelihood
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    I, J = np.triu_indices(n, 1)
    d_true = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
    ys = np.abs(d_true + 0.05 * rs.standard_normal(n * (n - 1) // 2))
    x = torch.from_numpy(truth.reshape(-1)[None, :] +
                         0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    lik = make_distance_likelihood(ys, n)
    t_g = _timed(lambda: lik.gradient(coordinates=x, precision=4.0), 20, settle_s=0.1)
    # one force evaluation INSIDE the fused trajectory kernel: launch time against the
    # trajectory length (the target distances reach registers once per launch)
    ymat = lik.error_model.ymat_device(dev)
    packed = lik.error_model.ypacked_device(dev)     # as HMCSampler._leapfrog passes it
    q, p = x.clone(), torch.zeros_like(x)
    t_l = {}
    for nst in (1, L):
        t_l[nst] = _timed(lambda: _native.pairdist_leapfrog(q, p, ymat, 4.0, (0.05, 0.0), True,
                                                            1e-5, None, nst, packed=packed), 40, warm=5)
    t_e = (t_l[L] - t_l[1]) / (L - 1)
    del q, p
    # the sampling itself: the leg `bench.py --gpus N` runs on every rank, with one rank
    from scripts import bench_legs
    comm = bench_legs.Comm(device=dev)
    leg = bench_legs.run_leg(bench_legs.C5Leg(dev, comm, chains_per_gpu=C, n=n, L=L), comm,
                             100 if C <= 256 else 30, thin=20)
    t_h = leg['sweep_ms'] * 1e-3
    pairs = float(C) * n * (n - 1)                # ordered pairs per force evaluation
    return {'workload': 'C5%s: %d beads x 3, %d chains, L=%d' % (' share' if C < 2048 else ' on one GPU', n, C, L),
            'force_kernel_ms': t_g * 1e3,
            'leapfrog_kernel_ms': t_l[L] * 1e3,
            'force_eval_in_trajectory_us': t_e * 1e6,
            'pair_interactions_per_s': pairs / t_e,
            # 21 algorithm-defined FP64 operations per UNORDERED pair (sqrt, divide = 1 each)
            # against 39.3e12 lane-operations/s; the kernel issues 24 VALU instructions per
            # pair and lane (valu_issue_frac: pipe utilisation, not a roofline)
            'valu_frac': C5_ALGORITHMIC_OPS_PER_PAIR * 0.5 * pairs / t_e / VALU_PEAK_LANEOPS,
            'valu_issue_frac': C5_ISA_OPS_PER_PAIR * 0.5 * pairs / t_e / VALU_PEAK_LANEOPS,
            'hmc_sample_ms': t_h * 1e3,
            'chain_leapfrog_steps_per_s': leg['chain_leapfrog_steps_per_s'],
            'acceptance': leg['ranks'][0]['self_check']['acceptance'],
            'record_every': leg['record_every'], 'sweeps_timed': leg['sweeps_timed'],
            'timing': leg['timing']}


def c5_more_beads(dev, L=20):
    """The same model beyond the configured 256 beads, with the few chains structure inference
    runs with: the symmetric force as ring kernels / a wave per tile, chi^2 by chunks (round 4);
    HMCSampler.sample() through the class stack, device draws."""
    from binf_amd.example.distance import make_distance_likelihood
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.pdf.posteriors import Posterior
    from binf_amd.samplers.hmc import HMCSampler
    from binf_amd.samplers.rng import DeviceRNG
    out = {}
    for n, C in ((1024, 32), (1024, 1024), (2048, 16), (4096, 8)):
        rs = np.random.RandomState(0)
        truth = rs.standard_normal((n, 3)) * 2.0
        I, J = np.triu_indices(n, 1)
        ys = np.abs(np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1)) + 0.05 * rs.standard_normal(len(I)))
        x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
        lik = make_distance_likelihood(ys, n)
        prior = IsotropicGaussian(0.05, 0.0, name='coordinates_prior', variable_name='coordinates')
        cond = Posterior({lik.name: lik}, {prior.name: prior}).conditional_factory(precision=4.0)
        s = HMCSampler(cond, x, 0.001 if n <= 1024 else 0.0005, L, variable_name='coordinates', rng=DeviceRNG(0, dev))
        t = _timed(s.sample, 8, warm=3)
        out['%d beads x %d chains' % (n, C)] = {
            'hmc_sample_ms': t * 1e3, 'chain_leapfrog_steps_per_s': C * L / t,
            'unordered_pairs_per_s': C * (n * (n - 1) / 2.0) * (L + 1) / t,
            'acceptance': float(s.acceptance_rate.mean())}
    out['workload'] = ('pair-distance posterior beyond C5\'s 256 beads: HMCSampler.sample(), L=%d, device draws; '
                       'start of round 4: 9.7 / 50 / 100 / 426 ms' % L)
    return out


def c5_gibbs(dev, C=256, n=256, L=20):
    """C5's model inside the reference's Gibbs scheme: HMC on the coordinates + the conjugate
    Gamma draw of one precision per chain (example/distance.py: make_restraint_gibbs_sampler)."""
    from binf_amd.example.distance import make_distance_likelihood, make_restraint_gibbs_sampler
    from binf_amd.example.priors import GammaPrior
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.pdf.posteriors import Posterior
    from binf_amd.samplers import BinfState
    from binf_amd.samplers.rng import DeviceRNG
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    I, J = np.triu_indices(n, 1)
    ys = np.abs(np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1)) + 0.5 * rs.standard_normal(len(I)))
    lik = make_distance_likelihood(ys, n)
    post = Posterior({lik.name: lik},
                     {'coordinates_prior': IsotropicGaussian(0.05, 0.0, name='coordinates_prior',
                                                             variable_name='coordinates'),
                      'precision_prior': GammaPrior(1.0, 0.2)})
    start = BinfState({'coordinates': torch.from_numpy(truth.reshape(-1)[None, :] +
                                                       0.1 * rs.standard_normal((C, 3 * n))).to(dev),
                       'precision': torch.full((C,), 4.0, dtype=torch.float64, device=dev)})
    gips = make_restraint_gibbs_sampler(post, 0.002, L, start, rng=DeviceRNG(0, dev))
    t = _timed(gips.sample, 30, warm=5, settle_s=0.1)
    return {'workload': 'C5 share, Gibbs-within-HMC: %d beads x 3, %d chains, L=%d, one precision per chain'
                        % (n, C, L),
            'gibbs_sweep_ms': t * 1e3, 'chain_leapfrog_steps_per_s': C * L / t,
            'precision_mean': float(gips.state.variables['precision'].mean())}


def c2_device_rng(dev, C=4096, D=1024, L=20, F=64):
    """C2 with the momentum / uniform draws generated on the device INSIDE the
    timed region (hmc.py:146,151 are part of sample())."""
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler
    from binf_amd.samplers.rng import DeviceRNG
    out = {'workload': 'C2 shape, sample_n(%d), every state recorded, draws generated inside '
                       'the sampling kernel (in the timed region)' % F}
    rec = torch.empty((F, C, D), dtype=torch.float64, device=dev)     # preallocated record buffer
    for mode in ('exact', 'fma'):
        s = HMCSampler(IsotropicGaussian(), torch.zeros((C, D), dtype=torch.float64, device=dev),
                       0.05, L, variable_name='x', rng=DeviceRNG(0, dev), mode=mode)
        t = _timed(lambda: s.sample_n(F, out=rec), 12, warm=3, settle_s=0.15) / F
        out['%s_us_per_transition' % mode] = t * 1e6
        out['%s_chain_steps_per_s' % mode] = C * L / t
        del s
    del rec
    torch.cuda.empty_cache()
    return out


def c2_strong_scaling_shares(dev, D=1024, L=20, F=64):
    """SURVEY 8(e), secondary: 4096 chains in total over N GPUs leave 4096 / N per
    GPU -- measured here on one GPU (chains spread over 2 / 4 waves below 2048,
    csrc/hmc_gauss_split.hip); draws pre-generated in HBM as in the headline."""
    from binf_amd.pdf import IsotropicGaussian
    from binf_amd.samplers.hmc import HMCSampler
    out = {'workload': 'C2 shape, 4096 chains / N per GPU, sample_n(%d), every state recorded' % F}
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    for n_gpus in (2, 4, 8):
        C = 4096 // n_gpus
        s = HMCSampler(IsotropicGaussian(), torch.randn((C, D), dtype=torch.float64, device=dev,
                                                         generator=gen), 0.05, L, variable_name='x')
        p0 = [torch.randn((F, C, D), dtype=torch.float64, device=dev, generator=gen) for _ in range(2)]
        u = [torch.rand((F, C), dtype=torch.float64, device=dev, generator=gen) for _ in range(2)]
        rec = torch.empty((F, C, D), dtype=torch.float64, device=dev)
        i = [0]

        def step():
            i[0] += 1
            s.sample_n(F, p0=p0[i[0] % 2], u=u[i[0] % 2], out=rec)
        t = _timed(step, 8, warm=2, settle_s=0.1) / F
        out['N=%d' % n_gpus] = {'chains_per_gpu': C, 'us_per_transition': t * 1e6,
                                'chain_steps_per_s_per_gpu': C * L / t}
        del s, p0, u, rec
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------
# Roofline blocks of the sub-results: the kernel's average duration comes from
# rocprofv3 --kernel-trace --stats over a child run of `roofline_child` (started
# by bench.py before it touches the GPU), so every frac can be recomputed from
# the kernel_stats rows carried in the block.
# ---------------------------------------------------------------------------
ROOFLINE_KERNELS = {'C3': 'poly_grad_mfma', 'C5': 'pairdist_leapfrog_sym_kernel',
                    'C1_gibbs': 'poly_chain_kernel<4, false, true, 0>'}
C3_SHAPE = dict(C=8192, K=33, N=16384)
C5_SHAPE = dict(C=256, n=256, L=20)
C1_SHAPE = dict(C=4096, K=4, N=20, L=50, sweeps=200)


def roofline_child():
    """The kernels the roofline blocks price, launched back to back after a settle
    phase each (run under rocprofv3 --kernel-trace --stats)."""
    import time
    from binf_amd import _native
    from binf_amd.example.distance import make_distance_likelihood
    from binf_amd.example.likelihood import POLYVAL, ForwardModel
    dev = torch.device('cuda:0')

    def settled(fn, n, settle_s=0.3):
        t = time.perf_counter()
        while time.perf_counter() - t < settle_s:
            fn()
            torch.cuda.synchronize()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    # C3 gradient
    C, K, N = C3_SHAPE['C'], C3_SHAPE['K'], C3_SHAPE['N']
    xs = np.linspace(-1, 1, N)
    ys = POLYVAL(xs, np.random.RandomState(7).standard_normal(K)) + \
        np.random.RandomState(9).standard_normal(N) / np.sqrt(2.5)
    q0 = torch.from_numpy(np.random.RandomState(8).standard_normal((C, K))).to(dev)
    A = ForwardModel(xs, POLYVAL).design_matrix(K, dev)
    ty = torch.from_numpy(ys).to(dev)
    settled(lambda: _native.poly_gauss_grad(q0, A, ty, 2.5), 40)
    # C5 fused leapfrog
    C, n, L = C5_SHAPE['C'], C5_SHAPE['n'], C5_SHAPE['L']
    rs = np.random.RandomState(0)
    truth = rs.standard_normal((n, 3)) * 2.0
    I, J = np.triu_indices(n, 1)
    d = np.sqrt(np.sum((truth[I] - truth[J]) ** 2, axis=1))
    lik = make_distance_likelihood(np.abs(d + 0.05 * rs.standard_normal(len(d))), n)
    x = torch.from_numpy(truth.reshape(-1)[None, :] + 0.1 * rs.standard_normal((C, 3 * n))).to(dev)
    ymat = lik.error_model.ymat_device(dev)
    packed = lik.error_model.ypacked_device(dev)
    q, p = x.clone(), torch.zeros_like(x)
    settled(lambda: _native.pairdist_leapfrog(q, p, ymat, 4.0, (0.05, 0.0), True, 1e-5, None, L,
                                              packed=packed), 60)
    # C1 shape: the multi-sweep Gibbs launch
    gips = _c1_gibbs(dev, C1_SHAPE['C'])
    settled(lambda: gips.sample_n(C1_SHAPE['sweeps'], record=False), 6, settle_s=0.2)


def _c1_gibbs(dev, C, L=50):
    from binf_amd.example.misc import make_posterior
    from binf_amd.example.samplers import make_hmc_sampler
    from binf_amd.samplers import BinfState
    from binf_amd.samplers.rng import DeviceRNG
    np.random.seed(0)
    xs = np.linspace(-2, 2, 20)
    poly = np.polynomial.polynomial.polyval
    ys = np.random.normal(loc=poly(xs, np.array([2.0, -4.0, 1.0, 1.5])), scale=1.0 / np.sqrt(2.5))
    st = BinfState(dict(coefficients=torch.ones((C, 4), dtype=torch.float64, device=dev),
                        precision=torch.ones(C, dtype=torch.float64, device=dev)))
    return make_hmc_sampler(make_posterior(xs, ys, poly), 0.02, L, st, rng=DeviceRNG(0, dev))


def c1_gibbs(dev, C=4096, L=50, sweeps=200):
    """C1's shape (example_script.py: K = 4, 20 data points) batched: Gibbs-within-HMC,
    `sweeps` sweeps per launch (GibbsSampler.sample_n) against one sample() per sweep."""
    g1, g2 = _c1_gibbs(dev, C, L), _c1_gibbs(dev, C, L)
    t_loop = _timed(g1.sample, 100, warm=10)
    t_n = _timed(lambda: g2.sample_n(sweeps, record=False), 4, warm=1) / sweeps
    return {'workload': 'C1 shape batched: Gibbs-within-HMC, polynomial K=4, N=20, %d chains, L=%d' % (C, L),
            'sweep_us_one_sample_per_launch': t_loop * 1e6,
            'sweep_us_%d_sweeps_per_launch' % sweeps: t_n * 1e6,
            'chain_leapfrog_steps_per_s': C * L / t_n}


def _roofline(kstats, key, bound, work, peak, unit, what):
    """kstats: {kernel name: (calls, average ns)} from rocprofv3's kernel_stats.csv."""
    if not kstats:
        return None
    rows = {k: v for k, v in kstats.items() if ROOFLINE_KERNELS[key] in k}
    if not rows:
        return None
    name, (calls, avg_ns) = max(rows.items(), key=lambda kv: kv[1][0])
    achieved = work / (avg_ns * 1e-9) / 1e12
    return {'bound': bound, 'achieved': achieved, 'peak': peak, 'unit': unit,
            'frac': achieved / peak, 'traffic': None, 'kernel': name,
            'rocprof_avg_us': avg_ns * 1e-3, 'rocprof_calls': calls,
            'algorithmic_work_per_launch': work, 'work_is': what,
            'source': 'rocprofv3 --kernel-trace --stats of scripts/bench_extra.py --roofline-child, '
                      'run by bench.py in this invocation'}


def run_all(dev, kstats=None):
    res = {}
    for name, fn in (('C3', c3_polynomial), ('C4', c4_gibbs), ('C5', c5_distance),
                     ('C5_2048_chains', lambda d: c5_distance(d, C=2048)),
                     ('C5_gibbs', c5_gibbs),
                     ('C5_more_beads', c5_more_beads),
                     ('C1_gibbs', c1_gibbs),
                     ('C2_device_rng', c2_device_rng),
                     ('C2_strong_scaling_shares', c2_strong_scaling_shares)):
        try:
            res[name] = fn(dev)
        except Exception as e:                    # one failing sub-result does not hide the others
            res[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    s3, s5, s1 = C3_SHAPE, C5_SHAPE, C1_SHAPE
    blocks = {
        'C3': _roofline(kstats, 'C3', 'mfma', 4.0 * s3['K'] * s3['N'] * s3['C'], MFMA_F64_PEAK_TFLOPS,
                        'TFLOP/s', 'SURVEY 8(d): 4 K N flops per chain and gradient x %d chains '
                        '(one launch = one gradient of every chain)' % s3['C']),
        'C5': _roofline(kstats, 'C5', 'valu',
                        C5_ALGORITHMIC_OPS_PER_PAIR * 0.5 * s5['C'] * s5['n'] * (s5['n'] - 1) * (s5['L'] + 1),
                        VALU_PEAK_LANEOPS / 1e12, 'T lane-op/s',
                        '21 FP64 operations per unordered pair as the algorithm defines them (3 sub, '
                        '3 mul + 2 add, 1 sqrt, 1 sub + 1 div + 1 mul, 3 mul, 6 add; sqrt and divide one '
                        'each) x n (n - 1) / 2 pairs x %d chains x (L + 1) = %d force evaluations of one '
                        'fused leapfrog launch; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.  The kernel '
                        'ISSUES 24 VALU instructions per pair (Newton steps of the correctly rounded sqrt '
                        'and of the reciprocal included): valu_issue_frac' % (s5['C'], s5['L'] + 1)),
        # 13 lane-operations per data point and force evaluation (Horner K - 1, residual 2,
        # K accumulates + K - 1 powers) -- the arithmetic of the model, not the butterflies
        'C1_gibbs': _roofline(kstats, 'C1_gibbs', 'valu',
                              13.0 * s1['N'] * (s1['L'] + 1) * s1['C'] * s1['sweeps'],
                              VALU_PEAK_LANEOPS / 1e12, 'T lane-op/s',
                              '13 FP64 lane-operations per data point and force evaluation x 20 '
                              'points x (L + 1) = 51 evaluations x %d chains x %d sweeps per launch '
                              '(latency-bound at this batch: 512 waves for 1024 SIMDs)'
                              % (s1['C'], s1['sweeps'])),
    }
    if blocks['C3'] is not None:
        b = blocks['C3']
        b['mfma_sustained_ceiling'] = MFMA_F64_SUSTAINED_TFLOPS
        b['frac_of_sustained'] = b['achieved'] / MFMA_F64_SUSTAINED_TFLOPS
        b['sustained_ceiling_is'] = ('bare v_mfma_f64_16x16x4_f64 loop on random operands, best of 1-4 waves '
                                     'per SIMD: 70.8 TFLOP/s at a 2.39 GHz in-kernel clock (no DVFS give-back '
                                     'on FP64 MFMA; 71 cycles per MFMA and SIMD, not 64) -- '
                                     'scripts/mfma64_duty.hip, profiles/r04_b_mfma64_duty.jsonl')
    if blocks['C5'] is not None:
        b = blocks['C5']
        b['algorithmic_ops_per_pair'] = C5_ALGORITHMIC_OPS_PER_PAIR
        b['isa_valu_per_pair'] = C5_ISA_OPS_PER_PAIR
        b['valu_issue_frac'] = b['frac'] * C5_ISA_OPS_PER_PAIR / C5_ALGORITHMIC_OPS_PER_PAIR
    if blocks['C1_gibbs'] is not None:
        blocks['C1_gibbs']['note'] = ('latency-bound by the batch, not a kernel-quality figure: 4096 chains are '
                                      '512 waves for 1024 SIMDs (0.27 of the VALU rate at 2^20 chains)')
    for k, b in blocks.items():
        if b is not None and k in res and 'error' not in res[k]:
            res[k]['roofline'] = b
    return res


if __name__ == '__main__':
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if '--roofline-child' in sys.argv:
        roofline_child()
    else:
        print(json.dumps(run_all(torch.device('cuda:0'))))
