"""
Native fast paths of the polynomial + Gaussian likelihood: when a
``Likelihood`` is built from :class:`ForwardModel` (with numpy's ``polyval``)
and :class:`GaussianErrorModel`, its log-prob and gradient run in fused HIP
kernels and the ``[C x n_data]`` mock data is never written to HBM
(reference path: ``binf/pdf/likelihoods.py:141-155`` calling
``binf/example/likelihood.py:24-30,54-61``).
"""
import torch

from binf_amd import _native, native

KIND = 'poly'            # the name this module's hooks are registered under


def _usable(coeffs):
    return isinstance(coeffs, torch.Tensor) and coeffs.is_cuda and \
        coeffs.dtype == torch.float64 and coeffs.shape[-1] <= 64


def _as2d(x):
    return x if x.dim() == 2 else x.reshape(1, -1)


def log_prob(likelihood, fwm, em, fwm_vars, em_vars):
    fwm_vars = dict(fwm_vars)
    em_vars = dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    coeffs = fwm_vars.get('coefficients')
    if not _usable(coeffs) or 'precision' not in em_vars:
        return None
    dev = coeffs.device
    c2 = _as2d(coeffs).contiguous()
    xs, ys = fwm.xs_device(dev), em.ys_device(dev)
    if ys.numel() >= CHI2_MEMO_MIN_DATA and USE_CHI2_MEMO and c2.numel() * 8 <= (1 << 28):
        # a Horner pass over this much data is worth remembering per chain: the Gibbs
        # sweep asks for the same coefficients' chi^2 three times (proposal, precision
        # update, next E_before); the memo is checked on the device, bit for bit
        return _native.poly_gauss_logp_memo(c2, xs, ys, em_vars['precision'],
                                            _chi2_memo(xs, ys, c2.shape))
    return _native.poly_gauss_logp(c2, xs, ys, em_vars['precision'])


USE_CHI2_MEMO = True
CHI2_MEMO_MIN_DATA = 2048       # below, the pass is a few microseconds


def _chi2_memo(xs, ys, shape):
    """(memo_coeffs, memo_chi2, memo_state) for this data set, batch shape and stream
    (binf_amd/memo.py: weakly keyed by the data, capped by bytes; clones of a model share
    their device copies and therefore their memo, new data get a new one)."""
    from binf_amd import memo
    return memo.chi2_memo(ys, xs, shape, xs.device, _native.new_chi2_memo)


def gradient(likelihood, fwm, em, fwm_vars, em_vars):
    fwm_vars = dict(fwm_vars)
    em_vars = dict(em_vars)
    fwm._complete_variables(fwm_vars)
    em._complete_variables(em_vars)
    coeffs = fwm_vars.get('coefficients')
    if not _usable(coeffs) or 'precision' not in em_vars:
        return None
    dev = coeffs.device
    c2 = _as2d(coeffs).contiguous()
    out = _native.poly_gauss_grad(c2, fwm.design_matrix(c2.shape[1], dev),
                                  em.ys_device(dev), em_vars['precision'])
    return out if coeffs.dim() == 2 else out.reshape(-1)


def _is_poly_pair(likelihood):
    """Polynomial forward model + Gaussian error model, neither overridden?"""
    fs = getattr(likelihood.forward_model, 'native_spec', lambda: None)()
    es = getattr(likelihood.error_model, 'native_spec', lambda: None)()
    return fs is not None and es is not None and fs[0] == 'polynomial' and es[0] == 'gaussian'


FUSED_MAX_COEFFS = 16       # binf_hmc_sample_poly_f64 limits (include/binf_hip.h)
FUSED_MAX_DATA = 1024        # > 128: one wave per chain (csrc/hmc_poly_wave.hip)


def posterior_hmc_spec(posterior, variable_name):
    """Descriptor of the fused small-data trajectory kernel
    (``binf_hmc_sample_poly_f64``) if ``posterior`` is one it integrates:
    a conditional posterior whose only free variable is ``coefficients``, made
    of ONE polynomial + Gaussian-error likelihood with the precision fixed, at
    most one :class:`GaussianPrior` on the coefficients, and components without
    free variables (constants of the energy) before and / or one after them in
    the posterior's summation order.  Otherwise None (per-step tier).

    ``('poly', forward_model, error_model, precision, prior or None,
    prior_first, [constants before], constant after or None)``"""
    from binf_amd.example.priors import GaussianPrior
    from binf_amd.pdf.likelihoods import Likelihood
    if variable_name != 'coefficients':
        return None
    lik = prior = None
    kinds, consts = [], []
    for f in posterior._ordered_components():
        if len(f.variables) == 0:
            kinds.append('c')
            consts.append(f)
        elif set(f.variables) == {variable_name} and isinstance(f, Likelihood) \
                and lik is None and _is_poly_pair(f) \
                and 'precision' in f.error_model.parameters:
            lik = f
            kinds.append('lik')
        elif set(f.variables) == {variable_name} and isinstance(f, GaussianPrior) \
                and prior is None:
            prior = f
            kinds.append('prior')
        else:
            return None
    n_data = len(lik.error_model.ys) if lik is not None else 0
    if lik is None or n_data > FUSED_MAX_DATA or \
            (n_data > 128 and _native.pairwise_tree_height(n_data) > 3):
        return None
    theta = [i for i, k in enumerate(kinds) if k != 'c']
    n_pre, n_post = theta[0], len(kinds) - 1 - theta[-1]
    if theta[-1] - theta[0] + 1 != len(theta) or n_post > 1:
        return None                  # a constant between the two, or two after
    return (KIND, lik.forward_model, lik.error_model,
            lik.error_model['precision'].value, prior,
            prior is not None and kinds.index('prior') < kinds.index('lik'),
            consts[:n_pre], consts[n_pre] if n_post else None)


def _params(fn):
    """A recogniser that returns the whole spec ``(kind, *params)`` as a registry
    match hook (which returns the params)."""
    def hook(pdf, variable_name):
        spec = fn(pdf, variable_name)
        return None if spec is None else tuple(spec[1:])
    return hook


# ---------------------------------------------------------------------------
# HMCSampler hooks of the kind (binf_amd/native.py)
# ---------------------------------------------------------------------------
_POLY_WAVE_MAX_WORK = 2.0e8    # chains x data points x coefficients (see covers)
# from this many chains on the fused transition lays a chain out on ONE lane (see lane_layout)
LANE_MIN_CHAINS = 65536


def covers(sampler, spec, D, C=None):
    """Is the fused small-data transition the right launch for ``D`` coefficients
    (and, when given, ``C`` chains)?  ``sampler.fused_transition``: True (layout by the
    batch), 'group' / 'lane' (one layout whatever the batch), 'always' (fused even where
    the per-step tier is faster) or False."""
    mode = getattr(sampler, 'fused_transition', True)
    if not mode or D > FUSED_MAX_COEFFS:
        return False
    n_data = len(spec[2].ys)
    if mode == 'lane' and n_data > 128:
        return False                 # one lane per chain covers <= 128 data points
    if n_data > 128 and C is not None and mode != 'always' and \
            float(C) * n_data * D > _POLY_WAVE_MAX_WORK:
        # one wave per chain wins while the batch is launch-bound (3-10x up to
        # ~1e8 chain x data x coefficient products, scripts/probe_poly_wave.py);
        # beyond that the MFMA gradient of the per-step tier is faster
        return False
    return True


def lane_layout(sampler, spec, C):
    """One lane per chain (csrc/hmc_poly.hip) instead of a lane group
    (csrc/poly_chain_kernel.hpp) for the fused polynomial transition?  A lane
    group fills the chip from a few thousand chains (20.7 vs 42 us per
    transition at 4096 chains); with LANE_MIN_CHAINS chains and more every
    SIMD has work either way and one lane per chain does half the instructions
    (2^20 chains: 0.5 vs ~2 ms).  The energies are the same bits in both; the
    force is summed in data order vs partial sums + butterfly, so a chain's
    trajectory differs at rounding level between batches on either side of the
    threshold (like the MFMA gradient's batch-dependent order)."""
    mode = getattr(sampler, 'fused_transition', True)
    if mode == 'lane':
        return True
    if mode in ('group', False):
        return False
    return len(spec[2].ys) <= 128 and C >= LANE_MIN_CHAINS


def hmc_sample(sampler, spec, q0, p0, u, accepted, adapt):
    """The example's polynomial posterior with a small data set: the whole
    transition in one launch (``csrc/hmc_poly.hip``)."""
    from binf_amd.samplers.hmc import _MODES
    _, fwm, em, precision, prior, prior_first, pre, post = spec
    C, K = q0.shape
    dev = q0.device

    def const_term(f):
        v = f.log_prob()
        if not isinstance(v, torch.Tensor):
            return torch.full((C,), float(v), dtype=torch.float64, device=dev)
        return v.to(device=dev, dtype=torch.float64).reshape(-1).expand(C).contiguous()
    # numpy.sum of a short list: left to right (one launch for two and more terms)
    pre_terms = [const_term(f) for f in pre]
    lp_pre = None if not pre_terms else \
        (pre_terms[0] if len(pre_terms) == 1 else _native.sum_terms(pre_terms))
    lp_post = const_term(post) if post is not None else None
    means = prior._vec('means', dev) if prior is not None else None
    variances = prior._vec('variances', dev) if prior is not None else None
    q_out = torch.empty_like(q0)
    eb = torch.empty(C, dtype=torch.float64, device=dev)
    ea = torch.empty(C, dtype=torch.float64, device=dev)
    _native.hmc_sample_poly(q0, p0, u, q_out, accepted, sampler.n_accepted, eb, ea,
                            fwm.xs_device(dev), em.ys_device(dev), precision,
                            means, variances, prior_first, lp_pre, lp_post,
                            sampler._timestep, sampler._dt_chain, sampler.leapfrog_steps, adapt,
                            sampler.adaption_uprate, sampler.adaption_downrate,
                            _MODES[sampler.mode] | (_native.MODE_LANE_PER_CHAIN
                                                    if lane_layout(sampler, spec, C) else 0))
    sampler.last_e_before, sampler.last_e_after = eb, ea
    return q_out


def hmc_n(sampler, spec, n, thin, p0, u, record, out, q0, shape):
    """``hmc_n`` hook: the multi-sweep kernel with the precision draw switched off, where
    it applies (lane-group layout, a shape the fused transition covers)."""
    C, K = q0.shape
    if lane_layout(sampler, spec, C) or not covers(sampler, spec, K, C):
        return False, None
    return hmc_sample_n(sampler, spec, n, thin, p0, u, record, out, q0)


# ---------------------------------------------------------------------------
# fused _leapfrog: gradient, partial-sum reduction, kick and drift of every step in
# one launch each (binf_poly_leapfrog_f64)
# ---------------------------------------------------------------------------
def posterior_leapfrog_spec(posterior, variable_name):
    """``(KIND, forward_model, error_model, precision)`` if the force on
    ``variable_name`` is exactly ONE polynomial + Gaussian-error likelihood with
    its precision fixed (every other component has no differentiable variable,
    quirk Q4) -- the example's conditional posterior of the coefficients."""
    from binf_amd.pdf.likelihoods import Likelihood
    if variable_name != 'coefficients':
        return None
    lik = None
    for f in posterior._ordered_components():
        if not (len(f.variables) > 0 and len(f.differentiable_variables) > 0):
            continue
        if lik is not None or not isinstance(f, Likelihood) or not _is_poly_pair(f) or \
                set(f.variables) != {variable_name} or \
                'precision' not in f.error_model.parameters:
            return None
        lik = f
    if lik is None:
        return None
    return (KIND, lik.forward_model, lik.error_model, lik.error_model['precision'].value)


def leapfrog(sampler, spec, q2, p2, dt, dtc, nsteps, mode, q_from):
    if not (q2.is_cuda and q2.shape[1] <= 64):
        return False
    _, fwm, em, precision = spec
    if q_from is not None:
        q2.copy_(q_from)
    _native.poly_leapfrog(q2, p2, fwm.design_matrix(q2.shape[1], q2.device),
                          em.ys_device(q2.device), precision, dt, dtc, nsteps, mode)
    return True


# ---------------------------------------------------------------------------
# n Gibbs sweeps in one launch (csrc/gibbs_poly.hip)
# ---------------------------------------------------------------------------
def _device_rng_of(obj):
    """The DeviceRNG behind a sampler's draw source (the generator itself, or
    a bound ``gamma`` method of one), else None."""
    from binf_amd.samplers.rng import DeviceRNG
    obj = getattr(obj, '__self__', obj)
    return obj if type(obj) is DeviceRNG else None


def _same_data(a, b):
    """Clones of a model share their device copies: identity settles it without a
    comparison on the device (and the synchronisation that would cost every sweep)."""
    return a is b or (a.shape == b.shape and torch.equal(a, b))


def _gibbs_structure(cs, ps, C, K, dev, hmc):
    """The static part of :func:`gibbs_sample_n`'s recognition: ``(forward model,
    error model, GaussianPrior or None, prior_first, gp_where, GammaPrior term or None,
    the precision sampler's GammaPrior)`` or False."""
    from binf_amd.example.priors import GammaPrior
    spec = posterior_hmc_spec(cs.pdf, 'coefficients')
    if spec is None or K > FUSED_MAX_COEFFS:
        return False
    _, fwm, em, _, prior, prior_first, pre, post = spec
    consts = list(pre) + ([post] if post is not None else [])
    if len(consts) > 1 or any(type(f) is not GammaPrior or
                              set(f._original_variables) != {'precision'} for f in consts):
        return False
    gp_where = 0 if not consts else (1 if pre else 2)
    gp = consts[0] if consts else None
    if hmc:
        if cs._variable_name != 'coefficients' or lane_layout(cs, spec, C) or \
                not covers(cs, spec, K, C):
            return False
    # the precision sampler must look at the same data
    try:
        em_p = ps.pdf.likelihoods['points'].error_model
        fwm_p = ps.pdf.likelihoods['points'].forward_model
        gprior = ps._get_prior()
    except (KeyError, IndexError, NotImplementedError, AttributeError):
        return False
    if getattr(fwm_p, 'native_spec', lambda: None)() is None or \
            getattr(em_p, 'native_spec', lambda: None)() is None or \
            not _same_data(em_p.ys_device(dev), em.ys_device(dev)) or \
            not _same_data(fwm_p.xs_device(dev), fwm.xs_device(dev)):
        return False
    return fwm, em, prior, prior_first, gp_where, gp, gprior


def gibbs_sample_n(gibbs, n, thin, record):
    """``n`` sweeps of ``gibbs`` in ONE launch of ``binf_gibbs_poly_sample_n_f64``
    if it is the example's scheme -- variables ``coefficients`` (an
    :class:`HMCSampler` or :class:`RWMCSampler` on a conditional posterior the
    fused polynomial kernel integrates) and ``precision`` (a
    :class:`GammaSampler`) -- with draw sources the kernel can reproduce: all
    :class:`DeviceRNG` (generated in the kernel, the same values n single sweeps
    draw) or all the reference's host ``np.random`` stream (drawn on the host in
    the reference's order, uploaded once).  Returns ``(handled, records)``;
    ``handled`` False = not this scheme, nothing was touched."""
    import numpy as np

    from binf_amd.example.priors import GammaPrior
    from binf_amd.example.samplers import GammaSampler, RWMCSampler
    from binf_amd.samplers.hmc import HMCSampler, _MODES
    from binf_amd.samplers.rng import HostLegacyRNG

    subs = gibbs.subsamplers
    if sorted(gibbs.pdf.variables) != ['coefficients', 'precision'] or \
            set(subs) != {'coefficients', 'precision'}:
        return False, None
    cs, ps = subs['coefficients'], subs['precision']
    if type(ps) is not GammaSampler or type(cs) not in (HMCSampler, RWMCSampler):
        return False, None
    state = gibbs.state.variables
    theta, tau = state['coefficients'], state['precision']
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dim() == 2 and
            theta.dtype == torch.float64 and isinstance(tau, torch.Tensor) and tau.is_cuda and
            tau.dtype == torch.float64 and tau.dim() == 1 and tau.numel() == theta.shape[0]):
        return False, None
    C, K = theta.shape
    dev = theta.device
    hmc = type(cs) is HMCSampler
    # the walk over the two conditional posteriors (what they are made of, whether the
    # fused kernel integrates them, whether both look at the same data) is remembered
    # per sampler pair and batch shape: a sweep per launch would otherwise spend more
    # time recognising itself than running
    key = (id(cs), id(ps), id(cs.pdf), id(ps.pdf), C, K, dev,
           getattr(cs, 'fused_transition', None), getattr(cs, '_variable_name', None))
    cache = gibbs.__dict__.setdefault('_fused_structure', {})
    st = cache.get(key)
    if st is None:
        st = _gibbs_structure(cs, ps, C, K, dev, hmc)
        cache.clear()
        cache[key] = st
    if st is False:
        return False, None
    fwm, em, prior, prior_first, gp_where, gp, gprior = st
    gamma_shape = 0.5 * len(em.ys) + gprior.shape - 1              # samplers.py:27-32

    # ---- draw sources -------------------------------------------------------
    # Nothing is consumed for good before the launch has been accepted: the device
    # generators' positions are advanced after it, and a failed launch puts the host
    # stream back where it was (a retry, or the per-variable loop, then draws what
    # the unfused run draws).
    rng_c = _device_rng_of(cs.rng) if cs.rng is not None else None
    rng_g = _device_rng_of(ps.gamma) if ps.gamma is not None else None
    host_c = (cs.rng is None) if not hmc else type(cs.rng) is HostLegacyRNG
    host_g = ps.gamma is None
    p0 = u = g = streams = None
    zig = True
    coff = 0
    advance = []                    # (generator, positions) to take once the launch is in
    host_state = None
    if rng_c is not None and rng_g is not None:
        if gamma_shape < 1.0 or rng_c.chain_offset != rng_g.chain_offset:
            return False, None
        coff = rng_c.chain_offset
        zig = rng_c._normal_kind == 'normal_zig'
        if rng_c is rng_g:
            o = rng_c.offset             # per sweep: momentum / step, acceptance, gamma (+128)
            streams = ((rng_c.seed, o, 130), (rng_c.seed, o + 1, 130), (rng_c.seed, o + 2, 130))
            advance = [(rng_c, 130 * n)]
        else:
            streams = ((rng_c.seed, rng_c.offset, 2), (rng_c.seed, rng_c.offset + 1, 2),
                       (rng_g.seed, rng_g.offset, 128))
            advance = [(rng_c, 2 * n), (rng_g, 128 * n)]
    elif host_c and host_g:
        # the global legacy stream in the order n sweeps consume it
        host_state = np.random.get_state()
        hp = np.empty((n, C, K))
        hu = np.empty((n, C))
        hg = np.empty((n, C))
        for i in range(n):
            if hmc:
                hp[i] = np.random.normal(size=(C, K))                # hmc.py:146
                hu[i] = np.random.uniform(size=C)                    # hmc.py:151
            else:
                hp[i] = np.random.uniform(low=-cs.stepsize, high=cs.stepsize, size=(C, K))
                hu[i] = np.random.random(size=C)                     # samplers.py:80,86
            hg[i] = np.random.gamma(gamma_shape, size=C)             # samplers.py:47
        p0, u, g = (torch.from_numpy(x).to(dev) for x in (hp, hu, hg))
    else:
        return False, None

    nrec = n // thin
    rec_c = torch.empty((nrec, C, K), dtype=torch.float64, device=dev) if record and nrec else None
    rec_t = torch.empty((nrec, C), dtype=torch.float64, device=dev) if record and nrec else None
    theta_out = torch.empty_like(theta)
    tau_out = torch.empty_like(tau)
    accepted = torch.empty((n, C), dtype=torch.uint8, device=dev)
    kw = {}
    if hmc:
        n_adapt = max(0, min(n, cs.timestep_adaption_limit - 1 - cs.counter))
        if n_adapt > 0 and cs._dt_chain is None:
            cs._dt_chain = torch.full((C,), float(cs._timestep), dtype=torch.float64, device=dev)
        if not isinstance(cs.n_accepted, torch.Tensor):
            cs.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)
        eb = torch.empty((n, C), dtype=torch.float64, device=dev)
        ea = torch.empty((n, C), dtype=torch.float64, device=dev)
        kw = dict(move=_native.MOVE_HMC, mode=_MODES[cs.mode], nsteps=cs.leapfrog_steps,
                  timestep=cs._timestep, dt_chain=cs._dt_chain, n_adapt=n_adapt,
                  uprate=cs.adaption_uprate, downrate=cs.adaption_downrate,
                  n_accepted=cs.n_accepted, e_before=eb, e_after=ea)
    else:
        if not isinstance(cs._n_accepted_moves, torch.Tensor):
            cs._n_accepted_moves = torch.zeros(C, dtype=torch.int64, device=dev)
        kw = dict(move=_native.MOVE_RWMC, stepsize=cs.stepsize, n_accepted=cs._n_accepted_moves)
    try:
        _native.gibbs_poly_sample_n(
            theta.contiguous(), tau.contiguous(), theta_out, tau_out, fwm.xs_device(dev),
            em.ys_device(dev), n, thin,
            prior_means=prior._vec('means', dev) if prior is not None else None,
            prior_vars=prior._vec('variances', dev) if prior is not None else None,
            prior_first=prior_first, gp_where=gp_where,
            gp_shape=gp.shape if gp is not None else 1.0, gp_rate=gp.rate if gp is not None else 0.0,
            gamma_shape=gamma_shape, gamma_rate=gprior.rate, rec_coefficients=rec_c,
            rec_precision=rec_t, accepted=accepted, p0=p0, u=u, g=g, streams=streams,
            chain_offset=coff, zig=zig, **kw)
    except Exception:
        if host_state is not None:
            np.random.set_state(host_state)
        raise
    for gen, positions in advance:
        gen.offset += positions

    # ---- what n single sweeps would have left behind ---------------------------
    flags = accepted.view(torch.bool)
    if hmc:
        cs.last_e_before, cs.last_e_after = eb[-1], ea[-1]
        cs._last_move_accepted = flags[-1]
        cs.accepted_history = flags
        cs.counter += n
    else:
        cs.last_move_accepted = flags[-1]
        cs._n_moves += n
    cs.state = theta_out
    ps.state = tau_out
    gibbs._update_state(coefficients=theta_out, precision=tau_out)
    return True, ({'coefficients': rec_c, 'precision': rec_t} if rec_c is not None else None)


def hmc_sample_n(sampler, spec, n, thin, p0, u, record, out, q0):
    """``HMCSampler.sample_n`` on the example's conditional posterior of the
    coefficients (precision fixed): n transitions in ONE launch of the multi-sweep
    kernel with the precision draw switched off (``keep_precision``), bit-identical to
    n ``sample()`` calls.  Returns ``(handled, records)``; not handled = a structure or
    draw source the launch does not reproduce (the caller loops over ``sample()``)."""
    import numpy as np

    from binf_amd.example.priors import GammaPrior
    from binf_amd.samplers.hmc import _MODES
    from binf_amd.samplers.rng import HostLegacyRNG

    _, fwm, em, precision, prior, prior_first, pre, post = spec
    consts = list(pre) + ([post] if post is not None else [])
    if len(consts) > 1 or any(type(f) is not GammaPrior or
                              set(f._original_variables) != {'precision'} for f in consts):
        return False, None
    gp = consts[0] if consts else None
    if gp is not None:
        # the constant must be evaluated at the likelihood's own precision
        pv = gp['precision'].value
        if pv is not precision and not (np.isscalar(pv) and np.isscalar(precision) and pv == precision):
            return False, None
    C, K = q0.shape
    dev = q0.device
    if isinstance(precision, torch.Tensor):
        if not (precision.is_cuda and precision.dtype == torch.float64 and precision.numel() == C):
            return False, None
        tau = precision.reshape(-1).contiguous()
    else:
        tau = torch.full((C,), float(precision), dtype=torch.float64, device=dev)
    rng = sampler.rng
    dev_rng = _device_rng_of(rng)
    streams = None
    zig = True
    coff = 0
    take = 0
    host_state = None
    if p0 is not None and u is not None:
        p0 = p0.reshape(n, C, K).contiguous()
        u = u.reshape(n, C).contiguous()
    elif p0 is None and u is None and dev_rng is not None:
        o = dev_rng.offset                      # per sample(): normal, then uniform
        streams = ((dev_rng.seed, o, 2), (dev_rng.seed, o + 1, 2), (0, 0, 0))
        take = 2 * n                            # ... taken once the launch is in (see gibbs_sample_n)
        zig = dev_rng._normal_kind == 'normal_zig'
        coff = dev_rng.chain_offset
    elif p0 is None and u is None and type(rng) is HostLegacyRNG:
        host_state = np.random.get_state()
        hp, hu = np.empty((n, C, K)), np.empty((n, C))
        for i in range(n):
            hp[i] = np.random.normal(size=(C, K))                  # hmc.py:146
            hu[i] = np.random.uniform(size=C)                      # hmc.py:151
        p0, u = torch.from_numpy(hp).to(dev), torch.from_numpy(hu).to(dev)
    else:
        return False, None
    nrec = n // thin
    samples = None
    if record and nrec > 0:
        samples = out.view(nrec, C, K) if out is not None else \
            torch.empty((nrec, C, K), dtype=torch.float64, device=dev)
    n_adapt = max(0, min(n, sampler.timestep_adaption_limit - 1 - sampler.counter))
    if n_adapt > 0 and sampler._dt_chain is None:
        sampler._dt_chain = torch.full((C,), float(sampler._timestep), dtype=torch.float64, device=dev)
    if not isinstance(sampler.n_accepted, torch.Tensor):
        sampler.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)
    accepted = torch.empty((n, C), dtype=torch.uint8, device=dev)
    eb = torch.empty((n, C), dtype=torch.float64, device=dev)
    ea = torch.empty((n, C), dtype=torch.float64, device=dev)
    q_out = torch.empty_like(q0)
    try:
        _native.gibbs_poly_sample_n(
            q0, tau, q_out, torch.empty_like(tau), fwm.xs_device(dev), em.ys_device(dev), n, thin,
            move=_native.MOVE_HMC, mode=_MODES[sampler.mode], nsteps=sampler.leapfrog_steps,
            timestep=sampler._timestep, dt_chain=sampler._dt_chain, n_adapt=n_adapt,
            uprate=sampler.adaption_uprate, downrate=sampler.adaption_downrate,
            prior_means=prior._vec('means', dev) if prior is not None else None,
            prior_vars=prior._vec('variances', dev) if prior is not None else None,
            prior_first=prior_first, gp_where=0 if gp is None else (1 if pre else 2),
            gp_shape=gp.shape if gp is not None else 1.0, gp_rate=gp.rate if gp is not None else 0.0,
            rec_coefficients=samples, accepted=accepted, n_accepted=sampler.n_accepted,
            e_before=eb, e_after=ea, p0=p0, u=u, streams=streams, chain_offset=coff, zig=zig,
            keep_precision=True)
    except Exception:
        if host_state is not None:
            np.random.set_state(host_state)
        raise
    if take:
        dev_rng.offset += take
    flags = accepted.view(torch.bool)
    sampler.last_e_before, sampler.last_e_after = eb, ea
    sampler._last_move_accepted = flags[-1]
    sampler.accepted_history = flags
    sampler.counter += n
    return True, (q_out, samples)


native.register(
    KIND, replace=True,
    match_hmc=_params(posterior_hmc_spec),
    match_leapfrog=_params(posterior_leapfrog_spec),
    covers=covers, hmc=hmc_sample, hmc_n=hmc_n, leapfrog=leapfrog,
    gibbs=gibbs_sample_n,
    likelihood={('polynomial', 'gaussian'): (log_prob, gradient)},
    extras={'lane_layout': lane_layout})
