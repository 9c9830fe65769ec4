"""
The built-in fused kind ``'gauss'``: the isotropic Gaussian (``binf_amd.pdf.IsotropicGaussian``,
the reference's ``TestHO``, ``binf/pdf/__init__.py:163-191``) behind
``HMCSampler`` -- registered with ``binf_amd.native`` like any other kind (``HMCSampler`` itself names
no model).  The hooks launch the kernels of ``csrc/hmc_gauss*.hip``:

``sample``      one transition per launch (``hmc`` hook; ``binf_hmc_sample_gauss_f64`` or, beyond
                8192 dimensions, the chunked kernels)
``sample_n``    n transitions in ONE launch of the persistent kernel, state in registers between
                them (``hmc_n`` hook), draws supplied, taken from the sampler's generator in the
                order n ``sample()`` calls consume it, or made inside the kernel
``sample_rng``  a transition that draws for itself (``hmc_rng`` hook; a ``DeviceRNG`` whose draws
                are the kernels' lane streams)

Every function takes the sampler first (``s``) and keeps its bookkeeping as ``sample()`` would:
``n_accepted``, ``_dt_chain``, ``counter``, ``last_e_before`` / ``last_e_after``,
``accepted_history``.
"""
import torch

from binf_amd import _native, native

GAUSS = 'gauss'


def _modes():
    from binf_amd.samplers.hmc import _MODES
    return _MODES


def _fill(rng, kind, out):
    from binf_amd.samplers.hmc import _fill as fill
    fill(rng, kind, out)


def sample_n(s, spec, n, thin, p0, u, record, out, q0, shape):
    """``hmc_n`` hook of the isotropic Gaussian: ONE launch of the persistent
    kernel (state kept in registers between transitions; D <= 8192), the chunked
    kernels for longer chains.  Returns ``(True, (q_out, samples))``."""
    C, D = q0.shape
    dev = q0.device
    nrec = n // thin
    persist = _native.gauss_persist_covers(D)
    in_kernel = p0 is None and u is None and fused_rng(s, spec) and persist
    if in_kernel and not draws_in_kernel(s, C, D):
        # a small batch: the split kernel with the draws in HBM is the faster
        # launch, so the SAME lane-stream draws are written out first (a seed
        # identifies the draws whatever the batch size)
        p0, u = _native.hmc_gauss_rng_draws(n, C, D, s.rng.seed, s.rng.offset,
                                            dev, chain_offset=chain_offset(s))
        take_positions(s, n)             # taken once the launch is in
        in_kernel = False
    if persist and not in_kernel and (p0 is None or u is None):
        # the draws of n sample() calls in the order those calls consume the
        # generator: normal, uniform, normal, uniform, ... (hmc.py:146,151)
        dp = torch.empty((n, C, D), dtype=torch.float64, device=dev) if p0 is None else None
        du = torch.empty((n, C), dtype=torch.float64, device=dev) if u is None else None
        for i in range(n):
            if dp is not None:
                _fill(s.rng, 'normal', dp[i])
            if du is not None:
                _fill(s.rng, 'uniform', du[i])
        p0 = dp if p0 is None else p0
        u = du if u is None else u
    if not persist:
        return True, sample_n_long(s, spec, n, thin, p0, u, record, out, q0, shape, nrec)

    _, k, x0 = spec
    n_adapt = max(0, min(n, s.timestep_adaption_limit - 1 - s.counter))
    if n_adapt > 0 and s._dt_chain is None:
        s._dt_chain = torch.full((C,), float(s._timestep),
                                    dtype=torch.float64, device=dev)
    if not isinstance(s.n_accepted, torch.Tensor):
        s.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)
    q_out = torch.empty_like(q0)
    if out is not None:
        samples = out.view(nrec, C, D)
    else:
        samples = torch.empty((nrec, C, D), dtype=torch.float64, device=dev) \
            if (record and nrec > 0) else None
    accepted = torch.empty((n, C), dtype=torch.uint8, device=dev)
    eb = ea = None
    if s.record_energies:
        eb = torch.empty((n, C), dtype=torch.float64, device=dev)
        ea = torch.empty((n, C), dtype=torch.float64, device=dev)
    if in_kernel:
        _native.hmc_sample_n_gauss_rng(q0, q_out, samples, accepted, s.n_accepted,
                                       eb, ea, s._timestep, s._dt_chain,
                                       s.leapfrog_steps, n, thin, k, x0, n_adapt,
                                       s.adaption_uprate, s.adaption_downrate,
                                       _modes()[s.mode], s.rng.seed,
                                       s.rng.offset,
                                       chain_offset=chain_offset(s))
        take_positions(s, n)             # taken once the launch is in
    else:
        _native.hmc_sample_n_gauss(q0, p0.contiguous(), u.contiguous(), q_out,
                                   samples, accepted, s.n_accepted, eb, ea,
                                   s._timestep, s._dt_chain, s.leapfrog_steps,
                                   n, thin, k, x0, n_adapt,
                                   s.adaption_uprate,
                                   s.adaption_downrate, _modes()[s.mode])
    s.last_e_before, s.last_e_after = eb, ea
    s._last_move_accepted = accepted[-1].view(torch.bool)
    s.accepted_history = accepted.view(torch.bool)
    s.counter += n
    return True, (q_out, samples)


def sample_rng(s, spec, q0, shape):
    """``hmc_rng`` hook of the isotropic Gaussian: with a generator whose draws are
    the lane streams of the fused kernels, sample() is one launch that draws for
    itself; None with any other generator."""
    if not fused_rng(s, spec):
        return None
    if _native.gauss_persist_covers(q0.shape[1]):
        return sample_n_fused_rng(s, 1)
    return sample_long_fused_rng(s, spec, q0, shape)


def fused_rng(s, name_or_spec, D=None, spec=False):
    """True if this sampler's draws are the lane streams of the fused Gaussian
    kernels (csrc/xoshiro.hpp): a device generator that allows it and a PDF of the
    built-in kind.  Depends on the PDF only, never on the number of chains -- so a
    shard of a run draws what the whole run draws for its chains.  Called with the
    PDF's spec, or with ``(variable name, D)`` to look it up."""
    if not getattr(s.rng, 'fused', False):
        return False
    if isinstance(name_or_spec, str):
        if spec is False:
            spec = s._fused_spec(name_or_spec, D)
    else:
        spec = name_or_spec
    return spec is not None and spec[0] == GAUSS


def draws_in_kernel(s, C, D):
    """Lane-stream draws: generated inside the sampling kernel (True) or
    written out first by the draw kernel and read back (False)?  Same draws
    either way; this only picks the faster launch.  Up to 1024 chains of
    D = 768 / 1024 the library spreads a chain over 4 waves when the draws
    come from HBM (hmc_gauss_split.hip); that beats the one-wave kernel with
    its own generator (512 chains: 4.0 vs 10.5 us per transition,
    scripts/probe_small_batch_rng.py)."""
    if getattr(s.rng, 'fused', False) == 'always':
        return True
    return D > 1024 or _native.gauss_waves_per_chain(C, D) < 4


def chain_offset(s):
    return int(getattr(s.rng, 'chain_offset', 0))


def take_positions(s, n):
    """Reserve the lane streams' positions of ``n`` transitions -- one per
    ``sample()`` call, whatever the call shape: ``sample_n(n)`` draws what n
    ``sample()`` calls draw.  Returns the first."""
    first = s.rng.offset
    s.rng.offset += int(n)
    return first


def sample_long_fused_rng(s, spec, q0, shape):
    """sample() for chains beyond the persistent kernel's reach with the
    draws generated in the kernels (csrc/hmc_gauss_big.hip)."""
    _, k, x0 = spec
    C = q0.shape[0]
    dev = q0.device
    adapt = (s.counter + 1) < s.timestep_adaption_limit
    if adapt and s._dt_chain is None:
        s._dt_chain = torch.full((C,), float(s._timestep), dtype=torch.float64,
                                    device=dev)
    if not isinstance(s.n_accepted, torch.Tensor):
        s.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)
    accepted = torch.empty(C, dtype=torch.uint8, device=dev)
    q_out = torch.empty_like(q0)
    eb = ea = None
    if s.record_energies:
        eb = torch.empty(C, dtype=torch.float64, device=dev)
        ea = torch.empty(C, dtype=torch.float64, device=dev)
    _native.hmc_sample_gauss_big_rng(q0, q_out, accepted, s.n_accepted, eb, ea,
                                     s._timestep, s._dt_chain, s.leapfrog_steps, k, x0,
                                     adapt, s.adaption_uprate, s.adaption_downrate,
                                     _modes()[s.mode], s.rng.seed,
                                     s.rng.offset,
                                     chain_offset=chain_offset(s))
    take_positions(s, 1)
    s.last_e_before, s.last_e_after = eb, ea
    s._last_move_accepted = accepted.view(torch.bool)
    s.counter += 1
    s.state = q_out.view(shape)
    return s.state


def sample_n_long(s, spec, n, thin, p0, u, record, out, q0, shape, nrec):
    """sample_n for chains beyond the persistent kernel's reach
    (csrc/hmc_gauss_big.hip): n transitions from one call, every recorded
    state written where it is kept; the draws are supplied, generated in the
    kernels (a generator with lane streams: exactly the draws of n sample()
    calls) or drawn a block of transitions at a time."""
    _, k, x0 = spec
    C, D = q0.shape
    dev = q0.device
    n_adapt = max(0, min(n, s.timestep_adaption_limit - 1 - s.counter))
    if n_adapt > 0 and s._dt_chain is None:
        s._dt_chain = torch.full((C,), float(s._timestep), dtype=torch.float64, device=dev)
    if not isinstance(s.n_accepted, torch.Tensor):
        s.n_accepted = torch.zeros(C, dtype=torch.int64, device=dev)
    samples = None
    if record and nrec > 0:
        samples = out.view(nrec, C, D) if out is not None else \
            torch.empty((nrec, C, D), dtype=torch.float64, device=dev)
    accepted = torch.empty((n, C), dtype=torch.uint8, device=dev)
    eb = ea = None
    if s.record_energies:
        eb = torch.empty((n, C), dtype=torch.float64, device=dev)
        ea = torch.empty((n, C), dtype=torch.float64, device=dev)
    q_out = torch.empty_like(q0)
    args = (s._timestep, s._dt_chain, s.leapfrog_steps)
    tail = (k, x0, n_adapt, s.adaption_uprate, s.adaption_downrate, _modes()[s.mode])
    if p0 is None and u is None and fused_rng(s, spec):
        _native.hmc_sample_n_gauss_big(q0, None, None, q_out, samples, accepted, s.n_accepted,
                                       eb, ea, *args, n, thin, *tail,
                                       rng=(s.rng.seed, s.rng.offset, chain_offset(s)))
        take_positions(s, n)                 # n sample() calls take n stream positions
    elif p0 is not None and u is not None:
        _native.hmc_sample_n_gauss_big(q0, p0.reshape(n, C, D).contiguous(),
                                       u.reshape(n, C).contiguous(), q_out, samples, accepted,
                                       s.n_accepted, eb, ea, *args, n, thin, *tail)
    else:
        # draws from the sampler's generator, a block of transitions at a time (<= 1 GiB
        # of momenta), in the order n sample() calls consume it: normal, uniform, ...
        block = max(1, min(n, (1 << 27) // max(1, C * D)))
        if samples is not None:
            block = max(thin, block // thin * thin)
        done, cur = 0, q0
        while done < n:
            m = min(block, n - done)
            dp = p0[done:done + m] if p0 is not None else \
                torch.empty((m, C, D), dtype=torch.float64, device=dev)
            du = u[done:done + m] if u is not None else \
                torch.empty((m, C), dtype=torch.float64, device=dev)
            for i in range(m):
                if p0 is None:
                    _fill(s.rng, 'normal', dp[i])
                if u is None:
                    _fill(s.rng, 'uniform', du[i])
            nxt = torch.empty_like(q0)
            r0, r1 = done // thin, (done + m) // thin
            _native.hmc_sample_n_gauss_big(
                cur, dp.contiguous(), du.contiguous(), nxt,
                samples[r0:r1] if samples is not None and r1 > r0 else None,
                accepted[done:done + m], s.n_accepted,
                eb[done:done + m] if eb is not None else None,
                ea[done:done + m] if ea is not None else None, *args, m, thin, k, x0,
                max(0, min(m, n_adapt - done)), s.adaption_uprate, s.adaption_downrate,
                _modes()[s.mode])
            cur = nxt
            done += m
        q_out = cur
    s.last_e_before, s.last_e_after = eb, ea
    s._last_move_accepted = accepted[-1].view(torch.bool)
    s.accepted_history = accepted.view(torch.bool)
    s.counter += n
    return q_out, samples


def sample_n_fused_rng(s, n):
    """sample() with in-kernel draws: one transition, the new state."""
    s.sample_n(n, record=False)
    if n == 1 and s.last_e_before is not None:
        s.last_e_before, s.last_e_after = s.last_e_before[0], s.last_e_after[0]
    return s.state


def sample(s, spec, q0, p0, u, accepted, adapt):
    """``hmc`` hook: one transition of every chain with the draws supplied, one launch
    (``binf_hmc_sample_gauss_f64``; chains beyond 8192 dimensions: the chunked kernels)."""
    _, k, x0 = spec
    C, D = q0.shape
    q_out = torch.empty_like(q0)
    eb = ea = None
    if s.record_energies:
        eb = torch.empty(C, dtype=torch.float64, device=q0.device)
        ea = torch.empty(C, dtype=torch.float64, device=q0.device)
    launch = _native.hmc_sample_gauss if _native.gauss_persist_covers(D) \
        else _native.hmc_sample_gauss_big       # chains of any length, chunked
    launch(q0, p0, u, q_out, accepted, s.n_accepted, eb, ea, s._timestep,
           s._dt_chain, s.leapfrog_steps, k, x0, adapt, s.adaption_uprate,
           s.adaption_downrate, _modes()[s.mode])
    s.last_e_before, s.last_e_after = eb, ea
    return q_out


native.register(GAUSS, replace=True, hmc=sample, hmc_rng=sample_rng, hmc_n=sample_n)
