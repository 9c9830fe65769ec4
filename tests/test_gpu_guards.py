"""Memory-safety checks without a GPU sanitizer (not available on this pool):
every buffer handed to a kernel is a window inside a larger allocation whose
surroundings are NaN (doubles) / 0xAB (bytes).  An out-of-bounds WRITE shows up
as a changed guard zone; an out-of-bounds READ that reaches the arithmetic
turns results into NaN or makes them differ from the same call on plain
tensors.  Shapes are ragged on purpose (partial waves, partial tiles, tails)."""
import numpy as np
import pytest
import torch

from binf_amd import _native

pytestmark = pytest.mark.gpu
PAD = 1024


class Guarded(object):
    def __init__(self, device):
        self.device = device
        self.zones = []

    def __call__(self, a, dtype=torch.float64):
        a = np.ascontiguousarray(a)
        n = a.size
        if dtype == torch.float64:
            buf = torch.full((n + 2 * PAD,), float('nan'), dtype=dtype, device=self.device)
        else:
            buf = torch.full((n + 2 * PAD,), 0xAB if dtype == torch.uint8 else -0x5555, dtype=dtype,
                             device=self.device)
        win = buf[PAD:PAD + n]
        win.copy_(torch.from_numpy(a.reshape(-1)).to(self.device).to(dtype))
        self.zones.append((buf, n, dtype))
        return win.view(a.shape)

    def check(self):
        torch.cuda.synchronize()
        for buf, n, dtype in self.zones:
            for z in (buf[:PAD], buf[PAD + n:]):
                if dtype == torch.float64:
                    assert bool(torch.isnan(z).all()), 'guard zone overwritten'
                else:
                    ref = 0xAB if dtype == torch.uint8 else -0x5555
                    assert bool((z == ref).all()), 'guard zone overwritten'


def plain(a, device, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device).to(dtype)


@pytest.mark.parametrize('D,C,n', [(1, 1, 1), (7, 3, 2), (33, 9, 3), (129, 5, 2), (258, 3, 2),
                                   (768, 5, 3), (1000, 3, 2), (1024, 1, 3), (1024, 1031, 2),
                                   (1024, 2049, 1), (1025, 3, 2), (3000, 2, 2), (8192, 1, 1),
                                   (7700, 2, 1)])
def test_gaussian_hmc_kernels_stay_inside_their_buffers(device, D, C, n):
    rs = np.random.RandomState(D + C)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        tq, tp, tu = t(q0), t(p0), t(u)
        qo, smp = t(np.zeros((C, D))), t(np.zeros((n, C, D)))
        acc = t(np.zeros((n, C), dtype=np.uint8), torch.uint8)
        nacc = t(np.zeros(C, dtype=np.int64), torch.int64)
        eb, ea = t(np.zeros((n, C))), t(np.zeros((n, C)))
        dtc = t(np.full(C, 0.2))
        if D <= 8192 and _native.pairwise_tree_height(D) <= 6:
            _native.hmc_sample_n_gauss(tq, tp, tu, qo, smp, acc, nacc, eb, ea, 0.2, dtc, 5, n, 1,
                                       2.5, 0.3, n, 1.05, 0.95, _native.MODE_EXACT)
        else:                                   # generic-tier kernels
            g = t(np.zeros((C, D)))
            _native.lib()
            qo.copy_(tq)
            for i in range(n):
                rc = _native.lib().binf_gauss_grad_f64(qo.data_ptr(), g.data_ptr(), 2.5, 0.3, C, D,
                                                       _native.stream_handle(device))
                assert rc == 0
                _native.leapfrog_kick(tp[i], g, 0.2, dtc, half=True)
                _native.leapfrog_kick_drift(qo, tp[i], g, 0.2, dtc)
                eb[i].copy_(_native.row_sum(qo, _native.ROW_SUMSQ_SHIFT, shift=0.3, scale=-1.25))
                ea[i].copy_(_native.row_sum(tp[i], _native.ROW_SUMSQ, scale=0.5))
                _native.accept_select(qo, tq, eb[i], ea[i], tu[i], qo, acc[i], nacc, dtc, True, 1.05, 0.95)
            smp.copy_(tp)
        if make is not None:
            make.check()
        outs.append([x.clone().cpu() for x in (qo, smp, acc, nacc, eb, ea, dtc)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
        assert not torch.isnan(a.double()).any()


@pytest.mark.parametrize('K,N,C', [(1, 1, 1), (4, 20, 70), (17, 100, 33), (33, 1000, 20), (33, 4099, 130),
                                   (34, 257, 65), (50, 64, 3), (64, 8200, 2), (5, 7700, 3),
                                   # N a multiple of 16: the whole-tile MFMA gradient kernel (buffer loads)
                                   (4, 32, 70), (33, 1024, 20), (17, 4096, 33), (33, 16384, 130), (64, 16, 1)])
def test_polynomial_kernels_stay_inside_their_buffers(device, K, N, C):
    rs = np.random.RandomState(K + N)
    xs, ys = np.linspace(-1, 1, N), rs.standard_normal(N)
    th, tau = 0.3 * rs.standard_normal((C, K)), rs.uniform(0.5, 2, size=C)
    A = np.vstack([xs ** i for i in range(K)])
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        tth, txs, tys, ttau, tA = t(th), t(xs), t(ys), t(tau), t(A)
        r = [_native.poly_forward(tth, txs), _native.poly_gauss_logp(tth, txs, tys, ttau),
             _native.poly_gauss_grad(tth, tA, tys, ttau)]
        if make is not None:
            make.check()
        outs.append([x.clone().cpu() for x in r])
    for a, b in zip(*outs):
        assert torch.equal(a, b) and not torch.isnan(a).any()


@pytest.mark.parametrize('K,N,C', [(1, 1, 1), (4, 20, 70), (9, 100, 65), (16, 128, 3)])
def test_fused_transition_transition_stays_inside_its_buffers(device, K, N, C):
    rs = np.random.RandomState(K * N)
    xs, ys = np.linspace(-1, 1, N), rs.standard_normal(N)
    q0, p0, u = 0.2 * rs.standard_normal((C, K)), rs.standard_normal((C, K)), rs.uniform(size=C)
    tau, lp = rs.uniform(0.5, 2, size=C), rs.standard_normal(C)
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        qo = t(np.zeros((C, K)))
        acc = t(np.zeros(C, dtype=np.uint8), torch.uint8)
        nacc = t(np.zeros(C, dtype=np.int64), torch.int64)
        eb, ea, dtc = t(np.zeros(C)), t(np.zeros(C)), t(np.full(C, 1e-3))
        _native.hmc_sample_poly(t(q0), t(p0), t(u), qo, acc, nacc, eb, ea, t(xs), t(ys), t(tau),
                                t(np.zeros(K)), t(np.full(K, 5.0)), True, t(lp), t(lp), 1e-3, dtc,
                                4, True, 1.05, 0.95)
        if make is not None:
            make.check()
        outs.append([x.clone().cpu() for x in (qo, acc, nacc, eb, ea, dtc)])
    for a, b in zip(*outs):
        assert torch.equal(a, b) and not torch.isnan(a.double()).any()


@pytest.mark.parametrize('n,C', [(2, 1), (3, 5), (17, 9), (100, 3), (256, 2), (257, 2), (513, 2), (1030, 1),
                                 (320, 3), (512, 2), (640, 2), (1000, 2), (1024, 2), (300, 700)])
def test_distance_kernels_stay_inside_their_buffers(device, n, C):
    rs = np.random.RandomState(n)
    x = rs.standard_normal((C, 3 * n)) * 2
    I, J = np.triu_indices(n, 1)
    ys = np.abs(rs.standard_normal(I.size)) + 0.5
    ym = np.zeros((n, n)); ym[I, J] = ys; ym[J, I] = ys
    tau, p = rs.uniform(0.5, 2, size=C), rs.standard_normal((C, 3 * n))
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        tx, tI, tJ = t(x), t(I.astype(np.int32), torch.int32), t(J.astype(np.int32), torch.int32)
        tys, tym, ttau = t(ys), t(ym), t(tau)
        r = [_native.pairdist_forward(tx, tI, tJ), _native.pairdist_gauss_logp(tx, tI, tJ, tys, ttau),
             _native.pairdist_gauss_grad(tx, tym, ttau)]
        if n <= 1024:
            q2, p2 = t(x), t(p)
            _native.pairdist_leapfrog(q2, p2, tym, ttau, (0.05, 0.1), True, 0.002, None, 3)
            r += [q2, p2]
        nbytes = _native.lib().binf_pairdist_packed_targets_bytes(n)
        if nbytes:
            # the packed targets (32..256 beads: one load per launch; 257..1024: the ring kernels
            # stream them for every force evaluation) as a guarded window of their own
            tpk = t(np.zeros(nbytes // 8))
            rc = _native.lib().binf_pairdist_pack_targets_f64(tym.data_ptr(), tpk.data_ptr(), n,
                                                              _native.stream_handle(device))
            assert rc == 0
            q3, p3 = t(x), t(p)
            _native.pairdist_leapfrog(q3, p3, tym, ttau, (0.05, 0.1), True, 0.002, None, 3, packed=tpk)
            r += [_native.pairdist_gauss_grad(tx, tym, ttau, packed=tpk), q3, p3, tpk]
        if make is not None:
            make.check()
        outs.append([z.clone().cpu() for z in r])
    for a, b in zip(*outs):
        assert torch.equal(a, b) and not torch.isnan(a).any()


@pytest.mark.parametrize('n', [1, 2, 3, 7, 8, 9, 63, 64, 65, 511, 513, 1001, 4097, 100003])
@pytest.mark.parametrize('kind', ['uniform', 'normal', 'normal_zig', 'gamma'])
def test_rng_kernels_stay_inside_their_buffers(device, kind, n):
    """The generators store pairs / blocks of draws; odd and tiny lengths must
    not spill, and the window start here is only 8-byte aligned relative to
    the 16-byte stores' natural alignment when PAD is odd-sized (it is not:
    a second window at an odd element offset covers that)."""
    for shift in (0, 1):
        g = Guarded(device)
        buf = g(np.zeros(n + shift))
        out = buf[shift:]                       # 8-byte aligned, possibly not 16-byte aligned
        _native.rng_fill(kind, out, 123, 4 * n, shape=10.0)
        g.check()
        assert bool(torch.isfinite(out).all())
        if shift:
            assert float(buf[0]) == 0.0         # the element before the window
        ref = torch.empty(n, dtype=torch.float64, device=device)
        _native.rng_fill(kind, ref, 123, 4 * n, shape=10.0)
        assert torch.equal(out, ref)


@pytest.mark.parametrize('D,C', [(8193, 3), (7689, 2), (20000, 3), (16384 + 5, 2), (24576, 2)])
def test_long_chain_kernels_stay_inside_their_buffers(device, D, C):
    """binf_hmc_sample_gauss_big_f64: full chunks, a ragged last chunk, a last
    chunk shorter than one accumulator row; guarded windows vs plain tensors."""
    rs = np.random.RandomState(D)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((C, D)), rs.uniform(size=C)
    u[0] = 0.999999
    p0[0] *= 4.0                                    # a rejected chain: restored from q0
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        tq, tp, tu = t(q0), t(p0), t(u)
        qo = t(np.zeros((C, D)))
        acc = t(np.zeros(C, dtype=np.uint8), torch.uint8)
        nacc = t(np.zeros(C, dtype=np.int64), torch.int64)
        eb, ea = t(np.zeros(C)), t(np.zeros(C))
        dtc = t(np.full(C, 0.05))
        _native.hmc_sample_gauss_big(tq, tp, tu, qo, acc, nacc, eb, ea, 0.05, dtc, 3, 2.5, 0.3,
                                     True, 1.05, 0.95, _native.MODE_EXACT)
        if make is not None:
            make.check()
        outs.append([x.cpu().numpy().copy() for x in (qo, acc, nacc, eb, ea, dtc)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert not np.isnan(outs[0][0]).any()
    assert outs[0][1][0] == 0 and np.array_equal(outs[0][0][0], q0[0])


@pytest.mark.parametrize('D,C,n', [(1, 3, 2), (7, 5, 2), (33, 9, 3), (258, 3, 2), (1000, 3, 2),
                                   (1024, 70, 2), (2048, 3, 2), (3000, 2, 1), (8192, 2, 1)])
def test_in_kernel_generator_stays_inside_its_buffers(device, D, C, n):
    """binf_hmc_gauss_rng_draws_f64 / binf_hmc_sample_n_gauss_rng_f64 on guarded
    windows: guard zones untouched, every element of the dump written, results
    equal to the same calls on plain tensors."""
    lib = _native.lib()
    st = _native.stream_handle(device)
    rs = np.random.RandomState(D)
    q0 = rs.standard_normal((C, D))
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        pd, ud = t(np.full((n, C, D), 77.0)), t(np.full((n, C), 77.0))
        rc = lib.binf_hmc_gauss_rng_draws_f64(pd.data_ptr(), ud.data_ptr(), C, D, n, 5, 2, 3, st)
        assert rc == 0
        tq, qo, smp = t(q0), t(np.zeros((C, D))), t(np.zeros((n, C, D)))
        acc = t(np.zeros((n, C), dtype=np.uint8), torch.uint8)
        nacc = t(np.zeros(C, dtype=np.int64), torch.int64)
        eb, ea = t(np.zeros((n, C))), t(np.zeros((n, C)))
        dtc = t(np.full(C, 0.1))
        rc = lib.binf_hmc_sample_n_gauss_rng_f64(
            tq.data_ptr(), qo.data_ptr(), smp.data_ptr(), acc.data_ptr(), nacc.data_ptr(),
            eb.data_ptr(), ea.data_ptr(), 0.1, dtc.data_ptr(), C, D, 4, n, 1, 2.5, 0.3, n, 1.05,
            0.95, _native.MODE_EXACT, 5, 2, 3, st)
        assert rc == 0
        if make is not None:
            make.check()
        outs.append([x.cpu().numpy().copy() for x in (pd, ud, qo, smp, acc, nacc, eb, ea, dtc)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    pd, ud = outs[0][0], outs[0][1]
    assert not (pd == 77.0).any() and not (ud == 77.0).any()      # every draw written
    assert np.isfinite(pd).all() and (0.0 <= ud).all() and (ud < 1.0).all()
    assert np.isfinite(outs[0][3]).all()


# ---------------------------------------------------------------------------
# round 3 entry points
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('K,N,C,n,thin,move', [(4, 20, 1, 3, 1, 'hmc'), (7, 37, 9, 5, 2, 'hmc'),
                                               (16, 128, 33, 2, 1, 'rwmc'), (3, 300, 5, 4, 3, 'hmc'),
                                               (1, 1000, 3, 2, 1, 'rwmc'), (15, 920, 2, 2, 2, 'hmc')])
def test_multi_sweep_gibbs_launch_stays_inside_its_buffers(device, K, N, C, n, thin, move):
    rs = np.random.RandomState(K * N + C)
    xs, ys = np.linspace(-1, 1, N), rs.standard_normal(N)
    th0, tau0 = rs.standard_normal((C, K)), 1.0 + rs.uniform(size=C)
    p0 = rs.standard_normal((n, C, K)) * (1.0 if move == 'hmc' else 0.05)
    u, g = rs.uniform(size=(n, C)), 8.0 + rs.uniform(size=(n, C))
    nrec = n // thin
    outs = []
    for supplied in (True, False):
        for make in (Guarded(device), None):
            t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
            o = dict(th=t(np.zeros((C, K))), tau=t(np.zeros(C)), rc=t(np.zeros((nrec, C, K))),
                     rt=t(np.zeros((nrec, C))), acc=t(np.zeros((n, C), dtype=np.uint8), torch.uint8),
                     nacc=t(np.zeros(C, dtype=np.int64), torch.int64), eb=t(np.zeros((n, C))),
                     ea=t(np.zeros((n, C))), dtc=t(np.full(C, 1e-3)))
            _native.gibbs_poly_sample_n(
                t(th0), t(tau0), o['th'], o['tau'], t(xs), t(ys), n, thin,
                move=_native.MOVE_HMC if move == 'hmc' else _native.MOVE_RWMC, nsteps=3,
                dt_chain=o['dtc'], n_adapt=1 if move == 'hmc' else 0, stepsize=0.05,
                prior_means=t(np.zeros(K)), prior_vars=t(np.full(K, 5.0)), prior_first=True,
                gp_where=2, gp_shape=1.0, gp_rate=1.0, gamma_shape=0.5 * N + 1.0, gamma_rate=1.0,
                rec_coefficients=o['rc'] if nrec else None, rec_precision=o['rt'] if nrec else None,
                accepted=o['acc'], n_accepted=o['nacc'], e_before=o['eb'], e_after=o['ea'],
                p0=t(p0) if supplied else None, u=t(u) if supplied else None,
                g=t(g) if supplied else None, streams=((3, 0, 130), (3, 1, 130), (3, 2, 130)),
                chain_offset=5)
            if make is not None:
                make.check()
            outs.append([o[k].clone() for k in ('th', 'tau', 'rc', 'rt', 'acc', 'nacc', 'dtc')])
        for a, b in zip(outs[-2], outs[-1]):
            assert torch.equal(a, b) and (a.dtype != torch.float64 or bool(torch.isfinite(a).all()))


@pytest.mark.parametrize('K,N,C,batched', [(1, 1, 1, False), (17, 65, 17, False), (33, 1000, 20, True),
                                           (65, 129, 3, False), (5, 63, 9, True), (64, 64, 16, False),
                                           # from 8192 chains x 1024 points up: 32 chains per workgroup
                                           (33, 1025, 8197, False), (16, 1024, 8193, False),
                                           (49, 1100, 8200, False), (64, 1030, 8223, False)])
def test_contraction_and_term_sum_stay_inside_their_buffers(device, K, N, C, batched):
    rs = np.random.RandomState(K + N + C)
    J = rs.standard_normal((C, K, N) if batched else (K, N))
    r = rs.standard_normal((C, N))
    gz = Guarded(device)
    got = _native.jacobian_contract(gz(J), gz(r))
    gz.check()
    assert torch.equal(got, _native.jacobian_contract(plain(J, device), plain(r, device)))
    assert bool(torch.isfinite(got).all())
    terms = [rs.standard_normal(C) for _ in range(3)]
    gz = Guarded(device)
    s1 = _native.sum_terms([gz(terms[0]), 0.5, gz(terms[1]), gz(terms[2])])
    gz.check()
    assert torch.equal(s1, _native.sum_terms([plain(terms[0], device), 0.5, plain(terms[1], device),
                                              plain(terms[2], device)]))


@pytest.mark.parametrize('K,N,C,L', [(4, 20, 5, 3), (33, 1000, 20, 2), (33, 16384, 130, 1), (17, 50, 2100, 2)])
def test_fused_transition_leapfrog_stays_inside_its_buffers(device, K, N, C, L):
    rs = np.random.RandomState(K + C)
    xs = np.linspace(-1, 1, N)
    A = np.vstack([xs ** i for i in range(K)])
    ys, q0, p0 = rs.standard_normal(N), rs.standard_normal((C, K)), rs.standard_normal((C, K))
    taus, dts = rs.uniform(1, 3, size=C), np.full(C, 1e-4)
    res = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        q, p = t(q0), t(p0)
        _native.poly_leapfrog(q, p, t(A), t(ys), t(taus), 0.0, t(dts), L)
        if make is not None:
            make.check()
        res.append((q.clone(), p.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert bool(torch.isfinite(res[0][0]).all())


@pytest.mark.parametrize('D,C,n,thin', [(8193, 3, 3, 1), (20000, 2, 4, 3), (16384 + 5, 2, 2, 2)])
def test_long_chain_loop_from_one_call_stays_inside_its_buffers(device, D, C, n, thin):
    rs = np.random.RandomState(D)
    q0, p0, u = rs.standard_normal((C, D)), rs.standard_normal((n, C, D)), rs.uniform(size=(n, C))
    nrec = n // thin
    res = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        qo, smp = t(np.zeros((C, D))), t(np.zeros((nrec, C, D)))
        acc = t(np.zeros((n, C), dtype=np.uint8), torch.uint8)
        nacc = t(np.zeros(C, dtype=np.int64), torch.int64)
        eb, ea = t(np.zeros((n, C))), t(np.zeros((n, C)))
        _native.hmc_sample_n_gauss_big(t(q0), t(p0), t(u), qo, smp, acc, nacc, eb, ea, 0.01, None, 3,
                                       n, thin, 1.0, 0.0, 0, 1.05, 0.95)
        if make is not None:
            make.check()
        res.append((qo.clone(), smp.clone(), acc.clone(), ea.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(res[0][0]).all())


@pytest.mark.parametrize('S,nx,ny', [(1, 1, 1), (5, 3, 7), (16, 16, 8), (17, 17, 9), (100, 33, 150), (1000, 5, 31),
                                     (5001, 3, 7), (700, 1, 1)])
def test_predictive_density_stays_inside_its_buffers(device, S, nx, ny):
    rs = np.random.RandomState(S + nx + ny)
    mock, tau, ys = rs.standard_normal((S, nx)), rs.gamma(4.0, 0.5, size=S), rs.standard_normal((nx, ny)) * 2
    h = 0.5 * np.log(2 * np.pi)
    outs = []
    for make in (Guarded(device), None):
        t = make if make is not None else (lambda a, dtype=torch.float64: plain(a, device, dtype))
        r = _native.predictive_density(t(mock), t(tau), t(ys), h)
        if make is not None:
            make.check()
        outs.append(r.clone().cpu())
    assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0]).any() and bool((outs[0] > 0).all())
