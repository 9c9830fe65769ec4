"""
ctypes binding of libbinf_hip.so (the C ABI declared in include/binf_hip.h).

The HIP library IS the product path: there is no CPU or PyTorch fallback.  If
the shared object is missing or a call fails, this module raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BINF_LIB_OVERRIDE') or \
    os.path.join(_HERE, 'csrc', 'libbinf_hip.so')   # override: A/B of builds

MODE_EXACT = 0
MODE_FMA = 1
MODE_LANE_PER_CHAIN = 16     # flag for hmc_sample_poly: one lane per chain (N <= 128)
MOVE_HMC = 0
MOVE_RWMC = 1

E_ARG = -1
E_UNSUPPORTED = -2
E_ALIAS = -3


class NativeLibraryError(RuntimeError):
    """libbinf_hip.so is missing, stale or failed."""


_lib = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_f64 = ctypes.c_double
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must list every symbol of include/binf_hip.h
SIGNATURES = {
    'binf_abi_version': (_i32, []),
    'binf_last_error': (_i32, [ctypes.c_char_p, ctypes.c_size_t]),
    'binf_hmc_sample_gauss_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _f64, _vp, _i64, _i64, _i32, _f64,
                                         _f64, _i32, _f64, _f64, _i32, _vp]),
    'binf_hmc_sample_n_gauss_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _f64, _vp, _i64, _i64,
                                           _i32, _i32, _i32, _f64, _f64, _i32,
                                           _f64, _f64, _i32, _vp]),
    'binf_hmc_gauss_waves_per_chain': (_i32, [_i64, _i64]),
    'binf_hmc_sample_gauss_big_workspace_bytes': (_i64, [_i64, _i64]),
    'binf_hmc_sample_gauss_big_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _f64, _vp, _i64, _i64, _i32, _f64,
                                             _f64, _i32, _f64, _f64, _i32, _vp,
                                             _i64, _vp]),
    'binf_hmc_sample_gauss_big_rng_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp,
                                                 _i64, _i64, _i32, _f64, _f64, _i32,
                                                 _f64, _f64, _i32, ctypes.c_uint64,
                                                 ctypes.c_uint64, _i64, _vp, _i64, _vp]),
    'binf_hmc_sample_n_gauss_big_workspace_bytes': (_i64, [_i64, _i64]),
    'binf_hmc_sample_n_gauss_big_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _f64, _vp, _i64, _i64, _i32, _i32, _i32, _f64,
                                               _f64, _i32, _f64, _f64, _i32, _vp, _i64, _vp]),
    'binf_hmc_sample_n_gauss_big_rng_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp,
                                                   _i64, _i64, _i32, _i32, _i32, _f64, _f64,
                                                   _i32, _f64, _f64, _i32, _u64, _u64, _i64,
                                                   _vp, _i64, _vp]),
    'binf_hmc_gauss_big_rng_draws_f64': (_i32, [_vp, _vp, _i64, _i64, ctypes.c_uint64,
                                                ctypes.c_uint64, _i64, _vp]),
    'binf_hmc_sample_n_gauss_rng_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _f64, _vp, _i64, _i64, _i32, _i32,
                                               _i32, _f64, _f64, _i32, _f64, _f64,
                                               _i32, ctypes.c_uint64,
                                               ctypes.c_uint64, _i64, _vp]),
    'binf_hmc_gauss_rng_draws_f64': (_i32, [_vp, _vp, _i64, _i64, _i32,
                                            ctypes.c_uint64, ctypes.c_uint64, _i64, _vp]),
    'binf_hmc_sample_poly_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _f64, _vp, _vp, _vp, _i32,
                                        _vp, _vp, _f64, _vp, _i64, _i64, _i64,
                                        _i32, _i32, _f64, _f64, _i32, _vp]),
    'binf_clipped_exp_f64': (_i32, [_vp, _vp, _i64, _vp]),
    'binf_row_sum_f64': (_i32, [_vp, _vp, _i64, _i64, _i32, _f64, _f64, _vp]),
    'binf_hmc_energy_f64': (_i32, [_vp, _vp, _vp, _i64, _i64, _vp]),
    'binf_gamma_logp_f64': (_i32, [_vp, _f64, _f64, _vp, _i64, _vp]),
    'binf_leapfrog_kick_f64': (_i32, [_vp, _vp, _f64, _vp, _i32, _i64, _i64,
                                      _i32, _vp]),
    'binf_leapfrog_drift_f64': (_i32, [_vp, _vp, _f64, _vp, _i64, _i64, _i32,
                                       _vp]),
    'binf_leapfrog_kick_drift_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _i64, _i64,
                                            _i32, _vp]),
    'binf_gauss_grad_f64': (_i32, [_vp, _vp, _f64, _f64, _i64, _i64, _vp]),
    'binf_accept_select_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _i32, _f64, _f64, _i64, _i64, _vp]),
    'binf_row_sumsq_diff_f64': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f64,
                                       _vp]),
    'binf_poly_forward_f64': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'binf_predictive_density_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'binf_predictive_density_f64': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f64, _vp, _i64, _vp]),
    'binf_gauss_err_grad_f64': (_i32, [_vp, _vp, _f64, _vp, _vp, _i64, _i64,
                                       _vp]),
    'binf_gauss_err_logp_f64': (_i32, [_vp, _vp, _f64, _vp, _vp, _i64, _i64,
                                       _vp]),
    'binf_poly_gauss_logp_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _vp, _i64,
                                        _i64, _i64, _vp]),
    'binf_poly_gauss_logp_memo_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp,
                                             _i64, _i64, _i64, _vp]),
    'binf_poly_gauss_grad_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'binf_poly_gauss_grad_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _vp, _vp,
                                        _i64, _i64, _i64, _i64, _vp]),
    'binf_gamma_precision_update_f64': (_i32, [_vp, _vp, _f64, _vp, _i64, _vp]),
    'binf_poly_leapfrog_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'binf_poly_leapfrog_f64': (_i32, [_vp, _vp, _vp, _vp, _f64, _vp, _vp, _i64, _i64, _i64, _i64,
                                      _f64, _vp, _i32, _i32, _vp]),
    'binf_pairdist_chi2_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'binf_pairdist_gauss_logp_f64': (_i32, [_vp, _vp, _vp, _vp, _f64, _vp, _vp, _i64,
                                            _i64, _i64, _vp, _i64, _vp]),
    'binf_pairdist_gauss_logp_memo_f64': (_i32, [_vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp,
                                                 _i64, _i64, _i64, _vp, _i64, _vp]),
    'binf_pairdist_forward_f64': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                         _vp]),
    'binf_pairdist_gauss_grad_f64': (_i32, [_vp, _vp, _f64, _vp, _vp, _i64,
                                            _i64, _vp]),
    'binf_pairdist_leapfrog_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _i32, _f64,
                                          _f64, _i32, _f64, _vp, _i32, _i64,
                                          _i64, _i32, _vp]),
    'binf_pairdist_hmc_energy_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _f64, _i32,
                                            ctypes.POINTER(_i32), _vp, _f64, _vp, _f64,
                                            _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    'binf_pairdist_packed_targets_bytes': (_i64, [_i64]),
    'binf_pairdist_pack_targets_f64': (_i32, [_vp, _vp, _i64, _vp]),
    'binf_pairdist_tiles_workspace_bytes': (_i64, [_i64, _i64]),
    'binf_pairdist_gauss_grad_packed_f64': (_i32, [_vp, _vp, _vp, _f64, _vp, _vp, _i64,
                                                   _i64, _vp, _i64, _vp]),
    'binf_pairdist_leapfrog_packed_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _f64, _vp, _i32, _f64,
                                                 _f64, _i32, _f64, _vp, _i32, _i64,
                                                 _i64, _i32, _vp, _i64, _vp]),
    'binf_rng_uniform_f64': (_i32, [_vp, _i64, ctypes.c_uint64, ctypes.c_uint64,
                                    _i64, _vp]),
    'binf_rng_normal_f64': (_i32, [_vp, _i64, ctypes.c_uint64, ctypes.c_uint64,
                                   _i64, _vp]),
    'binf_rng_normal_zig_uniform_f64': (_i32, [_vp, _i64, _vp, _i64, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_uint64, _i64, _i64, _vp]),
    'binf_rng_normal_zig_f64': (_i32, [_vp, _i64, ctypes.c_uint64,
                                       ctypes.c_uint64, _i64, _vp]),
    'binf_rng_gamma_f64': (_i32, [_vp, _i64, _f64, ctypes.c_uint64,
                                  ctypes.c_uint64, _i64, _vp]),
    'binf_rwmc_propose_f64': (_i32, [_vp, _vp, _vp, _f64, _i64, _i64, ctypes.c_uint64,
                                     ctypes.c_uint64, _i64, _vp]),
    'binf_rwmc_accept_f64': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                    ctypes.c_uint64, ctypes.c_uint64, _i64, _vp]),
    'binf_gibbs_poly_sample_n_f64': (_i32, [_vp, _vp]),
    'binf_jacobian_contract_f64': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    'binf_sum_terms_f64': (_i32, [_vp, _vp, _i32, _vp, _i64, _vp]),
    'binf_sum_terms_bcast_f64': (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    'binf_rng_philox4x32_10': (_i32, [ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_uint32)]),
    'binf_pairwise_tree_height': (_i32, [_i64]),
    'binf_pairwise_leaf': (_i32, [_i64, _i32, _i32,
                                  ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                  ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
}

ABI_VERSION = 6        # keep in step with BINF_ABI_VERSION (include/binf_hip.h)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                '%s not found: build it with `python -c "import '
                '__graft_entry__ as g; g.build()"` (hipcc --offload-arch='
                'gfx950).  binf_amd has no CPU fallback.' % LIB_PATH)
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            try:
                f = getattr(L, name)
            except AttributeError:
                raise NativeLibraryError('%s does not export %s (stale '
                                         'build?)' % (LIB_PATH, name))
            f.restype = res
            f.argtypes = args
        v = L.binf_abi_version()
        if v != ABI_VERSION:
            raise NativeLibraryError('ABI version %d != expected %d'
                                     % (v, ABI_VERSION))
        _lib = L
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().binf_last_error(buf, 512)
    return buf.value.decode('utf-8', 'replace')


def check(rc, what):
    """Map a C-ABI return code onto the exception classes the reference
    raises for the same mistakes (ValueError / NotImplementedError)."""
    if rc == 0:
        return
    msg = '%s: %s' % (what, last_error())
    if rc in (E_ARG, E_ALIAS):
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise NativeLibraryError(msg + ' [hipError_t %d]' % rc)


def dptr(t, dtype=torch.float64, numel=None, name='tensor'):
    """Device pointer of a contiguous ROCm tensor, with the shape checks the
    kernels rely on (a wrong size here would be an out-of-bounds access)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise ValueError('%s must live in GPU memory (got %s)'
                         % (name, t.device))
    if t.dtype != dtype:
        raise ValueError('%s must be %s (got %s)' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    if numel is not None and t.numel() != numel:
        raise ValueError('%s has %d elements, expected %d'
                         % (name, t.numel(), numel))
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_handle(device=None):
    """hipStream_t of torch's current stream on ``device`` (as an integer)."""
    if _raw_stream is not None:
        # the raw handle without building a torch.cuda.Stream object (~4 us less per launch)
        idx = device.index if isinstance(device, torch.device) and device.index is not None \
            else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


_n_devices = None


def _launcher(fn):
    """Every wrapper that enqueues a kernel: all tensor arguments must live on
    ONE device, and the launch runs with that device current (the stream handle
    passed to the C ABI belongs to it)."""
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kw):
        # a process that sees ONE device cannot mix devices: nothing to check
        # (the check costs ~2 us per launch, a fifth of a small launch's host time)
        global _n_devices
        if _n_devices is None:
            _n_devices = torch.cuda.device_count()
        if _n_devices <= 1:
            return fn(*args, **kw)
        dev = None
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise ValueError('%s: tensors on different devices (%s and %s)'
                                     % (fn.__name__, dev, a.device))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return guarded


def pairwise_tree_height(n):
    return lib().binf_pairwise_tree_height(int(n))


def pairwise_leaf(n, H, path):
    off, ln = _i64(), _i64()
    depth, canon = _i32(), _i32()
    rc = lib().binf_pairwise_leaf(int(n), int(H), int(path),
                                  ctypes.byref(off), ctypes.byref(ln),
                                  ctypes.byref(depth), ctypes.byref(canon))
    check(rc, 'binf_pairwise_leaf')
    return off.value, ln.value, depth.value, canon.value


@_launcher
def hmc_sample_gauss(q0, p0, u, q_out, accepted, n_accepted, e_before, e_after, timestep,
                     dt_chain, nsteps, k, x0, adapt, uprate, downrate,
                     mode=MODE_EXACT):
    """binf_hmc_sample_gauss_f64 on torch's current stream."""
    if q0.dim() != 2:
        raise ValueError('q0 must be [n_chains, n_dims]')
    C, D = q0.shape
    n = C * D
    rc = lib().binf_hmc_sample_gauss_f64(
        dptr(q0, numel=n, name='q0'), dptr(p0, numel=n, name='p0'),
        dptr(u, numel=C, name='u'), dptr(q_out, numel=n, name='q_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=C, name='e_before'),
        dptr(e_after, numel=C, name='e_after'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'),
        C, D, int(nsteps), float(k), float(x0), int(bool(adapt)),
        float(uprate), float(downrate), int(mode), stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_gauss_f64')


ROW_SUM, ROW_SUMSQ, ROW_SUMSQ_SHIFT = 0, 1, 2


def _cd(x):
    if x.dim() != 2:
        raise ValueError('expected a [n_chains, n_dims] tensor, got shape %s'
                         % (tuple(x.shape),))
    return x.shape


@_launcher
def row_sum(x, op=ROW_SUM, shift=0.0, scale=1.0, out=None):
    """scale * np.sum(f(x[c, :])) per chain, numpy pairwise order."""
    C, D = _cd(x)
    if out is None:
        out = torch.empty(C, dtype=torch.float64, device=x.device)
    rc = lib().binf_row_sum_f64(dptr(x, numel=C * D, name='x'),
                                dptr(out, numel=C, name='out'), C, D, int(op),
                                float(shift), float(scale),
                                stream_handle(x.device))
    check(rc, 'binf_row_sum_f64')
    return out


@_launcher
def hmc_energy(p, log_prob):
    """``-log_prob + 0.5 * np.sum(p**2)`` per chain (hmc.py:143,148,150), one
    launch; ``log_prob`` is a contiguous ``[C]`` tensor."""
    C, D = _cd(p)
    out = torch.empty(C, dtype=torch.float64, device=p.device)
    rc = lib().binf_hmc_energy_f64(dptr(p, numel=C * D, name='p'),
                                   dptr(log_prob, numel=C, name='log_prob'), dptr(out), C, D,
                                   stream_handle(p.device))
    check(rc, 'binf_hmc_energy_f64')
    return out


@_launcher
def leapfrog_kick(p, grad, timestep, dt_chain=None, half=False,
                  mode=MODE_EXACT):
    C, D = _cd(p)
    rc = lib().binf_leapfrog_kick_f64(
        dptr(p, numel=C * D, name='p'), dptr(grad, numel=C * D, name='grad'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'),
        int(bool(half)), C, D, int(mode), stream_handle(p.device))
    check(rc, 'binf_leapfrog_kick_f64')


@_launcher
def leapfrog_kick_drift(q, p, grad, timestep, dt_chain=None, mode=MODE_EXACT):
    """p -= dt * grad; q += p * dt  (kick then drift, one pass)."""
    C, D = _cd(q)
    rc = lib().binf_leapfrog_kick_drift_f64(
        dptr(q, numel=C * D, name='q'), dptr(p, numel=C * D, name='p'),
        dptr(grad, numel=C * D, name='grad'), float(timestep),
        dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(mode),
        stream_handle(q.device))
    check(rc, 'binf_leapfrog_kick_drift_f64')


@_launcher
def leapfrog_drift(q, p, timestep, dt_chain=None, mode=MODE_EXACT):
    C, D = _cd(q)
    rc = lib().binf_leapfrog_drift_f64(
        dptr(q, numel=C * D, name='q'), dptr(p, numel=C * D, name='p'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), C, D,
        int(mode), stream_handle(q.device))
    check(rc, 'binf_leapfrog_drift_f64')


@_launcher
def gauss_grad(x, k, x0, out=None):
    C, D = _cd(x)
    if out is None:
        out = torch.empty_like(x)
    rc = lib().binf_gauss_grad_f64(dptr(x, numel=C * D, name='x'),
                                   dptr(out, numel=C * D, name='out'),
                                   float(k), float(x0), C, D,
                                   stream_handle(x.device))
    check(rc, 'binf_gauss_grad_f64')
    return out


@_launcher
def accept_select(q_prop, q_old, e_before, e_after, u, q_out, accepted,
                  n_accepted=None, dt_chain=None, adapt=False, uprate=1.05, downrate=0.95):
    C, D = _cd(q_prop)
    n = C * D
    rc = lib().binf_accept_select_f64(
        dptr(q_prop, numel=n, name='q_prop'), dptr(q_old, numel=n, name='q_old'),
        dptr(e_before, numel=C, name='e_before'),
        dptr(e_after, numel=C, name='e_after'), dptr(u, numel=C, name='u'),
        dptr(q_out, numel=n, name='q_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(dt_chain, numel=C, name='dt_chain'), int(bool(adapt)),
        float(uprate), float(downrate), C, D, stream_handle(q_prop.device))
    check(rc, 'binf_accept_select_f64')


@_launcher
def clipped_exp(x):
    """exp(clip(x, -308, 709)) elementwise (csb.numeric.exp)."""
    require_device(x, 'x')
    xc = x.contiguous()
    out = torch.empty_like(xc)
    rc = lib().binf_clipped_exp_f64(dptr(xc), dptr(out), xc.numel(),
                                    stream_handle(x.device))
    check(rc, 'binf_clipped_exp_f64')
    return out


def require_device(x, what):
    """The package computes on the GPU only: ``x`` must be a ROCm tensor.
    Raises TypeError otherwise (there is no numpy / CPU evaluation path)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise TypeError('%s must be a ROCm (cuda) fp64 tensor, got %s; binf_amd '
                        'evaluates on the GPU only (no CPU path)'
                        % (what, type(x).__name__ if not isinstance(x, torch.Tensor)
                           else 'a %s tensor' % x.device.type))
    return x


def _precision_args(precision, C, device):
    """(host scalar, per-chain tensor or None) from a float or a [C] tensor."""
    if isinstance(precision, torch.Tensor):
        if precision.dim() == 0:
            return float(precision), None
        p = precision.reshape(-1)
        if p.numel() != C:
            raise ValueError('precision has %d entries for %d chains'
                             % (p.numel(), C))
        return 0.0, p.contiguous()
    return float(precision), None


@_launcher
def row_sumsq_diff(x, y, scale=1.0, weights=None):
    C, D = _cd(x)
    out = torch.empty(C, dtype=torch.float64, device=x.device)
    rc = lib().binf_row_sumsq_diff_f64(dptr(x, numel=C * D, name='x'),
                                       dptr(y, numel=D, name='y'),
                                       dptr(weights, numel=D, name='weights'),
                                       dptr(out), C, D, float(scale),
                                       stream_handle(x.device))
    check(rc, 'binf_row_sumsq_diff_f64')
    return out


@_launcher
def poly_forward(coeffs, xs):
    C, K = _cd(coeffs)
    N = xs.numel()
    out = torch.empty((C, N), dtype=torch.float64, device=coeffs.device)
    rc = lib().binf_poly_forward_f64(dptr(coeffs, numel=C * K, name='coeffs'),
                                     dptr(xs, numel=N, name='xs'), dptr(out),
                                     C, K, N, stream_handle(coeffs.device))
    check(rc, 'binf_poly_forward_f64')
    return out


@_launcher
def predictive_density(mock, precision, ys, half_log_2pi):
    """binf_predictive_density_f64: ``mock`` [S x nx], ``precision`` [S], ``ys``
    [nx x ny] -> densities [nx x ny] (``binf/example/misc.py:3-16`` for a grid)."""
    S, nx = _cd(mock)
    if ys.dim() != 2 or ys.shape[0] != nx:
        raise ValueError('predictive_density: ys must be [nx x ny] with nx = %d' % nx)
    ny = ys.shape[1]
    out = torch.empty((nx, ny), dtype=torch.float64, device=mock.device)
    need = lib().binf_predictive_density_workspace_bytes(S, nx, ny)
    ws = torch.empty(need // 8, dtype=torch.float64, device=mock.device) if need > 0 else None
    rc = lib().binf_predictive_density_f64(dptr(mock, numel=S * nx, name='mock'),
                                           dptr(precision, numel=S, name='precision'),
                                           dptr(ys, numel=nx * ny, name='ys'), dptr(out),
                                           S, nx, ny, float(half_log_2pi),
                                           None if ws is None else ws.data_ptr(), need,
                                           stream_handle(mock.device))
    check(rc, 'binf_predictive_density_f64')
    return out


@_launcher
def gauss_err_grad(mock, ys, precision):
    C, N = _cd(mock)
    tau, tau_chain = _precision_args(precision, C, mock.device)
    out = torch.empty_like(mock)
    rc = lib().binf_gauss_err_grad_f64(dptr(mock, numel=C * N, name='mock'),
                                       dptr(ys, numel=N, name='ys'), tau,
                                       dptr(tau_chain, numel=C, name='precision'),
                                       dptr(out), C, N,
                                       stream_handle(mock.device))
    check(rc, 'binf_gauss_err_grad_f64')
    return out


@_launcher
def gauss_err_logp(mock, ys, precision):
    C, N = _cd(mock)
    tau, tau_chain = _precision_args(precision, C, mock.device)
    out = torch.empty(C, dtype=torch.float64, device=mock.device)
    rc = lib().binf_gauss_err_logp_f64(dptr(mock, numel=C * N, name='mock'),
                                       dptr(ys, numel=N, name='ys'), tau,
                                       dptr(tau_chain, numel=C, name='precision'),
                                       dptr(out), C, N,
                                       stream_handle(mock.device))
    check(rc, 'binf_gauss_err_logp_f64')
    return out


@_launcher
def poly_gauss_logp(coeffs, xs, ys, precision):
    C, K = _cd(coeffs)
    N = xs.numel()
    tau, tau_chain = _precision_args(precision, C, coeffs.device)
    out = torch.empty(C, dtype=torch.float64, device=coeffs.device)
    rc = lib().binf_poly_gauss_logp_f64(
        dptr(coeffs, numel=C * K, name='coeffs'), dptr(xs, numel=N, name='xs'),
        dptr(ys, numel=N, name='ys'), tau,
        dptr(tau_chain, numel=C, name='precision'), dptr(out), C, K, N,
        stream_handle(coeffs.device))
    check(rc, 'binf_poly_gauss_logp_f64')
    return out


def new_chi2_memo(C, K, device):
    """Buffers of the two-entry per-chain chi^2 memo (include/binf_hip.h,
    binf_poly_gauss_logp_memo_f64): ``(memo_args [2 x C x K] NaN, memo_chi2 [2 x C] NaN,
    memo_state [2 x C] uint8 zero)``; ``memo_state[0]`` = chains the last call reused."""
    nan = float('nan')
    return (torch.full((2, C, K), nan, dtype=torch.float64, device=device),
            torch.full((2, C), nan, dtype=torch.float64, device=device),
            torch.zeros((2, C), dtype=torch.uint8, device=device))


@_launcher
def poly_gauss_logp_memo(coeffs, xs, ys, precision, memo):
    """binf_poly_gauss_logp_memo_f64; ``memo = new_chi2_memo(C, K, device)``."""
    C, K = _cd(coeffs)
    N = xs.numel()
    tau, tau_chain = _precision_args(precision, C, coeffs.device)
    mc, ms, sk = memo
    out = torch.empty(C, dtype=torch.float64, device=coeffs.device)
    rc = lib().binf_poly_gauss_logp_memo_f64(
        dptr(coeffs, numel=C * K, name='coeffs'), dptr(xs, numel=N, name='xs'),
        dptr(ys, numel=N, name='ys'), tau, dptr(tau_chain, numel=C, name='precision'), dptr(out),
        dptr(mc, numel=2 * C * K, name='memo_coeffs'), dptr(ms, numel=2 * C, name='memo_chi2'),
        dptr(sk, torch.uint8, 2 * C, 'memo_state'), C, K, N, stream_handle(coeffs.device))
    check(rc, 'binf_poly_gauss_logp_memo_f64')
    return out


@_launcher
def hmc_sample_poly(q0, p0, u, q_out, accepted, n_accepted, e_before, e_after,
                    xs, ys, precision, prior_means, prior_vars, prior_first,
                    lp_pre, lp_post, timestep, dt_chain, nsteps, adapt, uprate,
                    downrate, mode=MODE_EXACT):
    """binf_hmc_sample_poly_f64 on torch's current stream (K <= 16
    coefficients, <= 1024 data points; ``mode | MODE_LANE_PER_CHAIN``: one lane
    per chain, <= 128 data points)."""
    C, K = _cd(q0)
    N = xs.numel()
    tau, tau_chain = _precision_args(precision, C, q0.device)
    rc = lib().binf_hmc_sample_poly_f64(
        dptr(q0, numel=C * K, name='q0'), dptr(p0, numel=C * K, name='p0'),
        dptr(u, numel=C, name='u'), dptr(q_out, numel=C * K, name='q_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=C, name='e_before'),
        dptr(e_after, numel=C, name='e_after'),
        dptr(xs, numel=N, name='xs'), dptr(ys, numel=N, name='ys'), tau,
        dptr(tau_chain, numel=C, name='precision'),
        dptr(prior_means, numel=K, name='prior_means'),
        dptr(prior_vars, numel=K, name='prior_vars'), int(bool(prior_first)),
        dptr(lp_pre, numel=C, name='lp_pre'), dptr(lp_post, numel=C, name='lp_post'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), C, K, N,
        int(nsteps), int(bool(adapt)), float(uprate), float(downrate),
        int(mode), stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_poly_f64')


class GibbsPolyArgs(ctypes.Structure):
    """``binf_gibbs_poly_args`` of include/binf_hip.h, field for field."""
    _fields_ = [('struct_size', _u64)] + \
        [(n, _vp) for n in ('coefficients', 'precision', 'coefficients_out', 'precision_out',
                            'rec_coefficients', 'rec_precision', 'accepted', 'n_accepted',
                            'e_before', 'e_after', 'xs', 'ys', 'prior_means', 'prior_vars',
                            'p0', 'u', 'g', 'dt_chain')] + \
        [(n, _f64) for n in ('timestep', 'uprate', 'downrate', 'stepsize', 'gp_shape',
                             'gp_rate', 'gamma_shape', 'gamma_rate')] + \
        [(n, _i64) for n in ('C', 'K', 'N', 'chain_offset')] + \
        [(n, _u64) for n in ('seed_m', 'off_m', 'stride_m', 'seed_u', 'off_u', 'stride_u',
                             'seed_g', 'off_g', 'stride_g')] + \
        [(n, _i32) for n in ('move', 'mode', 'nsteps', 'n', 'thin', 'n_adapt', 'prior_first',
                             'gp_where', 'zig', 'keep_precision')]


@_launcher
def gibbs_poly_sample_n(coefficients, precision, coefficients_out, precision_out, xs, ys, n,
                        thin=1, move=MOVE_HMC, mode=MODE_EXACT, nsteps=1, timestep=0.0,
                        dt_chain=None, n_adapt=0, uprate=1.05, downrate=0.95, stepsize=0.0,
                        prior_means=None, prior_vars=None, prior_first=False, gp_where=0,
                        gp_shape=1.0, gp_rate=0.0, gamma_shape=1.0, gamma_rate=0.0,
                        rec_coefficients=None, rec_precision=None, accepted=None,
                        n_accepted=None, e_before=None, e_after=None, p0=None, u=None, g=None,
                        streams=None, chain_offset=0, zig=True, keep_precision=False):
    """binf_gibbs_poly_sample_n_f64 on torch's current stream: n sweeps of the
    example's Gibbs loop in one launch.  ``streams`` = ``((seed, offset, stride),) * 3``
    for the momentum / proposal, acceptance and gamma draws that are not supplied."""
    C, K = _cd(coefficients)
    N = xs.numel()
    n, thin = int(n), int(thin)
    nrec = n // thin
    a = GibbsPolyArgs()
    a.struct_size = ctypes.sizeof(GibbsPolyArgs)
    a.coefficients = dptr(coefficients, numel=C * K, name='coefficients')
    a.precision = dptr(precision, numel=C, name='precision')
    a.coefficients_out = dptr(coefficients_out, numel=C * K, name='coefficients_out')
    a.precision_out = dptr(precision_out, numel=C, name='precision_out')
    a.rec_coefficients = dptr(rec_coefficients, numel=nrec * C * K, name='rec_coefficients')
    a.rec_precision = dptr(rec_precision, numel=nrec * C, name='rec_precision')
    a.accepted = dptr(accepted, torch.uint8, n * C, 'accepted')
    a.n_accepted = dptr(n_accepted, torch.int64, C, 'n_accepted')
    a.e_before = dptr(e_before, numel=n * C, name='e_before')
    a.e_after = dptr(e_after, numel=n * C, name='e_after')
    a.xs = dptr(xs, numel=N, name='xs')
    a.ys = dptr(ys, numel=N, name='ys')
    a.prior_means = dptr(prior_means, numel=K, name='prior_means')
    a.prior_vars = dptr(prior_vars, numel=K, name='prior_vars')
    a.p0 = dptr(p0, numel=n * C * K, name='p0')
    a.u = dptr(u, numel=n * C, name='u')
    a.g = dptr(g, numel=n * C, name='g')
    a.dt_chain = dptr(dt_chain, numel=C, name='dt_chain')
    a.timestep, a.uprate, a.downrate = float(timestep), float(uprate), float(downrate)
    a.stepsize = float(stepsize)
    a.gp_shape, a.gp_rate = float(gp_shape), float(gp_rate)
    a.gamma_shape, a.gamma_rate = float(gamma_shape), float(gamma_rate)
    a.C, a.K, a.N, a.chain_offset = C, K, N, int(chain_offset)
    sm, su, sg = streams if streams is not None else ((0, 0, 0),) * 3
    a.seed_m, a.off_m, a.stride_m = [int(v) for v in sm]
    a.seed_u, a.off_u, a.stride_u = [int(v) for v in su]
    a.seed_g, a.off_g, a.stride_g = [int(v) for v in sg]
    a.move, a.mode, a.nsteps, a.n, a.thin = int(move), int(mode), int(nsteps), n, thin
    a.n_adapt, a.prior_first, a.gp_where = int(n_adapt), int(bool(prior_first)), int(gp_where)
    a.zig = int(bool(zig))
    a.keep_precision = int(bool(keep_precision))
    rc = lib().binf_gibbs_poly_sample_n_f64(ctypes.byref(a), stream_handle(coefficients.device))
    check(rc, 'binf_gibbs_poly_sample_n_f64')


@_launcher
def jacobian_contract(jacobian, emgrad):
    """``dfm.dot(emgrad)`` of ``Likelihood._evaluate_gradient``, batched:
    ``jacobian`` ``[K x N]`` (shared) or ``[C x K x N]``, ``emgrad`` ``[C x N]`` or
    ``[N]`` (one chain) -> ``[C x K]`` / ``[K]``."""
    one = emgrad.dim() == 1
    r = emgrad.reshape(1, -1) if one else emgrad
    C, N = r.shape
    batched = jacobian.dim() == 3
    K = jacobian.shape[-2]
    if jacobian.shape[-1] != N or (batched and jacobian.shape[0] != C):
        raise ValueError('jacobi matrix %s does not fit the error-model gradient %s'
                         % (tuple(jacobian.shape), tuple(emgrad.shape)))
    out = torch.empty((C, K), dtype=torch.float64, device=r.device)
    rc = lib().binf_jacobian_contract_f64(
        dptr(jacobian, numel=(C if batched else 1) * K * N, name='jacobi matrix'),
        dptr(r, numel=C * N, name='error-model gradient'), dptr(out), C, K, N, int(batched),
        stream_handle(r.device))
    check(rc, 'binf_jacobian_contract_f64')
    return out.reshape(-1) if one else out


@_launcher
def sum_terms(terms, like=None):
    """``((t0 + t1) + t2) + ...`` in one launch (at most 16 terms); a term is a device
    vector (all of one shape), a 0-dim device tensor (ONE device double, broadcast --
    never read back to the host) or a Python / numpy scalar.  ``like``: a tensor of the
    result's shape when no vector is among the terms."""
    tens = [t for t in terms if isinstance(t, torch.Tensor) and t.dim() > 0]
    if not tens and like is None:
        raise TypeError('sum_terms: no device vector among the terms')
    ref = tens[0] if tens else like
    n = ref.numel()
    T = len(terms)
    ptrs = (_vp * T)()
    scal = (_f64 * T)()
    bc = (ctypes.c_uint8 * T)()
    any_bc = False
    for i, t in enumerate(terms):
        if isinstance(t, torch.Tensor) and t.dim() > 0:
            if t.shape != ref.shape:
                raise ValueError('sum_terms: term shapes differ (%s, %s)'
                                 % (tuple(t.shape), tuple(ref.shape)))
            ptrs[i] = dptr(t, numel=n, name='term %d' % i)
        elif isinstance(t, torch.Tensor):
            ptrs[i] = dptr(t, numel=1, name='term %d' % i)
            bc[i] = 1
            any_bc = True
        else:
            ptrs[i] = None
            scal[i] = float(t)
    out = torch.empty_like(ref)
    if any_bc:
        rc = lib().binf_sum_terms_bcast_f64(ptrs, scal, bc, T, dptr(out), n, stream_handle(ref.device))
        check(rc, 'binf_sum_terms_bcast_f64')
    else:
        rc = lib().binf_sum_terms_f64(ptrs, scal, T, dptr(out), n, stream_handle(ref.device))
        check(rc, 'binf_sum_terms_f64')
    return out


_grad_ws = {}


@_launcher
def poly_gauss_grad(coeffs, design, ys, precision):
    C, K = _cd(coeffs)
    N = ys.numel()
    if design.shape != (K, N):
        raise ValueError('design matrix must be [%d x %d], got %s'
                         % (K, N, tuple(design.shape)))
    tau, tau_chain = _precision_args(precision, C, coeffs.device)
    need = lib().binf_poly_gauss_grad_workspace_bytes(C, K, N)
    ws = None
    st = stream_handle(coeffs.device)
    if need > 0:
        # one scratch buffer per (device, stream, size): calls on the same stream
        # are ordered, calls on different streams must not share it
        key = (coeffs.device, st, need)
        ws = _grad_ws.get(key)
        if ws is None:
            ws = torch.empty(need // 8, dtype=torch.float64,
                             device=coeffs.device)
            while len(_grad_ws) >= 4:
                _grad_ws.pop(next(iter(_grad_ws)))
            _grad_ws[key] = ws
    out = torch.empty((C, K), dtype=torch.float64, device=coeffs.device)
    rc = lib().binf_poly_gauss_grad_f64(
        dptr(coeffs, numel=C * K, name='coeffs'),
        dptr(design, numel=K * N, name='design'), dptr(ys, numel=N, name='ys'),
        tau, dptr(tau_chain, numel=C, name='precision'), dptr(out),
        ws.data_ptr() if ws is not None else None, need, C, K, N, st)
    check(rc, 'binf_poly_gauss_grad_f64')
    return out


@_launcher
def poly_leapfrog(q, p, design, ys, precision, timestep, dt_chain, nsteps, mode=MODE_EXACT):
    """binf_poly_leapfrog_f64: the whole ``_leapfrog`` under the polynomial
    likelihood's force, in place on ``q`` and ``p`` (``[C x K]``)."""
    C, K = _cd(q)
    N = ys.numel()
    if design.shape != (K, N):
        raise ValueError('design matrix must be [%d x %d], got %s' % (K, N, tuple(design.shape)))
    tau, tau_chain = _precision_args(precision, C, q.device)
    need = lib().binf_poly_leapfrog_workspace_bytes(C, K, N)
    st = stream_handle(q.device)
    key = (q.device, st, 'leap', need)
    ws = _grad_ws.get(key)
    if ws is None:
        ws = torch.empty(max(1, need // 8), dtype=torch.float64, device=q.device)
        while len(_grad_ws) >= 4:
            _grad_ws.pop(next(iter(_grad_ws)))
        _grad_ws[key] = ws
    rc = lib().binf_poly_leapfrog_f64(
        dptr(q, numel=C * K, name='q'), dptr(p, numel=C * K, name='p'),
        dptr(design, numel=K * N, name='design'), dptr(ys, numel=N, name='ys'), tau,
        dptr(tau_chain, numel=C, name='precision'), ws.data_ptr(), need, C, K, N,
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), int(nsteps), int(mode), st)
    check(rc, 'binf_poly_leapfrog_f64')


@_launcher
def gamma_precision_update(g, lp_unit, prior_rate):
    C = g.numel()
    out = torch.empty(C, dtype=torch.float64, device=g.device)
    rc = lib().binf_gamma_precision_update_f64(
        dptr(g, numel=C, name='g'), dptr(lp_unit, numel=C, name='lp_unit'),
        float(prior_rate), dptr(out), C, stream_handle(g.device))
    check(rc, 'binf_gamma_precision_update_f64')
    return out


@_launcher
def gamma_logp(precision, shape, rate):
    """``(shape - 1) * log(precision) - precision * rate`` per chain
    (priors.py:10-25), one launch."""
    C = precision.numel()
    out = torch.empty(precision.shape, dtype=torch.float64, device=precision.device)
    rc = lib().binf_gamma_logp_f64(dptr(precision, numel=C, name='precision'), float(shape),
                                   float(rate), dptr(out), C, stream_handle(precision.device))
    check(rc, 'binf_gamma_logp_f64')
    return out


@_launcher
def hmc_sample_n_gauss(q0, p0, u, q_out, samples, accepted, n_accepted,
                       e_before, e_after, timestep, dt_chain, nsteps, n, thin,
                       k, x0, n_adapt, uprate, downrate, mode=MODE_EXACT):
    """binf_hmc_sample_n_gauss_f64 on torch's current stream."""
    C, D = _cd(q0)
    n = int(n)
    thin = int(thin)
    nrec = n // thin
    rc = lib().binf_hmc_sample_n_gauss_f64(
        dptr(q0, numel=C * D, name='q0'),
        dptr(p0, numel=n * C * D, name='p0'), dptr(u, numel=n * C, name='u'),
        dptr(q_out, numel=C * D, name='q_out'),
        dptr(samples, numel=nrec * C * D, name='samples'),
        dptr(accepted, torch.uint8, n * C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=n * C, name='e_before'),
        dptr(e_after, numel=n * C, name='e_after'), float(timestep),
        dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(nsteps), n, thin,
        float(k), float(x0), int(n_adapt), float(uprate), float(downrate),
        int(mode), stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_n_gauss_f64')


def gauss_waves_per_chain(C, D):
    """Waves the persistent kernel spreads a chain over for a [C x D] batch."""
    return lib().binf_hmc_gauss_waves_per_chain(int(C), int(D))


def gauss_persist_covers(D):
    """Shapes the persistent kernel (binf_hmc_sample_[n_]gauss_f64) accepts."""
    return 1 <= D <= 8192 and pairwise_tree_height(D) <= 6


@_launcher
def hmc_sample_gauss_big(q0, p0, u, q_out, accepted, n_accepted, e_before, e_after,
                         timestep, dt_chain, nsteps, k, x0, adapt, uprate, downrate,
                         mode=MODE_EXACT):
    """binf_hmc_sample_gauss_big_f64 (chains of any length) on torch's current
    stream; the chunk-sum scratch is allocated here (stream-ordered by torch's
    caching allocator)."""
    C, D = _cd(q0)
    nbytes = lib().binf_hmc_sample_gauss_big_workspace_bytes(C, D)
    ws = torch.empty(max(1, nbytes // 8), dtype=torch.float64, device=q0.device)
    rc = lib().binf_hmc_sample_gauss_big_f64(
        dptr(q0, numel=C * D, name='q0'), dptr(p0, numel=C * D, name='p0'),
        dptr(u, numel=C, name='u'), dptr(q_out, numel=C * D, name='q_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=C, name='e_before'), dptr(e_after, numel=C, name='e_after'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(nsteps),
        float(k), float(x0), int(bool(adapt)), float(uprate), float(downrate), int(mode),
        dptr(ws), nbytes, stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_gauss_big_f64')


@_launcher
def hmc_sample_gauss_big_rng(q0, q_out, accepted, n_accepted, e_before, e_after, timestep,
                             dt_chain, nsteps, k, x0, adapt, uprate, downrate, mode, seed,
                             offset, chain_offset=0):
    """binf_hmc_sample_gauss_big_rng_f64 (long chains, draws generated in the
    kernels) on torch's current stream."""
    C, D = _cd(q0)
    nbytes = lib().binf_hmc_sample_gauss_big_workspace_bytes(C, D)
    ws = torch.empty(max(1, nbytes // 8), dtype=torch.float64, device=q0.device)
    rc = lib().binf_hmc_sample_gauss_big_rng_f64(
        dptr(q0, numel=C * D, name='q0'), dptr(q_out, numel=C * D, name='q_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=C, name='e_before'), dptr(e_after, numel=C, name='e_after'),
        float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(nsteps),
        float(k), float(x0), int(bool(adapt)), float(uprate), float(downrate), int(mode),
        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(chain_offset), dptr(ws),
        nbytes, stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_gauss_big_rng_f64')


_big_n_ws = {}


@_launcher
def hmc_sample_n_gauss_big(q0, p0, u, q_out, samples, accepted, n_accepted, e_before, e_after,
                           timestep, dt_chain, nsteps, n, thin, k, x0, n_adapt, uprate, downrate,
                           mode=MODE_EXACT, rng=None):
    """binf_hmc_sample_n_gauss_big_f64 (draws supplied: ``p0`` [n, C, D], ``u``
    [n, C]) or, with ``rng = (seed, offset, chain_offset)``, its _rng form."""
    C, D = _cd(q0)
    n, thin = int(n), int(thin)
    nrec = n // thin
    need = lib().binf_hmc_sample_n_gauss_big_workspace_bytes(C, D)
    st = stream_handle(q0.device)
    key = (q0.device, st, need)
    ws = _big_n_ws.get(key)
    if ws is None:
        ws = torch.empty(need // 8, dtype=torch.float64, device=q0.device)
        _big_n_ws.clear()                      # one long-chain scratch at a time
        _big_n_ws[key] = ws
    common = (dptr(samples, numel=nrec * C * D, name='samples'),
              dptr(accepted, torch.uint8, n * C, 'accepted'),
              dptr(n_accepted, torch.int64, C, 'n_accepted'),
              dptr(e_before, numel=n * C, name='e_before'), dptr(e_after, numel=n * C, name='e_after'),
              float(timestep), dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(nsteps), n, thin,
              float(k), float(x0), int(n_adapt), float(uprate), float(downrate), int(mode))
    if rng is None:
        rc = lib().binf_hmc_sample_n_gauss_big_f64(
            dptr(q0, numel=C * D, name='q0'), dptr(p0, numel=n * C * D, name='p0'),
            dptr(u, numel=n * C, name='u'), dptr(q_out, numel=C * D, name='q_out'), *common,
            ws.data_ptr(), need, st)
        check(rc, 'binf_hmc_sample_n_gauss_big_f64')
    else:
        seed, offset, coff = rng
        rc = lib().binf_hmc_sample_n_gauss_big_rng_f64(
            dptr(q0, numel=C * D, name='q0'), dptr(q_out, numel=C * D, name='q_out'), *common,
            int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(coff), ws.data_ptr(), need, st)
        check(rc, 'binf_hmc_sample_n_gauss_big_rng_f64')


def hmc_gauss_big_rng_draws(C, D, seed, offset, device, chain_offset=0):
    """(p0 [C, D], u [C]): the draws hmc_sample_gauss_big_rng consumes for global
    chains ``chain_offset .. chain_offset + C - 1``."""
    C, D = int(C), int(D)
    p0 = torch.empty((C, D), dtype=torch.float64, device=device)
    u = torch.empty(C, dtype=torch.float64, device=device)
    with torch.cuda.device(p0.device):
        rc = lib().binf_hmc_gauss_big_rng_draws_f64(
            dptr(p0), dptr(u), C, D, int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
            int(chain_offset), stream_handle(p0.device))
    check(rc, 'binf_hmc_gauss_big_rng_draws_f64')
    return p0, u


@_launcher
def hmc_sample_n_gauss_rng(q0, q_out, samples, accepted, n_accepted, e_before,
                           e_after, timestep, dt_chain, nsteps, n, thin, k, x0,
                           n_adapt, uprate, downrate, mode, seed, offset, chain_offset=0):
    """binf_hmc_sample_n_gauss_rng_f64 (draws generated in the kernel) on
    torch's current stream."""
    C, D = _cd(q0)
    n = int(n)
    thin = int(thin)
    nrec = n // thin
    rc = lib().binf_hmc_sample_n_gauss_rng_f64(
        dptr(q0, numel=C * D, name='q0'),
        dptr(q_out, numel=C * D, name='q_out'),
        dptr(samples, numel=nrec * C * D, name='samples'),
        dptr(accepted, torch.uint8, n * C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'),
        dptr(e_before, numel=n * C, name='e_before'),
        dptr(e_after, numel=n * C, name='e_after'), float(timestep),
        dptr(dt_chain, numel=C, name='dt_chain'), C, D, int(nsteps), n, thin,
        float(k), float(x0), int(n_adapt), float(uprate), float(downrate),
        int(mode), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
        int(chain_offset), stream_handle(q0.device))
    check(rc, 'binf_hmc_sample_n_gauss_rng_f64')


def hmc_gauss_rng_draws(n, C, D, seed, offset, device, chain_offset=0, out=None):
    """(p0 [n, C, D], u [n, C]): the draws the fused-generator kernel consumes
    for (seed, offset) and global chains ``chain_offset .. chain_offset + C - 1``
    (``out = (p0, u)``: fill the caller's buffers instead of new ones)."""
    n, C, D = int(n), int(C), int(D)
    if out is None:
        p0 = torch.empty((n, C, D), dtype=torch.float64, device=device)
        u = torch.empty((n, C), dtype=torch.float64, device=device)
    else:
        p0, u = out
    with torch.cuda.device(p0.device):
        rc = lib().binf_hmc_gauss_rng_draws_f64(
            dptr(p0, numel=n * C * D, name='p0'), dptr(u, numel=n * C, name='u'), C, D, n,
            int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(chain_offset),
            stream_handle(p0.device))
    check(rc, 'binf_hmc_gauss_rng_draws_f64')
    return p0, u


@_launcher
def pairdist_forward(x, pair_i, pair_j):
    C, D = _cd(x)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    P = pair_i.numel()
    out = torch.empty((C, P), dtype=torch.float64, device=x.device)
    rc = lib().binf_pairdist_forward_f64(
        dptr(x, numel=C * D, name='x'), dptr(pair_i, torch.int32, P, 'pair_i'),
        dptr(pair_j, torch.int32, P, 'pair_j'), dptr(out), C, D // 3, P,
        stream_handle(x.device))
    check(rc, 'binf_pairdist_forward_f64')
    return out


_chi2_ws = {}


def _pairdist_chi2_workspace(C, n, P, device):
    """(pointer, bytes) of the scratch for chi^2 by chunks (few chains, or more than 2048 beads:
    binf_pairdist_chi2_workspace_bytes), one buffer per device and stream, grown on demand;
    ``(None, 0)`` when the library does not ask for one."""
    need = lib().binf_pairdist_chi2_workspace_bytes(C, n, P)
    if need <= 0:
        return None, 0
    key = (device, stream_handle(device))
    ws = _chi2_ws.get(key)
    if ws is None or ws.numel() * 8 < need:
        ws = _chi2_ws[key] = torch.empty(need // 8, dtype=torch.float64, device=device)
    return ws.data_ptr(), ws.numel() * 8


@_launcher
def pairdist_gauss_logp(x, pair_i, pair_j, ys, precision):
    """Gaussian log-likelihood of the pair distances, fused (no [C x n_pairs]
    intermediate)."""
    C, D = _cd(x)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    P = pair_i.numel()
    tau, tau_chain = _precision_args(precision, C, x.device)
    out = torch.empty(C, dtype=torch.float64, device=x.device)
    rc = lib().binf_pairdist_gauss_logp_f64(
        dptr(x, numel=C * D, name='x'), dptr(pair_i, torch.int32, P, 'pair_i'),
        dptr(pair_j, torch.int32, P, 'pair_j'), dptr(ys, numel=P, name='ys'), tau,
        dptr(tau_chain, numel=C, name='precision'), dptr(out), C, D // 3, P,
        *_pairdist_chi2_workspace(C, D // 3, P, x.device), stream_handle(x.device))
    check(rc, 'binf_pairdist_gauss_logp_f64')
    return out


@_launcher
def pairdist_gauss_logp_memo(x, pair_i, pair_j, ys, precision, memo):
    """binf_pairdist_gauss_logp_memo_f64; ``memo = new_chi2_memo(C, 3 * n_beads, device)``."""
    C, D = _cd(x)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    P = pair_i.numel()
    tau, tau_chain = _precision_args(precision, C, x.device)
    mx, ms, sk = memo
    out = torch.empty(C, dtype=torch.float64, device=x.device)
    rc = lib().binf_pairdist_gauss_logp_memo_f64(
        dptr(x, numel=C * D, name='x'), dptr(pair_i, torch.int32, P, 'pair_i'),
        dptr(pair_j, torch.int32, P, 'pair_j'), dptr(ys, numel=P, name='ys'), tau,
        dptr(tau_chain, numel=C, name='precision'), dptr(out), dptr(mx, numel=2 * C * D, name='memo_x'),
        dptr(ms, numel=2 * C, name='memo_chi2'), dptr(sk, torch.uint8, 2 * C, 'memo_state'), C, D // 3, P,
        *_pairdist_chi2_workspace(C, D // 3, P, x.device), stream_handle(x.device))
    check(rc, 'binf_pairdist_gauss_logp_memo_f64')
    return out


@_launcher
@_launcher
def pairdist_hmc_energy(x, p, pair_i, pair_j, ys, precision, prior, prior_first, memo=None,
                        want_log_prob=False, terms=None):
    """binf_pairdist_hmc_energy_f64: ``0.5 * sum(p**2) - log_prob`` of the restraint
    posterior in one launch.  ``prior`` = ``(k, x0)`` of an isotropic Gaussian or None;
    the components are added in the order ``terms`` gives -- a list of ``'prior'``,
    ``'lik'`` and up to two constants of the move (``[C]`` device tensors or floats) --
    default: prior and likelihood as ``prior_first`` says.  ``memo = new_chi2_memo(C,
    3 * n_beads, device)`` or None.  Returns the energy, or ``(energy, log_prob)``."""
    C, D = _cd(x)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    P = pair_i.numel()
    tau, tau_chain = _precision_args(precision, C, x.device)
    k, x0 = prior if prior is not None else (0.0, 0.0)
    if terms is None:
        terms = ['lik'] if prior is None else (['prior', 'lik'] if prior_first else ['lik', 'prior'])
    kinds, extras = [], []
    for t in terms:
        if isinstance(t, str):
            if t not in ('prior', 'lik') or (t == 'prior' and prior is None):
                raise ValueError('pairdist_hmc_energy: unknown term %r' % (t,))
            kinds.append(0 if t == 'prior' else 1)
        else:
            if len(extras) == 2:
                raise ValueError('pairdist_hmc_energy: at most two constant terms')
            kinds.append(2 + len(extras))
            extras.append(t)
    ptrs, scalars = [None, None], [0.0, 0.0]
    for n_, e in enumerate(extras):
        if isinstance(e, torch.Tensor) and e.dim() > 0:
            ptrs[n_] = dptr(e, numel=C, name='constant term')
        else:
            scalars[n_] = float(e)
    kind_arr = (_i32 * 4)(*(kinds + [1] * (4 - len(kinds))))
    mx, ms, st = memo if memo is not None else (None, None, None)
    energy = torch.empty(C, dtype=torch.float64, device=x.device)
    lp = torch.empty(C, dtype=torch.float64, device=x.device) if want_log_prob else None
    rc = lib().binf_pairdist_hmc_energy_f64(
        dptr(x, numel=C * D, name='x'), dptr(p, numel=C * D, name='p'),
        dptr(pair_i, torch.int32, P, 'pair_i'), dptr(pair_j, torch.int32, P, 'pair_j'),
        dptr(ys, numel=P, name='ys'), tau, dptr(tau_chain, numel=C, name='precision'),
        float(k), float(x0), len(kinds), kind_arr, ptrs[0], scalars[0], ptrs[1], scalars[1],
        dptr(energy), dptr(lp),
        dptr(mx, numel=2 * C * D, name='memo_x'), dptr(ms, numel=2 * C, name='memo_chi2'),
        dptr(st, torch.uint8, 2 * C, 'memo_state'), C, D // 3, P,
        *_pairdist_chi2_workspace(C, D // 3, P, x.device), stream_handle(x.device))
    check(rc, 'binf_pairdist_hmc_energy_f64')
    return (energy, lp) if want_log_prob else energy


@_launcher
def pairdist_pack_targets(ymat):
    """binf_pairdist_pack_targets_f64: the targets in the order the 32..256-bead force
    kernels hold them (pass as ``packed=`` to the two functions below); None for bead
    counts that have no packed form."""
    n = ymat.shape[0]
    nbytes = lib().binf_pairdist_packed_targets_bytes(n)
    if nbytes <= 0:
        return None
    out = torch.empty(nbytes // 8, dtype=torch.float64, device=ymat.device)
    rc = lib().binf_pairdist_pack_targets_f64(dptr(ymat, numel=n * n, name='ymat'), dptr(out), n,
                                              stream_handle(ymat.device))
    check(rc, 'binf_pairdist_pack_targets_f64')
    return out


def _packed_ptr(packed, n):
    if packed is None:
        return None
    return dptr(packed, numel=lib().binf_pairdist_packed_targets_bytes(n) // 8, name='packed')


_tiles_ws = {}


def _pairdist_tiles_workspace(C, n, packed, device):
    """(pointer, bytes) of the scratch for the tiles' partial sums (few chains of 257..1024 beads:
    binf_pairdist_tiles_workspace_bytes), one buffer per device and stream, grown on demand;
    ``(None, 0)`` when the library does not ask for one."""
    if packed is None:
        return None, 0
    need = lib().binf_pairdist_tiles_workspace_bytes(C, n)
    if need <= 0:
        return None, 0
    key = (device, stream_handle(device))
    ws = _tiles_ws.get(key)
    if ws is None or ws.numel() * 8 < need:
        # scratch, not state: never more than a quarter of what is free (many chains of several
        # thousand beads would ask for tens of GB) -- without it the library takes its other kernels
        _tiles_ws.pop(key, None)
        if need > torch.cuda.mem_get_info(device)[0] // 4:
            return None, 0
        ws = _tiles_ws[key] = torch.empty(need // 8, dtype=torch.float64, device=device)
    return ws.data_ptr(), ws.numel() * 8


@_launcher
def pairdist_gauss_grad(x, ymat, precision, packed=None):
    C, D = _cd(x)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    n = D // 3
    tau, tau_chain = _precision_args(precision, C, x.device)
    out = torch.empty_like(x)
    ws, ws_bytes = _pairdist_tiles_workspace(C, n, packed, x.device)
    rc = lib().binf_pairdist_gauss_grad_packed_f64(
        dptr(x, numel=C * D, name='x'), dptr(ymat, numel=n * n, name='ymat'),
        _packed_ptr(packed, n), tau, dptr(tau_chain, numel=C, name='precision'), dptr(out), C, n,
        ws, ws_bytes, stream_handle(x.device))
    check(rc, 'binf_pairdist_gauss_grad_packed_f64')
    return out


@_launcher
def rwmc_propose(state, stepsize, change=None, seed=0, offset=0, chain_offset=0, out=None):
    """``state + uniform(-stepsize, stepsize)`` (samplers.py:81-83); ``change``
    supplied, or drawn on the device from the Philox stream (seed, offset)."""
    C, K = _cd(state)
    if out is None:
        out = torch.empty_like(state)
    rc = lib().binf_rwmc_propose_f64(
        dptr(state, numel=C * K, name='state'), dptr(change, numel=C * K, name='change'),
        dptr(out, numel=C * K, name='proposal'), float(stepsize), C, K,
        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(chain_offset),
        stream_handle(state.device))
    check(rc, 'binf_rwmc_propose_f64')
    return out


@_launcher
def rwmc_accept(proposal, state, lp_old, lp_new, state_out, accepted=None, n_accepted=None,
                u=None, seed=0, offset=0, chain_offset=0):
    """Metropolis test with numpy's exp + select + counters (samplers.py:84-90);
    ``u`` supplied or drawn on the device."""
    C, K = _cd(proposal)
    rc = lib().binf_rwmc_accept_f64(
        dptr(proposal, numel=C * K, name='proposal'), dptr(state, numel=C * K, name='state'),
        dptr(lp_old, numel=C, name='lp_old'), dptr(lp_new, numel=C, name='lp_new'),
        dptr(u, numel=C, name='u'), dptr(state_out, numel=C * K, name='state_out'),
        dptr(accepted, torch.uint8, C, 'accepted'),
        dptr(n_accepted, torch.int64, C, 'n_accepted'), C, K,
        int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(chain_offset),
        stream_handle(proposal.device))
    check(rc, 'binf_rwmc_accept_f64')


def philox4x32_10(counter, key):
    c = (ctypes.c_uint32 * 4)(*[int(x) & 0xffffffff for x in counter])
    k = (ctypes.c_uint32 * 2)(*[int(x) & 0xffffffff for x in key])
    o = (ctypes.c_uint32 * 4)()
    check(lib().binf_rng_philox4x32_10(c, k, o), 'binf_rng_philox4x32_10')
    return [int(x) for x in o]


@_launcher
def rng_fill(kind, out, seed, offset, shape=None, elem_offset=0):
    """Fill the contiguous f64 device tensor `out` with uniform / normal /
    gamma(shape) draws of the Philox stream (seed, offset); ``out[l]`` receives
    global element ``elem_offset + l`` of the stream."""
    n = out.numel()
    p = dptr(out, numel=n, name='out')
    st = stream_handle(out.device)
    seed, offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1)
    e0 = int(elem_offset)
    if kind == 'uniform':
        rc = lib().binf_rng_uniform_f64(p, n, seed, offset, e0, st)
    elif kind == 'normal':
        rc = lib().binf_rng_normal_f64(p, n, seed, offset, e0, st)
    elif kind == 'normal_zig':
        rc = lib().binf_rng_normal_zig_f64(p, n, seed, offset, e0, st)
    elif kind == 'gamma':
        rc = lib().binf_rng_gamma_f64(p, n, float(shape), seed, offset, e0, st)
    else:
        raise ValueError('unknown draw kind %r' % (kind,))
    check(rc, 'binf_rng_%s_f64' % kind)
    return out


@_launcher
def rng_fill_normal_zig_uniform(normals, uniforms, seed, offset_normals, offset_uniforms,
                                elem_offset_normals=0, elem_offset_uniforms=0):
    """binf_rng_normal_zig_uniform_f64: ``rng_fill('normal_zig', normals, ...)`` and
    ``rng_fill('uniform', uniforms, ...)`` in one launch."""
    m = 2 ** 64 - 1
    rc = lib().binf_rng_normal_zig_uniform_f64(
        dptr(normals, numel=normals.numel(), name='normals'), normals.numel(),
        dptr(uniforms, numel=uniforms.numel(), name='uniforms'), uniforms.numel(),
        int(seed) & m, int(offset_normals) & m, int(offset_uniforms) & m,
        int(elem_offset_normals), int(elem_offset_uniforms), stream_handle(normals.device))
    check(rc, 'binf_rng_normal_zig_uniform_f64')


@_launcher
def pairdist_leapfrog(q, p, ymat, precision, prior, prior_first, timestep,
                      dt_chain, nsteps, mode=MODE_EXACT, packed=None, q_from=None):
    """In-place leapfrog of (q, p) for the restraint posterior; prior is None
    or (k, x0) of an isotropic Gaussian on the coordinates.  ``q_from``: start positions
    read from there instead of ``q`` (which then only receives the end positions)."""
    C, D = _cd(q)
    if D % 3:
        raise ValueError('coordinates must be [n_chains, 3 * n_beads]')
    n = D // 3
    tau, tau_chain = _precision_args(precision, C, q.device)
    k, x0 = prior if prior is not None else (0.0, 0.0)
    ws, ws_bytes = _pairdist_tiles_workspace(C, n, packed, q.device)
    rc = lib().binf_pairdist_leapfrog_packed_f64(
        dptr(q, numel=C * D, name='q'), dptr(q_from, numel=C * D, name='q_from'),
        dptr(p, numel=C * D, name='p'),
        dptr(ymat, numel=n * n, name='ymat'), _packed_ptr(packed, n), tau,
        dptr(tau_chain, numel=C, name='precision'), int(prior is not None),
        float(k), float(x0), int(bool(prior_first)), float(timestep),
        dptr(dt_chain, numel=C, name='dt_chain'), int(nsteps), C, n, int(mode),
        ws, ws_bytes, stream_handle(q.device))
    check(rc, 'binf_pairdist_leapfrog_packed_f64')
