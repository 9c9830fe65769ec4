"""CPU tests of the C-ABI boundary: the library loads, exports every symbol
include/binf_hip.h declares, and its reduction geometry is numpy's."""
import ctypes
import os
import re

import numpy as np
import pytest

from binf_amd import _native
from conftest import ROOT
from oracle import ref_numpy as R


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'binf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(binf_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_reports_abi_version():
    assert _native.lib().binf_abi_version() == _native.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert 'binf_hmc_sample_gauss_f64' in names
    raw = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'missing export: ' + n
        assert n in _native.SIGNATURES, 'no ctypes signature for ' + n
    assert sorted(_native.SIGNATURES) == names


def header_prototypes():
    """{name: (return C type, [argument C types])} parsed from include/binf_hip.h: comments
    stripped, every `type name(args);` at file scope; an argument's type is everything but
    its (optional) name, normalised (`const double *const *` -> 'const double*const*')."""
    text = open(os.path.join(ROOT, 'include', 'binf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'^\s*#.*$', '', text, flags=re.M)
    protos = {}
    for m in re.finditer(r'([A-Za-z_][A-Za-z0-9_ \*]*?)\b(binf_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)

        def norm(t):
            return re.sub(r'\s+', ' ', t.replace('*', ' * ')).strip().replace(' *', '*').replace('* ', '*')
        arg_types = []
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                arr = re.match(r'^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*\d*\s*\]$', a, flags=re.S)
                if arr:                                  # `const uint32_t counter[4]` is a pointer
                    arg_types.append(norm(arr.group(1) + '*'))
                    continue
                # drop the parameter name (last identifier, unless the declarator is a bare type)
                mm = re.match(r'^(.*?[\*\s])([A-Za-z_][A-Za-z0-9_]*)$', a, flags=re.S)
                base = mm.group(1) if mm and mm.group(2) not in ('int', 'double', 'void', 'char') else a
                arg_types.append(norm(base))
        protos[name] = (norm(ret.replace('extern', '').replace('"C"', '')), arg_types)
    return protos


# what each C type must be bound as in binf_amd/_native.py: SIGNATURES
def _ctype_of(c):
    if c.endswith('*'):
        return 'pointer'
    return {'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64,
            'uint32_t': ctypes.c_uint32, 'double': ctypes.c_double, 'size_t': ctypes.c_size_t,
            'int': ctypes.c_int}[c]


def _is_pointer_binding(t):
    return t is ctypes.c_void_p or t is ctypes.c_char_p or (isinstance(t, type) and
                                                            issubclass(t, ctypes._Pointer))


def test_ctypes_signatures_match_the_header_prototypes():
    """Arity and C type of every argument (and the return type) of every entry in
    _native.SIGNATURES against the prototype in include/binf_hip.h -- the table is
    written by hand and a call through a wrong one corrupts arguments silently (an int64
    passed as int32, a double in an integer register)."""
    protos = header_prototypes()
    assert sorted(protos) == sorted(_native.SIGNATURES)
    assert len(protos) >= 59
    for name, (ret, args) in sorted(protos.items()):
        res, bound = _native.SIGNATURES[name]
        assert res is _ctype_of(ret), '%s: returns %s, bound as %s' % (name, ret, res)
        assert len(bound) == len(args), '%s: %d arguments in the header, %d bound' % (
            name, len(args), len(bound))
        for i, (c, b) in enumerate(zip(args, bound)):
            want = _ctype_of(c)
            if want == 'pointer':
                assert _is_pointer_binding(b), '%s arg %d: %s bound as %s' % (name, i, c, b)
            else:
                assert b is want, '%s arg %d: %s bound as %s' % (name, i, c, b)
    # the parser itself: a prototype it must read exactly
    assert protos['binf_sum_terms_bcast_f64'] == ('int32_t', [
        'const double*const*', 'const double*', 'const uint8_t*', 'int32_t', 'double*', 'int64_t', 'void*'])
    assert protos['binf_abi_version'] == ('int32_t', [])


def test_memo_entry_points_refuse_before_touching_the_memo():
    """A shape the reduction would refuse is refused BEFORE the memo's check kernel runs
    (it rewrites the stored arguments of a missed row; their chi^2 is stored by the
    reduction after it -- a refusal in between would leave a matching entry with a stale
    sum, a silent wrong hit on the next call).  No launch, no dereference: host checks."""
    L = _native.lib()
    fake = [i << 40 for i in range(1, 10)]
    big = (1 << 31) + 5
    rc = L.binf_poly_gauss_logp_memo_f64(fake[0], fake[1], fake[2], 1.0, None, fake[3], fake[4],
                                         fake[5], fake[6], big, 4, 20, None)
    assert rc == _native.E_UNSUPPORTED and 'too large' in _native.last_error()
    rc = L.binf_pairdist_gauss_logp_memo_f64(fake[0], fake[1], fake[2], fake[3], 1.0, None, fake[4],
                                             fake[5], fake[6], fake[7], big, 4, 6, None, 0, None)
    assert rc == _native.E_UNSUPPORTED and 'too large' in _native.last_error()


def test_alias_refusals_without_gpu():
    """Partial overlaps the kernels cannot survive are refused on the host (BINF_E_ALIAS)."""
    L = _native.lib()
    base = 1 << 40
    C, D, n = 4, 9000, 3
    CD = C * D * 8
    q0, p0, u, q_out, samples = base, base + 10 * CD, base + 20 * CD, base + 30 * CD, base + 40 * CD
    acc, ws = base + 60 * CD, base + 70 * CD
    need = L.binf_hmc_sample_n_gauss_big_workspace_bytes(C, D)

    def call(samples_, q_out_=q_out, p0_=p0):
        return L.binf_hmc_sample_n_gauss_big_f64(q0, p0_, u, q_out_, samples_, acc, None, None, None, 0.1,
                                                 None, C, D, 2, n, 1, 1.0, 0.0, 0, 1.05, 0.95, 0, ws,
                                                 need, None)
    for bad in (q0 + 8, q_out - CD + 8, p0 + 2 * CD, ws + 16):        # samples against q0 / q_out / p0 / workspace
        assert call(bad) == _native.E_ALIAS, hex(bad)
    assert call(samples, q_out_=p0 + 2 * CD + 8) == _native.E_ALIAS     # q_out inside a later momentum block
    # the pair-distance leapfrog's start buffer
    nb = 10
    q, p, qf, ymat = base, base + (1 << 30), base + (2 << 30), base + (3 << 30)
    args = lambda qf_, p_=p: (q, qf_, p_, ymat, None, 1.0, None, 0, 0.0, 0.0, 0, 0.01, None, 2, C, nb, 0, None, 0, None)
    assert L.binf_pairdist_leapfrog_packed_f64(*args(q + 8)) == _native.E_ALIAS
    assert L.binf_pairdist_leapfrog_packed_f64(*args(p + 8)) == _native.E_ALIAS
    assert L.binf_pairdist_leapfrog_packed_f64(*args(qf, p_=q + 16)) == _native.E_ALIAS
    # a broadcast term inside the output of the term sum
    terms = (ctypes.c_void_p * 2)(base, base + 4096 + 8)
    flags = (ctypes.c_uint8 * 2)(0, 1)
    assert L.binf_sum_terms_bcast_f64(terms, None, flags, 2, base + 4096, 16, None) == _native.E_ALIAS


def test_error_codes_and_text_without_gpu():
    L = _native.lib()
    rc = L.binf_hmc_sample_gauss_f64(None, None, None, None, None, None, None,
                                     None, 0.1, None, 4, 0, 1, 1.0, 0.0, 0,
                                     1.05, 0.95, 0, None)
    assert rc == _native.E_ARG and 'D>=1' in _native.last_error()
    # never dereferenced: argument validation returns before any launch
    fake = [i << 40 for i in range(1, 6)]
    rc = L.binf_hmc_sample_gauss_f64(fake[0], fake[1], fake[2], fake[3], fake[4],
                                     None, None, None, 0.1, None, 4, 9000, 1,
                                     1.0, 0.0, 0, 1.05, 0.95, 0, None)
    assert rc == _native.E_UNSUPPORTED
    with pytest.raises(NotImplementedError):
        _native.check(rc, 'x')
    rc = L.binf_hmc_sample_gauss_f64(fake[0], fake[1], fake[2], fake[0] + 8,
                                     fake[4], None, None, None, 0.1, None, 4, 64,
                                     1, 1.0, 0.0, 0, 1.05, 0.95, 0, None)
    assert rc == _native.E_ALIAS
    rc = L.binf_row_sum_f64(None, None, -1, 4, 0, 0.0, 1.0, None)
    assert rc == _native.E_ARG
    with pytest.raises(ValueError):
        _native.check(rc, 'x')


def test_dptr_rejects_host_and_misshapen_tensors():
    import torch
    t = torch.zeros(4, dtype=torch.float64)
    with pytest.raises(ValueError):
        _native.dptr(t)
    with pytest.raises(TypeError):
        _native.dptr(np.zeros(4))


@pytest.mark.parametrize('n', [1, 7, 8, 100, 128, 129, 200, 258, 260, 300, 768,
                               920, 921, 1000, 1023, 1024, 1025, 4096, 5000,
                               16384, 100003])
def test_tree_walk_matches_numpy_recursion(n):
    """The padded-path walk the kernels use enumerates exactly the leaves of
    numpy's pairwise recursion, with their depths."""
    leaves, tree = R.pairwise_leaves(n)

    def height(t):
        return 0 if isinstance(t, int) else 1 + max(height(t[0]), height(t[1]))

    def depths(t, d, out):
        if isinstance(t, int):
            out[t] = d
        else:
            depths(t[0], d + 1, out)
            depths(t[1], d + 1, out)

    H = _native.pairwise_tree_height(n)
    assert H == height(tree)
    want_depth = {}
    depths(tree, 0, want_depth)
    canon = []
    for path in range(1 << H):
        off, ln, depth, c = _native.pairwise_leaf(n, H, path)
        if c:
            canon.append((off, ln, depth))
        else:
            # redundant path: same leaf as the canonical path above it
            base = path & ~((1 << (H - depth)) - 1)
            assert _native.pairwise_leaf(n, H, base)[:3] == (off, ln, depth)
    assert [(o, l) for o, l, _ in canon] == leaves
    assert [d for _, _, d in canon] == [want_depth[i] for i in range(len(leaves))]


def test_fused_kernel_coverage_claim():
    # one wave per chain: every D <= 920 and all multiples of 8 up to 1024
    for D in list(range(1, 921)) + list(range(928, 1025, 8)):
        assert _native.pairwise_tree_height(D) <= 3, D
    # header: every D <= 7400 and all multiples of 64 up to 8192 have height <= 6
    for D in list(range(1, 7401)) + list(range(7424, 8193, 64)):
        assert _native.pairwise_tree_height(D) <= 6, D


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under binf_amd/ may import,
    load or execute it (a product path through the oracle would void every
    parity claim)."""
    import glob
    pkg = os.path.join(ROOT, 'binf_amd')
    files = glob.glob(os.path.join(pkg, '**', '*.py'), recursive=True) + \
        glob.glob(os.path.join(pkg, 'csrc', '*'))
    assert len(files) > 15
    for f in files:
        if f.endswith('.so') or os.path.isdir(f):
            continue
        text = open(f, errors='replace').read()
        assert 'oracle' not in text, f


def test_missing_library_fails_loudly(monkeypatch):
    from binf_amd import _native as N
    monkeypatch.setattr(N, '_lib', None)
    monkeypatch.setattr(N, 'LIB_PATH', '/nonexistent/libbinf_hip.so')
    with pytest.raises(N.NativeLibraryError):
        N.lib()


def test_philox4x32_10_known_answers():
    """Random123's published known-answer vectors for philox4x32-10."""
    kat = [([0, 0, 0, 0], [0, 0],
            [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2,
            [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
            [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert _native.philox4x32_10(ctr, key) == want


def test_gibbs_args_struct_layout_matches_the_header(tmp_path):
    """binf_gibbs_poly_args is passed by pointer: the ctypes mirror must agree with
    what a C compiler makes of include/binf_hip.h, field by field."""
    import ctypes
    import subprocess
    fields = [f[0] for f in _native.GibbsPolyArgs._fields_]
    src = tmp_path / 'layout.c'
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "binf_hip.h"\n'
        'int main(void) {\n'
        '  printf("%zu\\n", sizeof(binf_gibbs_poly_args));\n' +
        ''.join('  printf("%%zu\\n", offsetof(binf_gibbs_poly_args, %s));\n' % f for f in fields) +
        '  return 0; }\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_native.GibbsPolyArgs)
    for f, off in zip(fields, out[1:]):
        assert getattr(_native.GibbsPolyArgs, f).offset == off, f
    # every field of the C struct is mirrored (same count as declared in the header)
    hdr = open(os.path.join(ROOT, 'include', 'binf_hip.h')).read()
    body = hdr[hdr.index('typedef struct binf_gibbs_poly_args {'):hdr.index('} binf_gibbs_poly_args;')]
    import re
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    declared = []
    for stmt in body.split('{', 1)[1].split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        names = stmt.replace('*', ' ').split(',')
        declared.append(names[0].split()[-1])
        declared += [n.strip() for n in names[1:]]
    assert declared == fields
