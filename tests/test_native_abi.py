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
