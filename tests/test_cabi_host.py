"""The drop-in boundary bound WITHOUT Python: tests/cabi/host_check.cpp dlopens
libbinf_hip.so, resolves its entry points by name, feeds them hipMalloc'ed pointers on
a stream of its own and compares with the C restatement of the reference path
(binf/samplers/hmc.py:92-164) bit for bit -- one fused transition, per-chain step sizes
with adaption, n transitions per launch, the per-step tier composed by the caller,
hipGraph capture / replay, error codes.  No torch allocator, no ctypes in that process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

LIB = os.path.join(ROOT, 'binf_amd', 'csrc', 'libbinf_hip.so')
ORACLE = os.path.join(ROOT, 'oracle', 'liboracle_c.so')


def _program():
    from oracle import c_oracle
    c_oracle.build()
    if not os.path.exists(LIB):
        entry.build()
    return entry.build_host_check()


def test_host_program_resolves_the_entry_points_it_binds():
    """CPU: the program builds against include/binf_hip.h alone, dlopens the library and
    finds every symbol it binds; the library's ABI number is the header's."""
    out = subprocess.run([_program(), '--symbols', LIB], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith('OK symbols'), out.stdout + out.stderr


@pytest.mark.gpu
def test_host_program_matches_the_oracle_without_python_in_the_process():
    out = subprocess.run([_program(), LIB, ORACLE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith('OK ') and 'FAIL' not in out.stdout, out.stdout
    assert int(last.split()[1]) >= 60
