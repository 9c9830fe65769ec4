import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # The shared library is a build artefact (git-ignored): make sure it exists
    # and is current before any test imports it.  hipcc cross-compiles gfx950
    # without a GPU; on the GPU box the prebuilt file travels with the snapshot.
    lib = os.path.join(ROOT, 'binf_amd', 'csrc', 'libbinf_hip.so')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(lib) and os.path.exists(hipcc):
        import __graft_entry__
        __graft_entry__.build()


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + '*.npz')))


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
