"""pytest configuration: `gpu` marker + import paths.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol check.
`-m gpu`      : HIP path (through the C-ABI) vs the oracle on seeded inputs.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mm-dti_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))

    return load


@pytest.fixture(autouse=True)
def _oracle_follows_device_precision_mode():
    """The oracle's 16-bit emulation (oracle.mmdti_oracle with bf16=True) rounds at the sites and to the types of the device's
    precision mode: fp16 forward operands (the default) or bf16 everywhere (MMDTI_FWD_FP16=0 / ops.set_forward_fp16(False))."""
    from oracle import mmdti_oracle as O
    from mmdti_hip import ops
    O.set_forward_fp16(ops.FWD_F16)
    yield
    O.set_forward_fp16(ops.FWD_F16)
