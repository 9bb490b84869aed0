import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the suite's problems are small: keep them on the fused ELBO path (the product skips it below ~5 GF of data-GP
# contraction per step, where it does not pay); tests/test_fused_elbo.py checks the default line itself
os.environ.setdefault("GPSA_FUSE_MIN_FLOPS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
