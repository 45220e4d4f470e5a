"""pytest configuration: registers the `gpu` marker and puts the source roots on sys.path.

`nf-isam_amd/` is a *source root* (like the reference's `src/`): it holds the drop-in modules
`flows`, `slam`, `utils` and the ctypes binding `nfisam_hip`.  `oracle/` is test
infrastructure (CPU restatement) importable as the package `oracle`.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nf-isam_amd")
for p in (SRC, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
