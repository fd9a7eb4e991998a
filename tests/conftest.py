import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "qwen-image-finetune_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """Refuse to test a stale libqfx.so (sources newer than the built library)."""
    import __graft_entry__ as g
    if os.path.exists(g.LIB) and g._stale():
        g.build()


@pytest.fixture(autouse=True)
def _reset_attn_policy():
    """Tests switch the attention kernel-selection policy through qfx_attn_tune (process-wide): back to "by shape" after every test."""
    yield
    mod = sys.modules.get("qflux_amd.ops")
    if mod is not None:
        mod.attn_tune("fwd64=auto,dq64=auto,fwd_waves=0")
