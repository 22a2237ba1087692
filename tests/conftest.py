import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def ctx():
    import ray_tracing_amd as rt
    c = rt.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _reset_shared_context(request):
    """Tests that switch the shared context's visit-count mode reset it themselves on the success path only; a failed
    assertion in one of them must not change what the tests after it measure."""
    yield
    if "ctx" in request.fixturenames:
        import ray_tracing_amd as rt
        request.getfixturevalue("ctx").set_visit_count_mode(rt.api.COUNT_REFERENCE)
