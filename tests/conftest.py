import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _reset_options():
    """Options live in a process-wide table (as in the reference); isolate tests."""
    import sparsex_amd as sx
    sx.options_reset()
    sx.lib().spx_log_error_console()
    yield
    sx.options_reset()
