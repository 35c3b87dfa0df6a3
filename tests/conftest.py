import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # session-wide disk cache of ORACLE results (tests/helpers.py::oracle_backward); child interpreters of the forced-kernel
    # tests inherit it through the environment
    if 'MOLGYM_ORACLE_CACHE' not in os.environ:
        import tempfile
        os.environ['MOLGYM_ORACLE_CACHE'] = tempfile.mkdtemp(prefix='molgym_oracle_')


@pytest.fixture(scope='session')
def built_lib():
    import __graft_entry__ as g
    g.build()
    from molgym_amd import _lib
    return _lib.lib()
