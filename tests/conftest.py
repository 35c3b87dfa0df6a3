import os
import sys

import pytest

# The float64 oracle is thousands of tiny ATen ops: an OpenMP team of all 256 cores of the GPU box is pathologically slow on them
# (bench.py::cpu_baseline measured it), and the forced-kernel child interpreters of tests/test_gpu_parity_full.py run side by side --
# five such teams at once turned 254 s into 665 s.  A bounded team, set before torch is imported; children inherit it.
for _v in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, str(max(1, min(16, os.cpu_count() or 1))))

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
