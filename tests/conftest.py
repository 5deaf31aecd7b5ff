import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(1, TESTS)   # tests/opcheck.py (shared test infrastructure)

GOLDEN = os.path.join(REPO, 'tests', 'golden')

# Native frames on a host fault (csrc/diag.hip): set before anything loads libpseg_amd.so, inherited by the worker processes
# the tests spawn.  Round 4's two segmentation faults inside the HIP runtime left Python frames only.
os.environ.setdefault('PSEG_SEGV_BACKTRACE', '1')


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # models are built with the reference's pretrained=True call; offline that means random init + a loud warning
    # (tests/test_host_cpu.py asserts the warning itself)
    config.addinivalue_line('filterwarnings', 'ignore:.*RANDOM-INIT.*:RuntimeWarning')
    # the CPU oracle must not oversubscribe a cgroup CPU quota (16 of 256 threads on the GPU box: ~1000x slower)
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
