import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


def pytest_sessionfinish(session, exitstatus):
    """The ledger of operand-scaled tolerances (tests/_cases.py::assert_close_nan): per (test, variable) how many gates passed
    only thanks to their `atol` -- appended to gpurun_out/atol_ledger.jsonl (copied to profiles/ per round).  Child pytest
    sessions (tests/test_gpu_boundary.py runs the parity module again in other staging modes) append their own rows: `session`
    tells them apart."""
    try:
        import json
        import _cases
        if not _cases.ATOL_LEDGER:
            return
        out = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'atol_ledger.jsonl'), 'a') as f:
            for (test, var), (n_atol, n, worst) in sorted(_cases.ATOL_LEDGER.items()):
                f.write(json.dumps({'session': os.getpid(), 'test': test, 'var': var, 'gates_that_needed_atol': n_atol, 'gates_compared': n,
                                    'worst_pure_rel_among_them': worst}) + '\n')
    except Exception:                    # (never turn a finished run red)
        pass
