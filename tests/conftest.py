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
    # A GPU session on a cold box: the first `import torch` pages in gigabytes of libraries and has taken more than seven minutes
    # (round 6: a run killed for silence in the middle of it, with every test before it green).  One line a minute on the real
    # stderr says the session is alive and where it is.
    expr = getattr(config.option, 'markexpr', '') or ''
    if 'gpu' in expr and 'not gpu' not in expr:
        import threading
        import time

        def beat():
            t0 = time.time()
            while True:
                time.sleep(60)
                line = '[tests] alive after %d s: %s\n' % (time.time() - t0, os.environ.get('PYTEST_CURRENT_TEST', '(between tests)'))
                try:
                    # (pytest's capture holds file descriptors 1 and 2 while a test runs: the line also goes to a file under
                    # gpurun_out/, which the box's watchdog reads as activity)
                    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                    with open(os.path.join(ROOT, 'gpurun_out', 'test_heartbeat.log'), 'a') as f:
                        f.write(line)
                    sys.__stderr__.write(line)
                    sys.__stderr__.flush()
                except Exception:
                    return
        threading.Thread(target=beat, name='cpol-test-heartbeat', daemon=True).start()


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
