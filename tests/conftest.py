import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _parity_path():
    return os.environ.get('SSC_PARITY_LOG', os.path.join(ROOT, 'gpurun_out', 'parity.jsonl'))


def pytest_sessionstart(session):
    """A session writes its own parity log (tests/test_zz_parity_summary.py reads it at the end)."""
    if os.environ.get('SSC_PARITY_LOG_KEEP') != '1':
        try:
            os.remove(_parity_path())
        except OSError:
            pass


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def parity_log(test, config, max_abs_err, bound, **extra):
    """Append one parity measurement to the JSON-lines file SSC_PARITY_LOG (default gpurun_out/parity.jsonl): every test that
    asserts against the oracle records how far it actually was from it, so the margins are visible outside the GPU box
    (the final tree's file is committed as profiles/r05_parity.jsonl)."""
    import json
    path = _parity_path()
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rec = {'test': test, 'config': config, 'max_abs_err': float(max_abs_err), 'bound': float(bound)}
        rec.update(extra)
        with open(path, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
