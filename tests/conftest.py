import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def report_parity(record_property, key: str, **counts):
    """Parity bookkeeping that survives `pytest -q`: the counts go into the junit properties of the test AND into a warning, which
    pytest lists in its warnings summary (so the driver's record of the GPU run shows how far from its allowance a test ran)."""
    import json
    import warnings
    for k, v in counts.items():
        record_property(f"{key}.{k}", v)
    warnings.warn(UserWarning(f"PARITY-COUNT {key} " + json.dumps(counts, sort_keys=True, default=str)))
