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


# SDDP_POISON_LDS=1: the whole GPU suite on NaN-filled LDS.  The hook lives HERE, not in the library (which reads no environment
# variable): every entry point that launches a solve / sweep / pass kernel is wrapped so that sddp_debug_poison_lds runs first.
_LAUNCHING = ("sddp_solve", "sddp_solve_device", "sddp_solve_range_device", "sddp_solve_resident", "sddp_solve_resident_first",
              "sddp_backward", "sddp_forward")


@pytest.fixture(scope="session", autouse=True)
def _poison_lds_before_every_launch():
    if os.environ.get("SDDP_POISON_LDS") != "1" or not _has_gpu():
        yield
        return
    from srbd_horizon_amd import _lib
    lib = _lib.load()
    originals = {}
    for name in _LAUNCHING:
        fn = getattr(lib, name)
        originals[name] = fn

        def wrapped(h, *args, _fn=fn):
            rc = lib.sddp_debug_poison_lds(h)
            if rc != 0:
                return rc
            return _fn(h, *args)
        setattr(lib, name, wrapped)
    yield
    for name, fn in originals.items():
        setattr(lib, name, fn)


def report_parity(record_property, key: str, **counts):
    """Parity bookkeeping that survives `pytest -q`: the counts go into the junit properties of the test AND into a warning, which
    pytest lists in its warnings summary (so the driver's record of the GPU run shows how far from its allowance a test ran)."""
    import json
    import warnings
    for k, v in counts.items():
        record_property(f"{key}.{k}", v)
    warnings.warn(UserWarning(f"PARITY-COUNT {key} " + json.dumps(counts, sort_keys=True, default=str)))
