import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
PKG = os.path.join(ROOT, "r-nad_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.realpath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "last: long full-size cases, moved to the end of the run")


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=lambda item: 1 if "last" in item.keywords else 0)  # (stable: everything else keeps its order)

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------------------------------------------- achieved parity errors
# Every tolerance-based comparison (np.testing.assert_allclose) of a test is recorded -- largest absolute error, and the largest
# share of its tolerance any element used -- and summarised per test function at the end of the run, so that a regression that
# still passes is visible in the output (and in gpurun_out/parity_errors.json when that directory exists).
_PARITY = {}


@pytest.fixture(autouse=True)
def _record_achieved_errors(request, monkeypatch):
    import numpy as np

    real = np.testing.assert_allclose
    name = request.node.nodeid.split("[")[0]

    def recording(actual, desired, rtol=1e-7, atol=0, *args, **kwargs):
        try:
            x, y = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
            with np.errstate(all="ignore"):
                err = np.abs(x - y)
                used = err / (atol + rtol * np.abs(y))
            ok = np.isfinite(err)
            if ok.any():
                e = _PARITY.setdefault(name, dict(checks=0, max_abs=0.0, max_share=0.0, elements=0))
                e["checks"] += 1
                e["elements"] += int(ok.sum())
                e["max_abs"] = max(e["max_abs"], float(err[ok].max()))
                share = used[ok & np.isfinite(used)]
                if share.size:
                    e["max_share"] = max(e["max_share"], float(share.max()))
        except Exception:  # the record is a by-product: never the reason a test fails
            pass
        return real(actual, desired, rtol, atol, *args, **kwargs)

    monkeypatch.setattr(np.testing, "assert_allclose", recording)
    yield


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _PARITY:
        return
    tr = terminalreporter
    tr.section("achieved parity errors (assert_allclose: largest |error|, largest share of the tolerance used)")
    for name in sorted(_PARITY):
        e = _PARITY[name]
        tr.write_line(f"{name:110s} checks={e['checks']:4d} max_abs_err={e['max_abs']:.3e} tolerance_used={e['max_share']:.3f}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json

        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(_PARITY, f, indent=1, sort_keys=True)
