"""Parity-margin recorder (test infrastructure).

Every floating-point assert of the `-m gpu` suite reports the worst |error| / bar it saw; at the end of the session
conftest.py writes the table to gpurun_out/parity_margins.json (merged back by gpurun), from which
tools/margins_md.py makes profiles/r04_parity_margins.md.  A ratio <= 1 is a passing assert; "survey" is the same error
measured against SURVEY.md 8d's bar (forward 1e-5 abs + 1e-5 rel, losses rel 1e-4) where the test's own bar differs."""
import os

import numpy as np

_ROWS = {}


def _test_id():
    t = os.environ.get("PYTEST_CURRENT_TEST", "?")
    return t.split(" (")[0]


def record(what, ratio, bar, survey_ratio=None, note=None):
    """Keep the worst ratio per (test, what)."""
    key = (_test_id(), str(what))
    ratio = float(ratio)
    cur = _ROWS.get(key)
    if cur is None or ratio > cur["ratio"]:
        _ROWS[key] = dict(test=key[0], what=key[1], ratio=ratio, bar=str(bar),
                          survey_ratio=None if survey_ratio is None else float(survey_ratio), note=note)
    elif survey_ratio is not None and (cur["survey_ratio"] is None or survey_ratio > cur["survey_ratio"]):
        cur["survey_ratio"] = float(survey_ratio)


def record_close(what, got, ref, atol, rtol, survey=(1e-5, 1e-5)):
    """Worst |got - ref| / (atol + rtol |ref|) of an allclose-style assert, and the same against SURVEY 8d's forward bar."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if got.size == 0:
        return 0.0
    d = np.abs(got - ref)
    r = float((d / (atol + rtol * np.abs(ref) + 1e-300)).max())
    s = None
    if survey is not None:
        s = float((d / (survey[0] + survey[1] * np.abs(ref))).max())
    record(what, r, "%g abs + %g rel" % (atol, rtol), s)
    return r


def rows():
    return [v for _, v in sorted(_ROWS.items())]
