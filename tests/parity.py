"""Measured-error ledger of the GPU parity tests.  Every stage-level comparison against the oracle goes through
``check(name, measured, bound)``: it asserts ``measured <= bound`` and records both; at session end the ledger is
written to ``gpurun_out/parity_measured.json`` (scratch on the GPU box; the copy judged is ``profiles/r02_parity.json``).
Bounds are set to <= 2x the value measured on MI355X (rounded up to 2 significant digits), see that file."""
import json
import os

LEDGER = {}


def rel_l2(a, b) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name: str, measured: float, bound: float, note: str = "") -> float:
    measured = float(measured)
    LEDGER[name] = {"measured": measured, "bound": float(bound), "ratio_bound_over_measured": (bound / measured) if measured > 0 else None}
    if note:
        LEDGER[name]["note"] = note
    assert measured <= bound, f"{name}: measured {measured:.3e} exceeds the stated bound {bound:.3e}"
    return measured


_COUNTS = {}


def auto(measured: float, bound: float, tag: str = "") -> float:
    """check() named after the running test (PYTEST_CURRENT_TEST) plus a per-test counter / tag."""
    cur = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" ")[0]
    cur = cur.replace("tests/", "").replace(".py::", "::")
    k = _COUNTS.get(cur, 0)
    _COUNTS[cur] = k + 1
    return check(f"{cur}#{tag or k}", measured, bound)


def dump() -> None:
    if not LEDGER:
        return
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "parity_measured.json")
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(LEDGER)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
