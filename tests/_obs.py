"""Observed-error log for the tolerance gates of the GPU tests: every gated comparison goes through gate(),
which prints `OBS <name> <observed> (tol <tol>)` (visible with pytest -s / -rP) and keeps the per-name maximum
in gpurun_out/observed_errors.txt when that directory exists -- the tolerances in the tests are set to about
twice the maxima observed on MI355X."""
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_MAX = {}


def gate(name, err, tol):
    err = float(err)
    print("OBS %-58s %.3e (tol %.1e)" % (name, err, tol))
    if err > _MAX.get(name, (-1.0, 0))[0]:
        _MAX[name] = (err, tol)
        d = os.path.join(_ROOT, "gpurun_out")
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "observed_errors.txt"), "a") as f:
                    f.write("%s %.6e %.3e\n" % (name, err, tol))
            except OSError:
                pass
    assert err < tol, (name, err, tol)
