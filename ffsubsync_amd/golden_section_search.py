"""Golden-section search: the host control loop around FFTAligner used by ``--gss``.

Mirrors ``gss(f, a, b, tol)`` of ffsubsync/golden_section_search.py:15-74: same bracketing
points, same strict ``yc < yd`` branch, and ``f`` is told when an evaluation is the final one
(MaxScoreAligner.fit_gss records only that one, aligners.py:124-125).
"""
import logging
import math

logger = logging.getLogger(__name__)

_INV_PHI = (math.sqrt(5.0) - 1.0) / 2.0  # 1/phi
_INV_PHI_SQ = (3.0 - math.sqrt(5.0)) / 2.0  # 1/phi^2


def _call(f, x, is_last_iter):
    """golden_section_search.py:44-48 -- objectives may or may not accept the is-last flag."""
    try:
        return f(x, is_last_iter)
    except TypeError:
        return f(x)


def gss(f, a, b, tol=1e-4):
    """Return an interval of width <= tol (approximately) containing the minimiser of a unimodal
    ``f`` on [a, b], reusing one evaluation per step."""
    lo, hi = (a, b) if a <= b else (b, a)
    width = hi - lo
    if width <= tol:
        return lo, hi
    steps = int(math.ceil(math.log(tol / width) / math.log(_INV_PHI)))
    logger.info("About to perform %d iterations of golden section search to find the best framerate", steps)
    x_left = lo + _INV_PHI_SQ * width
    x_right = lo + _INV_PHI * width
    y_left = _call(f, x_left, steps == 1)
    y_right = _call(f, x_right, steps == 1)
    for it in range(steps - 1):
        final = it == steps - 2
        width *= _INV_PHI
        if y_left < y_right:
            hi, x_right, y_right = x_right, x_left, y_left
            x_left = lo + _INV_PHI_SQ * width
            y_left = _call(f, x_left, final)
        else:
            lo, x_left, y_left = x_left, x_right, y_right
            x_right = lo + _INV_PHI * width
            # the reference calls f directly on this branch (golden_section_search.py:69)
            y_right = f(x_right, final)
    if y_left < y_right:
        return lo, x_right
    return x_left, hi
