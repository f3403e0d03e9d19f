"""Minimal stand-in for the slice of ``ffsubsync/sklearn_shim.py`` the alignment path touches.

The reference vendors scikit-learn's ``TransformerMixin`` / ``Pipeline`` (sklearn_shim.py:52-80,
89-335, 362).  The aligner only relies on: ``fit_transform`` chaining, ``Pipeline.fit`` /
``.transform`` / ``.fit_transform``, ``named_steps`` and ``__getitem__``.  This module provides
that contract for use when ffsubsync itself is not importable (e.g. on the GPU box); objects
from the real shim and from this one are interchangeable by duck typing.
"""
from typing import Any, List, Tuple


class _TransformerMixin:
    """sklearn_shim.py:52-80: ``fit_transform(X, y=None, **fit_params)``."""

    def fit_transform(self, X: Any, y: Any = None, **fit_params: Any) -> Any:
        if y is None:
            return self.fit(X, **fit_params).transform(X)  # type: ignore[attr-defined]
        return self.fit(X, y, **fit_params).transform(X)  # type: ignore[attr-defined]


class _Pipeline:
    """Sequential ``(name, transformer)`` steps (sklearn_shim.py:89-335, reduced)."""

    def __init__(self, steps: List[Tuple[str, Any]], verbose: bool = False) -> None:
        self.steps = list(steps)
        self.verbose = verbose
        names = [n for n, _ in self.steps]
        if len(set(names)) != len(names):
            raise ValueError("Names provided are not unique: %r" % (names,))
        for _, est in self.steps[:-1]:
            if est is not None and est != "passthrough" and not (
                (hasattr(est, "fit") or hasattr(est, "fit_transform")) and hasattr(est, "transform")
            ):
                raise TypeError("intermediate steps must implement fit and transform: %r" % (est,))

    # -- introspection -----------------------------------------------------------------------
    @property
    def named_steps(self):
        return dict(self.steps)

    @property
    def _final_estimator(self):
        est = self.steps[-1][1]
        return "passthrough" if est is None else est

    def __len__(self) -> int:
        return len(self.steps)

    def __getitem__(self, ind):
        if isinstance(ind, slice):
            if ind.step not in (1, None):
                raise ValueError("Pipeline slicing only supports a step of 1")
            return self.__class__(self.steps[ind])
        try:
            return self.steps[ind][1]
        except TypeError:
            return self.named_steps[ind]

    def _active(self, with_final: bool = True):
        stop = len(self.steps) if with_final else len(self.steps) - 1
        for name, est in self.steps[:stop]:
            if est is not None and est != "passthrough":
                yield name, est

    # -- fitting -----------------------------------------------------------------------------
    def _fit_head(self, X, y=None, **fit_params):
        per_step = {name: {} for name, _ in self.steps}
        for key, val in fit_params.items():
            if "__" not in key:
                raise ValueError("Pipeline.fit does not accept the %s parameter" % key)
            step, param = key.split("__", 1)
            per_step[step][param] = val
        Xt = X
        for name, est in self._active(with_final=False):
            if hasattr(est, "fit_transform"):
                Xt = est.fit_transform(Xt, y, **per_step[name])
            else:
                Xt = est.fit(Xt, y, **per_step[name]).transform(Xt)
        return Xt, per_step[self.steps[-1][0]]

    def fit(self, X, y=None, **fit_params):
        Xt, last_params = self._fit_head(X, y, **fit_params)
        if self._final_estimator != "passthrough":
            self._final_estimator.fit(Xt, y, **last_params)
        return self

    def fit_transform(self, X, y=None, **fit_params):
        Xt, last_params = self._fit_head(X, y, **fit_params)
        last = self._final_estimator
        if last == "passthrough":
            return Xt
        if hasattr(last, "fit_transform"):
            return last.fit_transform(Xt, y, **last_params)
        return last.fit(Xt, y, **last_params).transform(Xt)

    @property
    def transform(self):
        """A *property* returning the chained transform, as in the reference (sklearn_shim.py:294-321)."""
        if self._final_estimator != "passthrough":
            self._final_estimator.transform  # noqa: B018 -- AttributeError if the last step cannot transform
        return self._transform

    def _transform(self, X):
        Xt = X
        for _, est in self._active():
            Xt = est.transform(Xt)
        return Xt


def _make_pipeline(*steps, **kwargs) -> "_Pipeline":
    """sklearn_shim.py:362 -- name steps after their lower-cased class, numbering duplicates."""
    verbose = kwargs.pop("verbose", False)
    if kwargs:
        raise TypeError('Unknown keyword arguments: "%s"' % list(kwargs)[0])
    names = [type(s).__name__.lower() for s in steps]
    counts = {n: names.count(n) for n in names}
    seen = {}
    out = []
    for n, s in zip(names, steps):
        if counts[n] > 1:
            seen[n] = seen.get(n, 0) + 1
            out.append(("%s-%d" % (n, seen[n]), s))
        else:
            out.append((n, s))
    return _Pipeline(out, verbose=verbose)


# When the reference package is importable its own classes are used, so the drop-in aligners are
# instances of ffsubsync.sklearn_shim.TransformerMixin and pipelines are the caller's Pipeline type;
# otherwise (e.g. on a GPU box without ffsubsync) the stand-ins above provide the same contract.
# Only a missing package selects the stand-ins: any other failure of the reference import is the caller's
# broken install and is not papered over.  (Importing ffsubsync runs its own logging.basicConfig -- that is
# what every ffsubsync user gets; nothing here adds to it.)
import logging as _logging

try:
    from ffsubsync.sklearn_shim import Pipeline, TransformerMixin, make_pipeline  # type: ignore  # noqa: F401

    USING_REFERENCE_CLASSES = True
except ImportError:  # ffsubsync (or one of its dependencies) is not installed
    TransformerMixin, Pipeline, make_pipeline = _TransformerMixin, _Pipeline, _make_pipeline
    USING_REFERENCE_CLASSES = False
_logging.getLogger(__name__).debug("Pipeline / TransformerMixin: %s",
                                   "ffsubsync.sklearn_shim" if USING_REFERENCE_CLASSES else "built-in stand-ins")


def reference_module(name: str):
    """``ffsubsync.<name>`` when the reference package is importable, else None (a missing package only)."""
    if not USING_REFERENCE_CLASSES:
        return None
    import importlib

    try:
        return importlib.import_module("ffsubsync." + name)
    except ImportError:
        return None
