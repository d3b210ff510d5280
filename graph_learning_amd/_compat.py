"""Minimal stand-in for sklearn.base.BaseEstimator when scikit-learn is absent.

Only what the estimators here rely on: constructor arguments are stored
verbatim and exposed through get_params / set_params (so copy/clone idioms work).
"""

import inspect


class BaseEstimator:
    @classmethod
    def _param_names(cls):
        sig = inspect.signature(cls.__init__)
        return [p.name for p in sig.parameters.values()
                if p.name != "self" and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]

    def get_params(self, deep=True):
        return {k: getattr(self, k) for k in self._param_names()}

    def set_params(self, **params):
        valid = set(self._param_names())
        for k, v in params.items():
            if k not in valid:
                raise ValueError(f"Invalid parameter {k!r} for estimator {type(self).__name__}")
            setattr(self, k, v)
        return self

    def __repr__(self):
        args = ", ".join(f"{k}={v!r}" for k, v in self.get_params().items())
        return f"{type(self).__name__}({args})"
