"""Exception classes of the drop-in boundary.

Same names, same hierarchy and same base (``BaseException`` -- so ``except Exception`` does not catch them) as the
reference's nn_fac/utils/errors.py:8-18; the raise sites mirror nn_fac/update_rules/nnls.py:130-135,176-177,
nn_fac/update_rules/mu.py:28-29,79-80, nn_fac/nmf.py:184-185,387-393 and nn_fac/ntf.py:186-191,422-425.
"""


class ArgumentException(BaseException):
    pass


class InvalidRanksException(ArgumentException):
    pass


class CustomNotEngouhFactors(ArgumentException):
    pass


class CustomNotValidFactors(ArgumentException):
    pass


class CustomNotValidCore(ArgumentException):
    pass


class InvalidInitializationType(ArgumentException):
    pass


class InvalidArgumentValue(ArgumentException):
    pass


class OptimException(BaseException):
    pass


class ZeroColumnWhenUnautorized(OptimException):
    pass


class EngineError(RuntimeError):
    """The HIP engine is missing, failed to load, or returned an error status (no CPU fallback exists)."""
