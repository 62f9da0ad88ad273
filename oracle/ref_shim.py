"""In-memory import shim for the READ-ONLY reference at /root/reference (test infrastructure).

Only used in the build container to (i) validate the restatement in oracle/np_oracle.py and
(ii) generate the golden vectors under tests/golden/.  Nothing under /root/reference is
modified or copied; the shim only patches names that newer NumPy/SciPy removed and stubs the
absent `future` package.  It never travels to the GPU box.
"""
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "qinfer"))


def install():
    """Make `import qinfer` resolve to the reference tree. Idempotent."""
    if "qinfer" in sys.modules and getattr(sys.modules["qinfer"], "__file__", "").startswith(REFERENCE_SRC):
        return sys.modules["qinfer"]
    if not available():
        raise ImportError("reference tree not present at %s" % REFERENCE_SRC)
    sys.dont_write_bytecode = True
    ver = types.ModuleType("qinfer.version")
    ver.version = "1.0"
    sys.modules["qinfer.version"] = ver

    fut, futu = types.ModuleType("future"), types.ModuleType("future.utils")

    def with_metaclass(meta, *bases):
        class metaclass(type):
            def __new__(cls, name, this_bases, d):
                return meta(name, bases, d)
        return type.__new__(metaclass, "temporary_class", (), {})

    futu.with_metaclass = with_metaclass
    futu.iteritems = lambda d: iter(d.items())
    fut.utils = futu
    sys.modules.setdefault("future", fut)
    sys.modules.setdefault("future.utils", futu)
    past, pb = types.ModuleType("past"), types.ModuleType("past.builtins")
    pb.basestring = str
    past.builtins = pb
    sys.modules.setdefault("past", past)
    sys.modules.setdefault("past.builtins", pb)

    import scipy.integrate as si
    if not hasattr(si, "cumtrapz"):
        si.cumtrapz = si.cumulative_trapezoid
    import numpy as np
    for n, t in (("float", float), ("int", int), ("bool", bool), ("complex", complex)):
        if n not in np.__dict__:
            setattr(np, n, t)
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid
    if "issctype" not in np.__dict__:      # removed in NumPy 2.0; used by simple_est.py:86-87
        def issctype(rep):
            if not isinstance(rep, (type, np.dtype)):
                return False
            try:
                t = np.dtype(rep).type
            except TypeError:
                return False
            return t is not np.object_ and issubclass(t, np.generic)
        np.issctype = issctype
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import qinfer
    return qinfer
