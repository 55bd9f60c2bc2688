"""ctypes harness over tests/r_mock/libdsq_rshim_test.so: deseq2_amd/csrc/r_shim.c (the product's .Call binding) running on
a mock R runtime (tests/r_mock/r_mock.c).  Test infrastructure -- see tests/test_r_shim.py."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "r_mock")
SO = os.path.join(HERE, "libdsq_rshim_test.so")
LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, NILSXP = 10, 13, 14, 16, 19, 0
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", HERE, "-s"])
        # the product library first, the way deseq2_amd/_lib.py loads it (torch's bundled HIP runtime before anything else
        # pulls in a second one): the shim library then binds to that same instance
        from deseq2_amd import _lib
        _lib.lib()
        L = C.CDLL(SO)
        P, I, LG = C.c_void_p, C.c_int, C.c_long
        for name, res, args in [("rmock_new", P, [I, I, I, P]), ("rmock_nil", P, []), ("rmock_type", I, [P]),
                                ("rmock_length", LG, [P]), ("rmock_nrow", I, [P]), ("rmock_ncol", I, [P]), ("rmock_data", P, [P]),
                                ("rmock_elt", P, [P, LG]), ("rmock_name", C.c_char_p, [P, LG]), ("rmock_last_error", C.c_char_p, []),
                                ("rmock_protect_depth", I, []), ("rmock_protect_max", I, []), ("rmock_unprotect_underflow", I, []),
                                ("rmock_interrupt_polls", LG, []), ("rmock_live_objects", LG, []), ("rmock_live_transients", LG, []),
                                ("rmock_reset", None, []), ("rmock_load", None, []), ("rmock_n_routines", I, []),
                                ("rmock_routine_name", C.c_char_p, [I]), ("rmock_routine_arity", I, [I]),
                                ("rmock_dynamic_symbols", I, []), ("rmock_is_na_real", I, [C.c_double]),
                                ("rmock_call", P, [C.c_char_p, I, C.POINTER(P)])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.rmock_load()
        _lib = L
    return _lib


class RError(RuntimeError):
    pass


def sexp(v, kind=None):
    """numpy / scalar / None -> SEXP.  2-d arrays become matrices (column-major, as R stores them); kind forces the R
    type: 'int', 'real', 'lgl'."""
    L = lib()
    if v is None:
        return L.rmock_nil()
    a = np.asarray(v)
    if kind is None:
        kind = "lgl" if a.dtype == bool else "int" if a.dtype.kind in "iu" else "real"
    rtype, dt = {"int": (INTSXP, np.int32), "lgl": (LGLSXP, np.int32), "real": (REALSXP, np.float64)}[kind]
    a = np.asfortranarray(a.astype(dt))
    if a.ndim == 2:
        nr, nc = a.shape
    else:
        a = a.reshape(-1)
        nr, nc = a.size, -1
    return L.rmock_new(rtype, nr, nc, a.ctypes.data_as(C.c_void_p))


def value(s):
    """SEXP -> numpy (a copy) / dict for a named list / None for NULL"""
    L = lib()
    if not s:
        raise RError("NULL pointer")
    t = L.rmock_type(s)
    if t == NILSXP:
        return None
    n = L.rmock_length(s)
    if t == VECSXP:
        return {L.rmock_name(s, i).decode(): value(L.rmock_elt(s, i)) for i in range(n)}
    dt = np.float64 if t == REALSXP else np.int32
    buf = (C.c_double if t == REALSXP else C.c_int32) * max(n, 1)
    a = np.frombuffer(buf.from_address(L.rmock_data(s)), dtype=dt, count=n).copy() if n else np.zeros(0, dt)
    if L.rmock_nrow(s) >= 0:
        a = a.reshape((L.rmock_nrow(s), L.rmock_ncol(s)), order="F")
    return a


def rtype(s):
    return lib().rmock_type(s)


def dotCall(name, *args, keep=False):
    """.Call(name, ...): the value (see `value`), or RError carrying R's error message.  Checks after a successful call
    that the shim left the protect stack where it found it and released its transient memory."""
    L = lib()
    arr = (C.c_void_p * max(len(args), 1))(*args)
    out = L.rmock_call(name.encode(), len(args), arr)
    if not out:
        raise RError(L.rmock_last_error().decode())
    assert L.rmock_protect_depth() == 0, "unbalanced PROTECT / UNPROTECT: depth %d after %s" % (L.rmock_protect_depth(), name)
    assert L.rmock_unprotect_underflow() == 0, "UNPROTECT below the stack bottom in %s" % name
    assert L.rmock_live_transients() == 0
    return out if keep else value(out)
