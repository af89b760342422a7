"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module, and only as the checker.  All blocks are numpy uint64 arrays, row-major rows x n.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Coo(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64),
                ("i", C.POINTER(C.c_int32)), ("j", C.POINTER(C.c_int32)), ("x", C.POINTER(C.c_uint32))]


class Trace(C.Structure):
    _fields_ = [("iteration", C.c_int), ("npiv", C.c_int),
                ("vtAv", C.POINTER(C.c_uint64)), ("vtAAv", C.POINTER(C.c_uint64)),
                ("winv", C.POINTER(C.c_uint64)), ("d", C.POINTER(C.c_uint64)),
                ("v", C.POINTER(C.c_uint64)), ("tmp", C.POINTER(C.c_uint64)),
                ("Av", C.POINTER(C.c_uint64)), ("p", C.POINTER(C.c_uint64))]


TRACE_FN = C.CFUNCTYPE(None, C.POINTER(Trace), C.c_void_p)
U64P = C.POINTER(C.c_uint64)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE, "liboracle.so"])
        L = C.CDLL(so)
        L.orc_rng_next.restype = C.c_uint64
        L.orc_invmod.restype = C.c_uint64
        L.orc_invmod.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_semi_inverse.restype = C.c_int
        L.orc_block_lanczos.restype = C.c_int
        L.orc_final_check.restype = C.c_int
        L.orc_iteration_omp.restype = C.c_int
        L.orc_iteration_csr_omp.restype = C.c_int
        _LIB = L
    return _LIB


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def ptr(a):
    return a.ctypes.data_as(U64P) if a is not None else None


class Matrix:
    """COO triplets as the reference keeps them (file order, canonical values)."""

    def __init__(self, nrows, ncols, i, j, x):
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.uint32)
        self.nrows, self.ncols, self.nnz = int(nrows), int(ncols), len(self.i)
        self.c = Coo(self.nrows, self.ncols, self.nnz,
                     self.i.ctypes.data_as(C.POINTER(C.c_int32)),
                     self.j.ctypes.data_as(C.POINTER(C.c_int32)),
                     self.x.ctypes.data_as(C.POINTER(C.c_uint32)))

    @staticmethod
    def load(path, prime):
        M = Coo()
        err = C.create_string_buffer(256)
        if lib().orc_mm_load(path.encode(), C.c_uint64(prime), C.byref(M), err, 256):
            raise ValueError(err.value.decode())
        out = Matrix(M.nrows, M.ncols, np.ctypeslib.as_array(M.i, (max(M.nnz, 1),))[:M.nnz].copy(),
                     np.ctypeslib.as_array(M.j, (max(M.nnz, 1),))[:M.nnz].copy(),
                     np.ctypeslib.as_array(M.x, (max(M.nnz, 1),))[:M.nnz].copy())
        lib().orc_coo_free(C.byref(M))
        return out


def rng_draws(count):
    s = (C.c_uint64 * 4)()
    lib().orc_rng_seed(s)
    return [int(lib().orc_rng_next(s)) for _ in range(count)]


def init_v(nrows, n, prime):
    s = (C.c_uint64 * 4)()
    lib().orc_rng_seed(s)
    nxt = lib().orc_rng_next
    return np.array([nxt(s) % prime for _ in range(nrows * n)], dtype=np.uint64)


def spmv(M, x, transpose, n, prime):
    rows_out = M.ncols if transpose else M.nrows
    y = np.zeros(rows_out * n, dtype=np.uint64)
    x = u64(x)
    lib().orc_spmv(ptr(y), C.byref(M.c), ptr(x), C.c_int(int(transpose)), C.c_int(n), C.c_uint64(prime))
    return y


def spmv_omp(M, x, transpose, n, prime, threads=0):
    rows_out = M.ncols if transpose else M.nrows
    y = np.zeros(rows_out * n, dtype=np.uint64)
    x = u64(x)
    lib().orc_spmv_omp(ptr(y), C.byref(M.c), ptr(x), C.c_int(int(transpose)), C.c_int(n),
                       C.c_uint64(prime), C.c_int(threads))
    return y


def block_dot(N, Av, v, n, prime, omp_threads=None):
    a = np.zeros(n * n, dtype=np.uint64)
    b = np.zeros(n * n, dtype=np.uint64)
    Av, v = u64(Av), u64(v)
    if omp_threads is None:
        lib().orc_block_dot(ptr(a), ptr(b), C.c_int64(N), ptr(Av), ptr(v), C.c_int(n), C.c_uint64(prime))
    else:
        lib().orc_block_dot_omp(ptr(a), ptr(b), C.c_int64(N), ptr(Av), ptr(v), C.c_int(n),
                                C.c_uint64(prime), C.c_int(omp_threads))
    return a, b


def invmod(a, prime):
    return int(lib().orc_invmod(a, prime))


def semi_inverse(M, n, prime):
    M = u64(M)
    winv = np.zeros(n * n, dtype=np.uint64)
    d = np.zeros(n, dtype=np.uint64)
    npiv = lib().orc_semi_inverse(ptr(M), ptr(winv), ptr(d), C.c_int(n), C.c_uint64(prime))
    return npiv, winv, d


def orthogonalize(v, pblk, d, vtAv, vtAAv, winv, N, Av, n, prime, omp_threads=None):
    """Returns (v_next, p_next) for the first N rows."""
    v, Av = u64(v), u64(Av)
    pn = u64(pblk).copy()
    tmp = np.zeros(N * n, dtype=np.uint64)
    args = [ptr(v), ptr(tmp), ptr(pn), ptr(u64(d)), ptr(u64(vtAv)), ptr(u64(vtAAv)), ptr(u64(winv)),
            C.c_int64(N), ptr(Av), C.c_int(n), C.c_uint64(prime)]
    if omp_threads is None:
        lib().orc_orthogonalize(*args)
    else:
        lib().orc_orthogonalize_omp(*args, C.c_int(omp_threads))
    return tmp, pn


def block_lanczos(M, n, prime, right=False, stop_after=-1, trace=None, v_init=None, p_init=None, start_iter=0):
    """Returns dict(v, tmp, p, iterations).  trace(rec: dict) is called once per iteration."""
    nrows = M.ncols if right else M.nrows
    ncols = M.nrows if right else M.ncols
    v = np.zeros(nrows * n, dtype=np.uint64)
    t = np.zeros(ncols * n, dtype=np.uint64)
    pb = np.zeros(nrows * n, dtype=np.uint64)

    def _cb(tp, _user):
        r = tp.contents
        rec = dict(iteration=r.iteration, npiv=r.npiv)
        for name, cnt in (("vtAv", n * n), ("vtAAv", n * n), ("winv", n * n), ("d", n),
                          ("v", nrows * n), ("tmp", ncols * n), ("Av", nrows * n), ("p", nrows * n)):
            rec[name] = np.ctypeslib.as_array(getattr(r, name), (cnt,)).copy()
        trace(rec)

    cb = TRACE_FN(_cb) if trace else C.cast(None, TRACE_FN)
    vi = u64(v_init) if v_init is not None else None
    pi = u64(p_init) if p_init is not None else None
    its = lib().orc_block_lanczos(C.byref(M.c), C.c_int(n), C.c_uint64(prime), C.c_int(int(right)),
                                  C.c_int(stop_after), ptr(v), ptr(t), ptr(pb), ptr(vi), ptr(pi),
                                  C.c_int(start_iter), cb, None)
    return dict(v=v, tmp=t, p=pb, iterations=its)


def final_check(nrows, ncols, n, v, vtM):
    return lib().orc_final_check(C.c_int64(nrows), C.c_int64(ncols), C.c_int(n), ptr(u64(v)), ptr(u64(vtM)))


def save_block(path, nrows, n, v):
    if lib().orc_save_block(path.encode(), C.c_int64(nrows), C.c_int(n), ptr(u64(v))):
        raise OSError("cannot write " + path)


def check_kernel(matrix_path, kernel_path, prime, right=False):
    err = C.create_string_buffer(256)
    rc = lib().orc_check_kernel(matrix_path.encode(), kernel_path.encode(), C.c_uint64(prime),
                                C.c_int(int(right)), err, 256)
    return rc, err.value.decode()


class CsrPair:
    """CSR of M and of M^T for the by-rows OpenMP iteration (orc_iteration_csr_omp)."""

    def __init__(self, M):
        L = lib()
        L.orc_csr_build.restype = C.c_void_p
        L.orc_csr_free.argtypes = [C.c_void_p]
        L.orc_csr_free.restype = None
        self.nrows, self.ncols = M.nrows, M.ncols
        self.a = L.orc_csr_build(C.byref(M.c), C.c_int(0))
        self.at = L.orc_csr_build(C.byref(M.c), C.c_int(1))
        if not self.a or not self.at:
            raise MemoryError("orc_csr_build")

    def spmv(self, x, transpose, n, prime, threads=0):
        rows = self.ncols if transpose else self.nrows
        y = np.zeros(rows * n, dtype=np.uint64)
        lib().orc_spmv_csr_omp(ptr(y), C.c_void_p(self.at if transpose else self.a), ptr(u64(x)), C.c_int(n),
                               C.c_uint64(prime), C.c_int(threads))
        return y

    def iteration(self, n, prime, right, v, tmp, Av, pblk, threads=0):
        nrows = self.ncols if right else self.nrows
        return lib().orc_iteration_csr_omp(C.c_void_p(self.a), C.c_void_p(self.at), C.c_int64(nrows), C.c_int(n),
                                           C.c_uint64(prime), C.c_int(int(right)), ptr(v), ptr(tmp), ptr(Av), ptr(pblk),
                                           C.c_int(threads))

    def close(self):
        for h in (self.a, self.at):
            if h:
                lib().orc_csr_free(C.c_void_p(h))
        self.a = self.at = None


class CsrOne:
    """CSR of M (transpose=False) or of M^T alone, for one by-rows product (orc_spmv_csr_omp): y = A x with one output
    row per loop iteration.  What the full-size slab tests compare a rank's product with."""

    def __init__(self, M, transpose=False):
        L = lib()
        L.orc_csr_build.restype = C.c_void_p
        L.orc_csr_free.argtypes = [C.c_void_p]
        L.orc_csr_free.restype = None
        self.rows = M.ncols if transpose else M.nrows
        self.h = L.orc_csr_build(C.byref(M.c), C.c_int(1 if transpose else 0))
        if not self.h:
            raise MemoryError("orc_csr_build")

    def spmv(self, x, n, prime, threads=0):
        y = np.zeros(self.rows * n, dtype=np.uint64)
        lib().orc_spmv_csr_omp(ptr(y), C.c_void_p(self.h), ptr(u64(x)), C.c_int(n), C.c_uint64(prime), C.c_int(threads))
        return y

    def close(self):
        if self.h:
            lib().orc_csr_free(C.c_void_p(self.h))
        self.h = None


def iteration_omp(M, n, prime, right, v, tmp, Av, pblk, threads=0):
    return lib().orc_iteration_omp(C.byref(M.c), C.c_int(n), C.c_uint64(prime), C.c_int(int(right)),
                                   ptr(v), ptr(tmp), ptr(Av), ptr(pblk), C.c_int(threads))
