"""ctypes binding of libblz_hip.so (include/blz.h) for tests and bench.py.

Host-language note: the reference is a C program, so the product's host side is C
(csrc/host/lanczos_modp.c drives the same ABI).  This module is only the thin Python view the
test-suite and the benchmark use; it adds no computation of its own and there is NO fallback:
if the library is missing, or no GPU is visible, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG, "lib", "libblz_hip.so")

V, TMP, AV, P = 0, 1, 2, 3
VTAV, VTAAV, WINV, D = 0, 1, 2, 3
OK, EINVAL, EIO, EFORMAT, ENOMEM, EHIP, ENOGPU, ECOMM = 0, -1, -2, -3, -4, -5, -6, -7

U64P = C.POINTER(C.c_uint64)
_lib = None


class BlzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"blz error {code}: {msg}")
        self.code = code


class Coo(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64),
                ("i", C.POINTER(C.c_int32)), ("j", C.POINTER(C.c_int32)), ("x", C.POINTER(C.c_uint32))]


class Csr(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("nnz", C.c_int64),
                ("row_ptr", C.POINTER(C.c_uint32)), ("col_idx", C.POINTER(C.c_int32)),
                ("val", C.POINTER(C.c_uint32))]


def lib():
    """Load libblz_hip.so.  Raises if it has not been built: there is no Python/CPU substitute."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built (run __graft_entry__.build() or make in {PKG})")
        # the pool's host driver only supports dmabuf IPC; RCCL across processes needs this before HIP initialises
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        L.blz_last_error.restype = C.c_char_p
        L.blz_rng_next.restype = C.c_uint64
        L.blz_rows.restype = C.c_int64
        L.blz_local_rows.restype = C.c_int64
        L.blz_iterations.restype = C.c_int64
        L.blz_local_nnz.restype = C.c_int64
        L.blz_matrix_stream_bytes.restype = C.c_int64
        L.blz_panel_rows.restype = C.c_int64
        L.blz_prepare_key.restype = C.c_uint64
        L.blz_file_hash.restype = C.c_uint64
        L.blz_prepared_free.restype = None
        L.blz_prepared_free.argtypes = [C.c_void_p]
        L.blz_destroy.restype = None
        L.blz_coo_free.restype = None
        L.blz_csr_free.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BlzError(rc, lib().blz_last_error().decode(errors="replace"))


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def ptr(a):
    return a.ctypes.data_as(U64P) if a is not None else None


def device_count():
    return int(lib().blz_device_count())


class Matrix:
    """struct sparsematrix_t of the reference (sequential/lanczos_modp.c:55-62) as numpy arrays."""

    def __init__(self, nrows, ncols, i, j, x):
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.uint32)
        self.nrows, self.ncols, self.nnz = int(nrows), int(ncols), len(self.i)
        self.c = Coo(self.nrows, self.ncols, self.nnz, self.i.ctypes.data_as(C.POINTER(C.c_int32)),
                     self.j.ctypes.data_as(C.POINTER(C.c_int32)), self.x.ctypes.data_as(C.POINTER(C.c_uint32)))

    @staticmethod
    def _take(M):
        n = int(M.nnz)
        out = Matrix(M.nrows, M.ncols, np.ctypeslib.as_array(M.i, (max(n, 1),))[:n].copy(),
                     np.ctypeslib.as_array(M.j, (max(n, 1),))[:n].copy(),
                     np.ctypeslib.as_array(M.x, (max(n, 1),))[:n].copy())
        lib().blz_coo_free(C.byref(M))
        return out

    @staticmethod
    def load(path, prime):
        """sparsematrix_mm_load(), sequential/lanczos_modp.c:199-263."""
        M = Coo()
        check(lib().blz_mm_load(path.encode(), C.c_uint64(prime), C.byref(M)))
        return Matrix._take(M)

    @staticmethod
    def synth(nrows, ncols, nnz, seed, prime, pattern=False):
        M = Coo()
        check(lib().blz_synth_coo(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(nnz), C.c_uint64(seed),
                                  C.c_int(int(pattern)), C.c_uint64(prime), C.byref(M)))
        return Matrix._take(M)

    @staticmethod
    def synth_part(nrows, ncols, nnz, seed, prime, rows=None, cols=None, pattern=False):
        """the entries of synth(...)'s matrix in rows [rows[0], rows[1]) and columns [cols[0], cols[1]), global indices,
        made without the rest of the matrix (blz_synth_coo_part)"""
        r0, r1 = rows if rows is not None else (0, nrows)
        c0, c1 = cols if cols is not None else (0, ncols)
        M = Coo()
        check(lib().blz_synth_coo_part(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(nnz), C.c_uint64(seed), C.c_int(int(pattern)),
                                       C.c_uint64(prime), C.c_int64(r0), C.c_int64(r1), C.c_int64(c0), C.c_int64(c1), C.byref(M)))
        return Matrix._take(M)

    @staticmethod
    def synth_structured(nrows, ncols, nnz, seed, prime, pattern=False, hot_pct=40, band_pct=30, band=4096):
        """Heavy-tailed column degrees + banded supports (blz_synth_structured): the extra, non-headline workload."""
        M = Coo()
        check(lib().blz_synth_structured(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(nnz), C.c_uint64(seed),
                                         C.c_int(int(pattern)), C.c_uint64(prime), C.c_int(hot_pct), C.c_int(band_pct),
                                         C.c_int64(band), C.byref(M)))
        return Matrix._take(M)

    def save(self, path):
        check(lib().blz_mm_save_coo(path.encode(), C.byref(self.c)))

    def csr(self, transpose=False, pattern=True):
        A = Csr()
        check(lib().blz_csr_from_coo(C.byref(self.c), C.c_int(int(transpose)), C.c_int(int(pattern)), C.byref(A)))
        rp = np.ctypeslib.as_array(A.row_ptr, (A.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(A.col_idx, (max(A.nnz, 1),))[:A.nnz].copy()
        va = np.ctypeslib.as_array(A.val, (max(A.nnz, 1),))[:A.nnz].copy() if A.val else None
        bounds = lambda parts: _partition(A, parts)
        res = dict(rows=int(A.rows), cols=int(A.cols), nnz=int(A.nnz), row_ptr=rp, col_idx=ci, val=va)
        res["partition"] = {p: bounds(p) for p in (1, 2, 3, 4, 8)}
        lib().blz_csr_free(C.byref(A))
        return res


def _partition(A, parts):
    b = (C.c_int64 * (parts + 1))()
    check(lib().blz_partition_rows(C.byref(A), C.c_int(parts), b))
    return list(b)


def shard_matrix(M, right, rank, nranks, chunks=1):
    """blz_shard_matrix(): the slabs, bounds and strides rank `rank` of `nranks` works with (host only)."""
    slabs = (Csr * 2)()
    b0 = (C.c_int64 * (nranks + 1))()
    b1 = (C.c_int64 * (nranks + 1))()
    stride = (C.c_int64 * 2)()
    check(lib().blz_shard_matrix(C.byref(M.c), C.c_int(int(right)), C.c_int(rank), C.c_int(nranks), C.c_int(chunks), slabs,
                                 b0, b1, stride))
    out = []
    for A in slabs:
        rp = np.ctypeslib.as_array(A.row_ptr, (A.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(A.col_idx, (max(A.nnz, 1),))[:A.nnz].copy()
        va = np.ctypeslib.as_array(A.val, (max(A.nnz, 1),))[:A.nnz].copy() if A.val else np.ones(A.nnz, np.uint32)
        out.append(dict(rows=int(A.rows), cols=int(A.cols), nnz=int(A.nnz), row_ptr=rp, col_idx=ci, val=va))
        lib().blz_csr_free(C.byref(A))
    return dict(slabs=out, bounds=[list(b0), list(b1)], stride=list(stride), chunks=chunks if nranks > 1 else 1)


class Prepared:
    """blz_prepared: the rank-independent part of a matrix's set-up (renumbering, CSR(M), CSR(M^T), partition), made once
    (prepare / prepare_for), shared by several contexts, saved to and mmapped from a cache file."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def prepare(M, right, nranks, chunks=1, reorder=1, rows_per_line=2, hot_cap=0, min_share=0.25):
        h = C.c_void_p()
        check(lib().blz_prepare(C.byref(M.c), C.c_int(int(right)), C.c_int(nranks), C.c_int(chunks), C.c_int(reorder),
                                C.c_int(rows_per_line), C.c_int64(hot_cap), C.c_double(min_share), C.byref(h)))
        return Prepared(h)

    @staticmethod
    def prepare_rank(row_part, col_part, nrows, ncols, nnz_total, right, rank, nranks, row_bounds, col_bounds, chunks=1):
        """one rank's prepared matrix from its own rows / columns of M alone (blz_prepare_rank)"""
        rb = np.ascontiguousarray(row_bounds, dtype=np.int64)
        cb = np.ascontiguousarray(col_bounds, dtype=np.int64)
        assert len(rb) == nranks + 1 and len(cb) == nranks + 1
        h = C.c_void_p()
        check(lib().blz_prepare_rank(C.byref(row_part.c), C.byref(col_part.c), C.c_int64(nrows), C.c_int64(ncols),
                                     C.c_int64(nnz_total), C.c_int(int(right)), C.c_int(rank), C.c_int(nranks), C.c_int(chunks),
                                     rb.ctypes.data_as(C.POINTER(C.c_int64)), cb.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(h)))
        return Prepared(h)

    @staticmethod
    def prepare_for(ctx, M, right, nranks):
        h = C.c_void_p()
        check(lib().blz_prepare_for(ctx.h, C.byref(M.c), C.c_int(int(right)), C.c_int(nranks), C.byref(h)))
        return Prepared(h)

    @staticmethod
    def load(path, key):
        h = C.c_void_p()
        check(lib().blz_prepared_load(path.encode(), C.c_uint64(key), C.byref(h)))
        return Prepared(h)

    def save(self, path, key):
        check(lib().blz_prepared_save(self.h, path.encode(), C.c_uint64(key)))

    def slab(self, rank, t):
        A = Csr()
        check(lib().blz_prepared_slab(self.h, C.c_int(rank), C.c_int(t), C.byref(A)))
        rp = np.ctypeslib.as_array(A.row_ptr, (A.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(A.col_idx, (max(A.nnz, 1),))[:A.nnz].copy()
        va = np.ctypeslib.as_array(A.val, (max(A.nnz, 1),))[:A.nnz].copy() if A.val else np.ones(A.nnz, np.uint32)
        out = dict(rows=int(A.rows), cols=int(A.cols), nnz=int(A.nnz), row_ptr=rp, col_idx=ci, val=va)
        lib().blz_csr_free(C.byref(A))
        return out

    def layout(self):
        """(right, nranks, chunks, bounds of side 0, bounds of side 1, strides)"""
        r, nr, ch = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().blz_prepared_describe(self.h, C.byref(r), C.byref(nr), C.byref(ch)))
        b0 = (C.c_int64 * (nr.value + 1))()
        b1 = (C.c_int64 * (nr.value + 1))()
        st = (C.c_int64 * 2)()
        check(lib().blz_prepared_layout(self.h, b0, b1, st))
        return bool(r.value), nr.value, ch.value, list(b0), list(b1), list(st)

    def slab_short(self, rank, t):
        """blz_prepared_slab_short(): product t in its short-side form for rank `rank`."""
        A = Csr()
        check(lib().blz_prepared_slab_short(self.h, C.c_int(rank), C.c_int(t), C.byref(A)))
        rp = np.ctypeslib.as_array(A.row_ptr, (A.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(A.col_idx, (max(A.nnz, 1),))[:A.nnz].copy()
        va = np.ctypeslib.as_array(A.val, (max(A.nnz, 1),))[:A.nnz].copy() if A.val else np.ones(A.nnz, np.uint32)
        out = dict(rows=int(A.rows), cols=int(A.cols), nnz=int(A.nnz), row_ptr=rp, col_idx=ci, val=va)
        lib().blz_csr_free(C.byref(A))
        return out

    def close(self):
        if self.h:
            lib().blz_prepared_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def prepare_key(ctx, content_hash, M, right, nranks):
    return int(lib().blz_prepare_key(ctx.h, C.c_uint64(content_hash), C.c_int64(M.nrows), C.c_int64(M.ncols), C.c_int64(M.nnz),
                                     C.c_int(int(right)), C.c_int(nranks)))


def file_hash(path):
    return int(lib().blz_file_hash(path.encode()))


def reorder_hot(M, hot_rows, hot_cols, min_share=0.10):
    """blz_reorder_hot(): (row_perm, col_perm, (rows taken, columns taken), (their shares of the entries))."""
    rp = np.empty(M.nrows, dtype=np.int32)
    cp = np.empty(M.ncols, dtype=np.int32)
    hot = (C.c_int64 * 2)(hot_rows, hot_cols)
    share = (C.c_double * 2)(0.0, 0.0)
    check(lib().blz_reorder_hot(C.byref(M.c), rp.ctypes.data_as(C.POINTER(C.c_int32)), cp.ctypes.data_as(C.POINTER(C.c_int32)),
                                hot, C.c_double(min_share), share))
    return rp, cp, (int(hot[0]), int(hot[1])), (float(share[0]), float(share[1]))


def reorder_auto(M, hot_rows=0, hot_cols=0, min_share=0.25, rows_per_line=2):
    """blz_reorder_auto(): (row_perm, col_perm, hot taken, shares, (lines per entry M*x, M^T*x), order kind)."""
    rp = np.empty(M.nrows, dtype=np.int32)
    cp = np.empty(M.ncols, dtype=np.int32)
    hot = (C.c_int64 * 2)(hot_rows, hot_cols)
    share = (C.c_double * 2)(0.0, 0.0)
    loc = (C.c_double * 2)(1.0, 1.0)
    kind = C.c_int(0)
    check(lib().blz_reorder_auto(C.byref(M.c), rp.ctypes.data_as(C.POINTER(C.c_int32)), cp.ctypes.data_as(C.POINTER(C.c_int32)),
                                 hot, C.c_double(min_share), share, C.c_int(rows_per_line), loc, C.byref(kind)))
    return rp, cp, (int(hot[0]), int(hot[1])), (float(share[0]), float(share[1])), (float(loc[0]), float(loc[1])), int(kind.value)


def reorder(M):
    """blz_reorder(): (row_perm, col_perm), new index of every row / column of M."""
    rp = np.zeros(M.nrows, dtype=np.int32)
    cp = np.zeros(M.ncols, dtype=np.int32)
    check(lib().blz_reorder(C.byref(M.c), rp.ctypes.data_as(C.POINTER(C.c_int32)), cp.ctypes.data_as(C.POINTER(C.c_int32))))
    return rp, cp


def rng_draws(count):
    s = (C.c_uint64 * 4)()
    lib().blz_rng_seed(s)
    return [int(lib().blz_rng_next(s)) for _ in range(count)]


def rng_fill(words, prime):
    v = np.zeros(words, dtype=np.uint64)
    check(lib().blz_rng_fill(ptr(v), C.c_int64(words), C.c_uint64(prime)))
    return v


def save_block(path, nrows, n, v):
    check(lib().blz_save_block(path.encode(), C.c_int64(nrows), C.c_int(n), ptr(u64(v))))


def check_kernel(matrix_path, kernel_path, prime, right=False):
    """blz_check_kernel(): 0 OK, 1 all-zero kernel, 2 product not zero; raises on file/format errors."""
    row, col = C.c_int64(0), C.c_int(0)
    rc = lib().blz_check_kernel(matrix_path.encode(), kernel_path.encode(), C.c_uint64(prime), C.c_int(int(right)),
                                C.byref(row), C.byref(col))
    if rc < 0:
        check(rc)
    return rc


def checkpoint_save(path, prime, n, right, nrows, iterations, v, p):
    check(lib().blz_checkpoint_save(path.encode(), C.c_uint64(prime), C.c_int(n), C.c_int(int(right)),
                                    C.c_int64(nrows), C.c_int64(iterations), ptr(u64(v)), ptr(u64(p))))


def checkpoint_load(path, prime, n, right, nrows):
    v = np.zeros(nrows * n, dtype=np.uint64)
    p = np.zeros(nrows * n, dtype=np.uint64)
    its = C.c_int64(0)
    check(lib().blz_checkpoint_load(path.encode(), C.c_uint64(prime), C.c_int(n), C.c_int(int(right)),
                                    C.c_int64(nrows), C.byref(its), ptr(v), ptr(p)))
    return int(its.value), v, p


class LoopGroup:
    """blz_loop_group: the loopback communicator of several contexts on one device (one thread per context)."""

    def __init__(self, nranks):
        self.h = C.c_void_p()
        check(lib().blz_loop_group_create(C.c_int(nranks), C.byref(self.h)))
        self.nranks = nranks

    def close(self):
        if self.h:
            lib().blz_loop_group_destroy(self.h)
            self.h = None


class Context:
    """One GPU's solver state: the globals `n` and `prime` of the reference plus its four blocks."""

    def __init__(self, prime, n, device=0):
        self.h = C.c_void_p()
        self.prime, self.n = int(prime), int(n)
        check(lib().blz_create(C.byref(self.h), C.c_int(device), C.c_uint64(prime), C.c_int(n)))

    def close(self):
        if self.h:
            lib().blz_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def word_bytes(self):
        return int(lib().blz_word_bytes(self.h))

    def set_matrix(self, M, right=False, rank=0, nranks=1):
        check(lib().blz_set_matrix(self.h, C.byref(M.c), C.c_int(int(right)), C.c_int(rank), C.c_int(nranks)))
        self.right = bool(right)

    def set_matrix_prepared(self, P, rank=0):
        check(lib().blz_set_matrix_prepared(self.h, P.h, C.c_int(rank)))
        r = C.c_int(0)
        check(lib().blz_prepared_describe(P.h, C.byref(r), None, None))
        self.right = bool(r.value)

    def rows(self, block):
        return int(lib().blz_rows(self.h, C.c_int(block)))

    def local_rows(self, block):
        first = C.c_int64(0)
        cnt = int(lib().blz_local_rows(self.h, C.c_int(block), C.byref(first)))
        return int(first.value), cnt

    def local_nnz(self, transpose):
        return int(lib().blz_local_nnz(self.h, C.c_int(int(transpose))))

    def matrix_stream_bytes(self, transpose):
        return int(lib().blz_matrix_stream_bytes(self.h, C.c_int(int(transpose))))

    def locality(self):
        """((lines per gathered entry of M*x, of M^T*x), order kind) as found by the renumbering (blz_locality)."""
        loc = (C.c_double * 2)(1.0, 1.0)
        kind = C.c_int(0)
        check(lib().blz_locality(self.h, loc, C.byref(kind)))
        return (float(loc[0]), float(loc[1])), int(kind.value)

    def comm_init_loopback(self, group, rank):
        check(lib().blz_comm_init_loopback(self.h, group.h, C.c_int(rank)))

    def comm_info(self):
        """(ranks, rank) as the RCCL communicator reports them; (-1, -1) without one"""
        a, b = C.c_int(-1), C.c_int(-1)
        check(lib().blz_comm_info(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def exchange_pieces_for(self, mrows, mcols, nnz, nranks):
        return int(lib().blz_exchange_pieces_for(self.h, C.c_int64(mrows), C.c_int64(mcols), C.c_int64(nnz), C.c_int(nranks)))

    def exchange_pieces(self, transpose):
        return int(lib().blz_exchange_pieces(self.h, C.c_int(1 if transpose else 0)))

    def short_side(self, transpose):
        return bool(lib().blz_short_side(self.h, C.c_int(int(transpose))) == 1)

    def get_partial(self, transpose):
        rows = self.rows(TMP) if bool(transpose) == (not self.right) else self.rows(V)
        out = np.zeros(rows * self.n, dtype=np.uint64)
        check(lib().blz_get_partial(self.h, C.c_int(int(transpose)), ptr(out)))
        return out

    def panel_rows(self, transpose):
        """(block rows of the operand kept in LDS, share of the entries they serve) for M*x (False) / M^T*x (True)."""
        share = C.c_double(0.0)
        rows = int(lib().blz_panel_rows(self.h, C.c_int(int(transpose)), C.byref(share)))
        return rows, float(share.value)

    def owner_of_row(self, block, row):
        return int(lib().blz_owner_of_row(self.h, C.c_int(block), C.c_int64(row)))

    def init_v(self):
        check(lib().blz_init_v(self.h))

    def set_block(self, block, host):
        host = u64(host)
        assert host.size == self.rows(block) * self.n, (host.size, self.rows(block), self.n)
        check(lib().blz_set_block(self.h, C.c_int(block), ptr(host)))

    def get_block(self, block):
        out = np.zeros(self.rows(block) * self.n, dtype=np.uint64)
        check(lib().blz_get_block(self.h, C.c_int(block), ptr(out)))
        return out

    def set_small(self, which, host):
        check(lib().blz_set_small(self.h, C.c_int(which), ptr(u64(host))))

    def get_small(self, which):
        out = np.zeros(self.n if which == D else self.n * self.n, dtype=np.uint64)
        check(lib().blz_get_small(self.h, C.c_int(which), ptr(out)))
        return out

    def spmv(self, transpose, src, dst):
        check(lib().blz_spmv(self.h, C.c_int(int(transpose)), C.c_int(src), C.c_int(dst)))

    def block_dot(self):
        a = np.zeros(self.n * self.n, dtype=np.uint64)
        b = np.zeros(self.n * self.n, dtype=np.uint64)
        check(lib().blz_block_dot(self.h, ptr(a), ptr(b)))
        return a, b

    def semi_inverse(self):
        npiv = C.c_int(0)
        winv = np.zeros(self.n * self.n, dtype=np.uint64)
        d = np.zeros(self.n, dtype=np.uint64)
        check(lib().blz_semi_inverse(self.h, C.byref(npiv), ptr(winv), ptr(d)))
        return int(npiv.value), winv, d

    def orthogonalize(self):
        check(lib().blz_orthogonalize(self.h))

    def iterate(self, max_iters):
        done, stopped, ms = C.c_int(0), C.c_int(0), C.c_float(0)
        check(lib().blz_iterate(self.h, C.c_int(max_iters), C.byref(done), C.byref(stopped), C.byref(ms)))
        return int(done.value), bool(stopped.value), float(ms.value)

    @property
    def iterations(self):
        return int(lib().blz_iterations(self.h))

    def set_iterations(self, its):
        check(lib().blz_set_iterations(self.h, C.c_int64(its)))

    def final_check(self):
        a, b = C.c_int(0), C.c_int(0)
        check(lib().blz_final_check(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def time_kernel(self, which, reps):
        ms = C.c_float(0)
        check(lib().blz_time_kernel(self.h, C.c_int(which), C.c_int(reps), C.byref(ms)))
        return float(ms.value)

    PROFILE_CLASSES = ("spmv1", "spmv2", "block_dot", "semi_inverse", "orthogonalize", "allgather_v",
                       "allgather_tmp", "allreduce", "reduce_scatter")

    def profile(self, enable):
        check(lib().blz_profile(self.h, C.c_int(int(enable))))

    def profile_read(self):
        ms = (C.c_double * len(self.PROFILE_CLASSES))()
        cnt = (C.c_int64 * len(self.PROFILE_CLASSES))()
        check(lib().blz_profile_read(self.h, ms, cnt))
        return {k: dict(ms_total=float(ms[i]), launches=int(cnt[i])) for i, k in enumerate(self.PROFILE_CLASSES)}

    def set_exchange_mode(self, external):
        check(lib().blz_set_exchange_mode(self.h, C.c_int(int(external))))

    def snapshot_begin(self):
        check(lib().blz_snapshot_begin(self.h))

    def snapshot_wait(self, v=None, p=None):
        """(v, p, iterations) of the snapshot; this rank's rows are written into v / p (allocated zero if not given)."""
        n = self.n
        if v is None:
            v = np.zeros(self.rows(V) * n, dtype=np.uint64)
        if p is None:
            p = np.zeros(self.rows(V) * n, dtype=np.uint64)
        its = C.c_int64(0)
        check(lib().blz_snapshot_wait(self.h, ptr(v), ptr(p), C.byref(its)))
        return v, p, int(its.value)

    def sync(self):
        check(lib().blz_sync(self.h))

    def comm_init(self, uid, rank, nranks):
        buf = (C.c_char * len(uid)).from_buffer_copy(uid)
        check(lib().blz_comm_init(self.h, buf, C.c_size_t(len(uid)), C.c_int(rank), C.c_int(nranks)))


def comm_unique_id():
    buf = (C.c_char * 128)()
    check(lib().blz_comm_unique_id(buf, C.c_size_t(128)))
    return bytes(buf)


def solve(M, prime, n, right=False, stop_after=-1, batch=16, device=0):
    """block_lanczos(), sequential/lanczos_modp.c:585-669, on one GPU.  Returns dict(v, tmp, iterations)."""
    with Context(prime, n, device) as ctx:
        ctx.set_matrix(M, right)
        ctx.init_v()
        while True:
            todo = batch
            if stop_after > 0:
                todo = min(batch, stop_after - ctx.iterations)
                if todo <= 0:
                    break
            _, stopped, _ = ctx.iterate(todo)
            if stopped:
                break
        return dict(v=ctx.get_block(V), tmp=ctx.get_block(TMP), p=ctx.get_block(P), iterations=ctx.iterations,
                    final_check=ctx.final_check())
