/*
 * blz.h -- C ABI of libblz_hip.so: the block-Lanczos-mod-p hot path on MI355X (gfx950).
 *
 * The reference (T-amairi/block-lanczos-algorithm-parallelization) has no FFI or plugin seam:
 * its kernels are plain C functions inside sequential/lanczos_modp.c, parameterised by two
 * globals (`long n`, `u64 prime`, :39-40) and caller-owned flat arrays.  This header exports one
 * entry point per reference function on the hot path, with the globals folded into an opaque
 * per-GPU context.  File:line citations are relative to /root/reference/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++ or torch types cross the boundary.
 *   - every function returns 0 (BLZ_OK) or a negative BLZ_E* code; blz_last_error() holds the
 *     text the reference would have passed to errx() (thread-local, valid until the next call).
 *   - host blocks are row-major rows x n arrays of uint64_t canonical residues in [0,p), exactly
 *     the reference's `v[i*n + l]` layout (sequential/lanczos_modp.c:282-284) with the word
 *     widened from u32 to u64 (the reference's cap p <= 2^30-35, :189-193, is lifted to p < 2^62).
 *   - a context owns all device memory and one HIP stream; the caller owns every host pointer.
 *     One context per GPU, driven by one host thread at a time.
 *   - functions in the "host-side" section never touch the GPU and work on any machine.
 */
#ifndef BLZ_H
#define BLZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLZ_OK          0
#define BLZ_EINVAL     -1	/* bad argument / unsupported size */
#define BLZ_EIO        -2	/* file could not be opened / parsed */
#define BLZ_EFORMAT    -3	/* MatrixMarket type not "coordinate integer general" */
#define BLZ_ENOMEM     -4
#define BLZ_EHIP       -5	/* a HIP runtime call failed */
#define BLZ_ENOGPU     -6	/* no usable gfx950 device: there is NO CPU fallback */
#define BLZ_ECOMM      -7	/* RCCL failure */

#define BLZ_MAX_N      64	/* block width limit (one wavefront holds a block row) */

/* block selectors: the four N x n blocks of block_lanczos(), sequential/lanczos_modp.c:602-605 */
enum { BLZ_V = 0, BLZ_TMP = 1, BLZ_AV = 2, BLZ_P = 3 };
/* small n x n (or n) operands of one iteration, sequential/lanczos_modp.c:638-643 */
enum { BLZ_VTAV = 0, BLZ_VTAAV = 1, BLZ_WINV = 2, BLZ_D = 3 };

typedef struct blz_ctx blz_ctx;

const char *blz_last_error(void);
int blz_version(void);

/* ------------------------------------------------------------------ host-side (no GPU) */

/* struct sparsematrix_t, sequential/lanczos_modp.c:55-62: 0-based COO triplets in file order;
 * x already canonicalised as the loader does (below). */
typedef struct {
	int64_t nrows, ncols, nnz;
	int32_t *i, *j;
	uint32_t *x;
} blz_coo;

/* CSR of M (or of M^T): what the HIP SpMV streams.  row_ptr has rows+1 entries. */
typedef struct {
	int64_t rows, cols, nnz;
	uint32_t *row_ptr;
	int32_t *col_idx;
	uint32_t *val;		/* NULL = every entry is 1 (pattern path) */
} blz_csr;

/* sparsematrix_mm_load(), sequential/lanczos_modp.c:199-263: accepts only
 * "matrix coordinate integer general" (:216-221, BLZ_EFORMAT otherwise); entries are parsed as
 * C ints, stored into a u32 and reduced `% prime` (:238-243) -- negative entries therefore wrap
 * to 2^32-|x| before the reduction, exactly as in the reference and in checker_modp.c:170-175.
 * Unlike the reference, out-of-range indices are an error (BLZ_EIO) instead of undefined behaviour. */
int blz_mm_load(const char *path, uint64_t prime, blz_coo *out);
void blz_coo_free(blz_coo *M);

/* Write triplets as a MatrixMarket "coordinate integer general" file (1-based, values as stored).  Used to hand a
 * synthetic matrix to programs that only read files (the reference binaries). */
int blz_mm_save_coo(const char *path, const blz_coo *M);

/* Seeded synthetic stand-in for a SuiteSparse matrix that is not on the box:
 * row r gets floor(nnz/R) + (r < nnz mod R) distinct uniform columns; values from
 * {1,1,1,2,3,-1,-2} (pattern=0, canonicalised like the loader does) or all 1 (pattern=1). */
int blz_synth_coo(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern,
		  uint64_t prime, blz_coo *out);
/* The entries of that same matrix in rows [r0, r1) and columns [c0, c1), global indices, without making the rest
 * (every row is seeded by itself): a rank's own rows (c0 = 0, c1 = ncols) or own columns (r0 = 0, r1 = nrows) of a
 * matrix too large to hold whole -- config 5's 2e9 entries.  What the reference's MPI loader does with
 * the file (mpi/lanczos_modp.c:1841-1845). */
int blz_synth_coo_part(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern, uint64_t prime,
		       int64_t r0, int64_t r1, int64_t c0, int64_t c1, blz_coo *out);

/* A synthetic matrix WITH structure (never the headline workload): hot_pct % of a row's entries are drawn with
 * probability ~ 1/(c + 16) (heavy-tailed column degrees, dense columns first, as in a sieve relation matrix),
 * band_pct % uniformly from a band of `band` columns centred on r * C / R (correlated supports of neighbouring
 * rows), the rest uniformly.  Used to measure what the uniform stand-ins cannot show: the LDS-resident panel of
 * dense block rows and the per-XCD row ranges of the SpMV. */
int blz_synth_structured(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern, uint64_t prime,
			 int hot_pct, int band_pct, int64_t band, blz_coo *out);

/* COO -> CSR of M (transpose=0) or of M^T (transpose=1); duplicates are kept (they add,
 * as in the reference's scatter loop :277-286).  pattern!=0 drops the value array when all
 * values are 1. */
int blz_csr_from_coo(const blz_coo *M, int transpose, int pattern, blz_csr *out);
void blz_csr_free(blz_csr *A);

/* nnz-balanced contiguous row partition: bounds[0]=0 <= ... <= bounds[parts]=rows. */
int blz_partition_rows(const blz_csr *A, int parts, int64_t *bounds);

/* Locality reordering of both index spaces (host only).  Every step of the iteration is invariant under
 * permutations of the rows of v and of tmp (sums mod p are exact and order-free), so the solver is free to
 * renumber them: rows of M are sorted by their smallest column index, then columns by their smallest (new) row
 * index.  The entries that define the order then hit the same or the next block row as their neighbours
 * (shared L2 lines, two 64-byte rows per 128-byte fabric request) in BOTH products.  Measured on MI355X:
 * -5 % per iteration on the GL7d19-shape matrix, -9 % on the relat9 shape.
 * row_perm[r] / col_perm[c] = new index of row r / column c of M (arrays of nrows / ncols int32). */
int blz_reorder(const blz_coo *M, int32_t *row_perm, int32_t *col_perm);

/* blz_reorder with the densest rows / columns numbered first.  hot[0] (rows) and hot[1] (columns): in = the most a
 * panel can hold, out = how many were taken (0 when they hold less than min_share of the entries); share[] = the
 * fraction of the entries they hold.  The SpMV keeps the first hot[.] block rows of its operand in LDS. */
int blz_reorder_hot(const blz_coo *M, int32_t *row_perm, int32_t *col_perm, int64_t hot[2], double min_share,
		    double share[2]);
/* The renumbering the solver uses: blz_reorder_hot's hot rows / columns in front, and behind them the best of three
 * orders -- rows by smallest column (blz_reorder), the file's own order, rows by the mean of their columns -- judged on a
 * sample of windows of 4096 consecutive rows of each product by the number of distinct 128-byte lines of the operand
 * they touch (rows_per_line block rows share a line).  locality[t] = lines per gathered entry of product t (0: M * x,
 * 1: M^T * x) under the chosen order; 1.0 means no reuse to be had.  *kind (may be NULL): 0 smallest, 1 file order, 3 iterated barycentre sweeps,
 * 2 mean. */
int blz_reorder_auto(const blz_coo *M, int32_t *row_perm, int32_t *col_perm, int64_t hot[2], double min_share,
		     double share[2], int rows_per_line, double locality[2], int *kind);
/* entries of every row in ascending column order */
void blz_csr_sort_rows(blz_csr *A);

/* ---- the prepared matrix: everything blz_set_matrix needs that does not depend on the rank, made ONCE ----
 * (renumbering, CSR(M) and CSR(M^T) in the solver's numbering, nnz-balanced row partition of both sides).  The CLI's
 * main thread prepares for its G contexts; rank 0 of a multi-process job prepares, saves, and the other ranks load
 * (mmap: one copy of the pages per node).  Saved next to the matrix it is the binary cache of SURVEY 8(f)1: a second
 * run skips the renumbering and the CSR builds.  sequential/lanczos_modp.c:199-263 is what it stands in front of.
 *   reorder        0 keep the file's numbering, 1 scored choice (blz_reorder_auto), 2 round 1's order
 *   chunks         pieces per exchange (1 for one rank)
 *   rows_per_line  block rows of the context per 128-byte line (128 / (width in HBM * word bytes), at least 1)
 *   hot_cap        block rows the LDS panel can hold (0: none); only used with one rank and one piece
 * `key` ties a cache file to its inputs (the CLI uses blz_file_hash of the matrix ^ prime, width, ranks, ...): a
 * load with another key fails with BLZ_EFORMAT and the caller prepares afresh. */
typedef struct blz_prepared blz_prepared;
int blz_prepare(const blz_coo *M, int right, int nranks, int chunks, int reorder, int rows_per_line, int64_t hot_cap,
		double min_share, blz_prepared **out);
/* One rank's prepared matrix from that rank's share alone: row_part / col_part = the entries of M in its rows / in its
 * columns (global indices), the caller's partition (bounds: nranks + 1 ascending values from 0 to the dimension), the
 * caller's numbering (no renumbering).  Only blz_set_matrix_prepared(ctx, P, rank) with that rank accepts it; it cannot
 * be saved.  No process holds the whole matrix -- the reference's MPI variant works that way too
 * (mpi/lanczos_modp.c:1841-1900). */
int blz_prepare_rank(const blz_coo *row_part, const blz_coo *col_part, int64_t nrows, int64_t ncols, int64_t nnz_total,
		     int right, int rank, int nranks, int chunks, const int64_t *row_bounds, const int64_t *col_bounds,
		     blz_prepared **out);
int blz_prepared_save(const blz_prepared *P, const char *path, uint64_t key);
int blz_prepared_load(const char *path, uint64_t key, blz_prepared **out);
void blz_prepared_free(blz_prepared *P);
/* the row partition: bounds0 / bounds1 [nranks + 1] of side 0 (rows of v) / side 1 (rows of tmp), rows of a padded slab per side */
int blz_prepared_layout(const blz_prepared *P, int64_t *bounds0, int64_t *bounds1, int64_t stride[2]);
/* what P was prepared for (any pointer may be NULL) */
int blz_prepared_describe(const blz_prepared *P, int *right, int *nranks, int *chunks);
/* rank `rank`'s rows of M (t = 0) or of M^T (t = 1) as a CSR of its own, columns rewritten to positions in the gathered
 * operand (what blz_shard_matrix returns in slabs[t]) */
int blz_prepared_slab(const blz_prepared *P, int rank, int t, blz_csr *slab);
/* The matrix of product t in its SHORT-SIDE form (tall / wide matrices on several ranks): the transpose of this rank's
 * rows of the other orientation.  Rows = the padded rank-major numbering of the output side (nranks x stride), columns =
 * row numbers inside this rank's own slab of the operand: the rank multiplies it by its OWN slab (nothing is gathered)
 * and a reduce-scatter of the full-length partial products replaces the all-gather of the long block
 * (mpi/lanczos_modp.c:1108-1124 reduces partial products too, through rank 0). */
int blz_prepared_slab_short(const blz_prepared *P, int rank, int t, blz_csr *out);
/* 64-bit content hash of a file (0 on error) */
uint64_t blz_file_hash(const char *path);

/* What rank `rank` of `nranks` keeps of M for the solve (right=0: x*M=0, right=1: M*x=0).
 * "Side 0" is the row space of v/Av/p, "side 1" that of tmp (sequential/lanczos_modp.c:592-593).
 *   bounds0/bounds1 [nranks+1]  nnz-balanced row partition of each side
 *   stride[2]                   rows of a (padded) slab of each side, a multiple of `chunks`
 *   slabs[0] = this rank's rows of M, slabs[1] = its rows of M^T, column indices already rewritten to positions
 *              in the gathered operand of the opposite side:  with piece = stride/chunks, row q of rank g's slab
 *              sits at  (q / piece) * (nranks * piece) + g * piece + (q % piece)  -- piece-major, so that
 *              all-gather number k (piece k of every slab) lands contiguously and the product can start on it
 *              while piece k+1 is in flight.  chunks = 1 gives the plain rank-major padded layout
 *              g * stride + q; one rank gives the identity.
 * This is the whole multi-GPU data layout; blz_set_matrix uploads exactly these slabs. */
int blz_shard_matrix(const blz_coo *M, int right, int rank, int nranks, int chunks, blz_csr slabs[2],
		     int64_t *bounds0, int64_t *bounds1, int64_t stride[2]);

/* rng_state/random64(), sequential/lanczos_modp.c:67-87, and the initialisation
 * `v[i] = random64() % prime` in row-major order (:624-625). */
void blz_rng_seed(uint64_t state[4]);
uint64_t blz_rng_next(uint64_t state[4]);
int blz_rng_fill(uint64_t *v, int64_t words, uint64_t prime);

/* save_vector_block(), sequential/lanczos_modp.c:673-686: MatrixMarket "array integer general",
 * the same fixed comment line, column-major "%d" lines (words >= 2^32, which the reference
 * cannot produce, are written as unsigned decimals). */
int blz_save_block(const char *path, int64_t nrows, int n, const uint64_t *v);

/* checker_modp.c:81-204 widened to u64 words (the reference's checker parses kernel entries with "%d" into a
 * u32, so it cannot verify p > 2^31-1): loads the kernel block (MatrixMarket "array integer general", column-major,
 * rows must equal the matrix's rows -- or columns with right!=0, :99-124), rejects entries >= prime (:150),
 * computes y = x^T M (or M x) mod p and returns
 *   0 = OK, 1 = kernel vectors are all zero (:155-161), 2 = y != 0 (:199-203); bad_row and bad_col locate the first
 * non-zero word.  Negative returns are BLZ_E* errors (file, format, dimension mismatch). */
int blz_check_kernel(const char *matrix_path, const char *kernel_path, uint64_t prime, int right,
		     int64_t *bad_row, int *bad_col);

/* Checkpoints (openMP/lanczos_modp.c:571-676, :933-940, :1013-1022).  blz_checkpoint_save writes
 * one binary file atomically (tmp + rename): v, p, iteration count, prime, n, shape.
 * The *_ref_text pair reads/writes the reference's five text files (v.txt tmp.txt Av.txt p.txt
 * verbosity.txt) in `dir` so that runs can be handed over in either direction (p < 2^32 only). */
int blz_checkpoint_save(const char *path, uint64_t prime, int n, int right, int64_t nrows,
			int64_t iterations, const uint64_t *v, const uint64_t *p);
int blz_checkpoint_load(const char *path, uint64_t prime, int n, int right, int64_t nrows,
			int64_t *iterations, uint64_t *v, uint64_t *p);
int blz_checkpoint_save_ref_text(const char *dir, int n, int64_t nrows, int64_t ncols,
				 int64_t iterations, double start, double now, const uint64_t *v,
				 const uint64_t *tmp, const uint64_t *Av, const uint64_t *p);
int blz_checkpoint_load_ref_text(const char *dir, int n, int64_t nrows, int64_t ncols,
				 int64_t *iterations, uint64_t *v, uint64_t *p);

/* ------------------------------------------------------------------------- device side */

int blz_device_count(void);	/* 0 when no GPU is visible; never fails */

/* Replaces the globals `n` and `prime`.  Fails with BLZ_ENOGPU when there is no device:
 * the product has no CPU path.  2 <= prime < 2^62, 1 <= n <= BLZ_MAX_N. */
int blz_create(blz_ctx **out, int device, uint64_t prime, int n);
void blz_destroy(blz_ctx *ctx);
int blz_word_bytes(const blz_ctx *ctx);	/* 4 if prime < 2^32 else 8: width of a residue in HBM */

/* Upload M for the solve x*M=0 (right=0) or M*x=0 (right=1), as block_lanczos(M, n, transpose)
 * receives it (sequential/lanczos_modp.c:585).  Builds CSR(M) and CSR(M^T) on the host, keeps
 * this rank's nnz-balanced row slabs, allocates the four blocks and zeroes them (:617-622).
 * rank/nranks = 0/1 for a single GPU. */
int blz_set_matrix(blz_ctx *ctx, const blz_coo *M, int right, int rank, int nranks);

/* The same with the rank-independent work done once and shared (blz_prepare / blz_prepared_load above):
 *   blz_prepare_for   prepares M with the parameters THIS context wants (pieces per exchange from the slab sizes and
 *                     BLZ_AG_CHUNKS, renumbering, panel capacity from its block width and word size)
 *   blz_prepare_key   the cache key for those parameters and a content hash of the matrix
 *   blz_set_matrix_prepared  cuts rank `rank`'s slabs out of P and uploads them (P stays the caller's) */
int blz_prepare_for(const blz_ctx *c, const blz_coo *M, int right, int nranks, blz_prepared **out);
uint64_t blz_prepare_key(const blz_ctx *c, uint64_t content_hash, int64_t mrows, int64_t mcols, int64_t nnz, int right,
			 int nranks);
int blz_set_matrix_prepared(blz_ctx *c, const blz_prepared *P, int rank);
/* The solver renumbers rows internally (blz_reorder; BLZ_NO_REORDER=1 disables it).  Nothing of it is visible
 * through this ABI: blz_set_block / blz_get_block / blz_init_v / checkpoints all speak the ORIGINAL row numbering,
 * and results are bit-identical either way.  With nranks > 1 a rank's slab is a set of original rows that need
 * not be contiguous; blz_owner_of_row tells which rank holds a given original row of a block. */
int blz_owner_of_row(const blz_ctx *ctx, int block, int64_t row);

/* Block rows of the operand of product `transpose` (0: M * x, 1: M^T * x) that the SpMV keeps in LDS: the densest
 * columns (rows for the transpose) of a heavy-tailed matrix, numbered first by the solver's internal renumbering.
 * 0 for matrices without such structure, with several ranks, or with BLZ_NO_PANEL=1.  *share (may be NULL) = the
 * fraction of the entries those block rows serve. */
int64_t blz_panel_rows(const blz_ctx *c, int transpose, double *share);

/* What the renumbering found: locality[t] = distinct 128-byte lines of the operand per gathered entry in windows of 4096
 * consecutive rows of product t (0: M * x, 1: M^T * x) -- 1.0 on a matrix without structure, lower when neighbouring
 * rows share columns (the SpMV then walks per-XCD row ranges); *order_kind (may be NULL): 0 rows by smallest column,
 * 1 the file's order, 2 rows by the mean of their columns. */
int blz_locality(const blz_ctx *c, double locality[2], int *order_kind);

/* Short-side exchange: 1 when product `transpose` (0: M * x, 1: M^T * x) runs in its short-side form on this context
 * -- several ranks, 64-bit words, operand side at least 8 times longer than the output side (BLZ_SHORT_SIDE=0/1
 * overrides): the rank multiplies the transpose of its own rows of the other orientation by its own slab and a
 * reduce-scatter of the partial products replaces the all-gather of the long block.  In external-exchange mode
 * blz_spmv leaves the partial product on the device and blz_get_partial returns it (rows of the output side x n words,
 * original numbering, unreduced sums) for the caller to sum over the ranks. */
int blz_short_side(const blz_ctx *c, int transpose);
int blz_get_partial(blz_ctx *c, int transpose, uint64_t *host);

int64_t blz_rows(const blz_ctx *ctx, int block);	/* global row count of a block (N or C) */
/* size of this rank's slab of a block; *first = its first row in the SOLVER's numbering (see blz_owner_of_row) */
int64_t blz_local_rows(const blz_ctx *ctx, int block, int64_t *first);
int64_t blz_local_nnz(const blz_ctx *ctx, int transpose);		/* entries of this rank's slab of M (0) or M^T (1) */
int64_t blz_matrix_stream_bytes(const blz_ctx *ctx, int transpose);	/* bytes of that slab as resident in HBM
									 * (row_ptr + packed or plain col_idx/val) */

/* v <- random64() % p for this rank's rows, everything else 0 (:617-625). */
int blz_init_v(blz_ctx *ctx);

/* Copy a whole block (global rows x n, original row numbering) host->device / device->host.  With
 * nranks > 1 set_block fills the whole padded layout (so it doubles as an emulated all-gather); get_block
 * writes only the rows this rank owns and leaves the rest of `host` untouched. */
int blz_set_block(blz_ctx *ctx, int block, const uint64_t *host);
int blz_get_block(blz_ctx *ctx, int block, uint64_t *host);
int blz_set_small(blz_ctx *ctx, int which, const uint64_t *host);
int blz_get_small(blz_ctx *ctx, int which, uint64_t *host);

/* sparse_matrix_vector_product(y, M, x, transpose), sequential/lanczos_modp.c:266-287:
 * dst = M*src (transpose=0) or M^T*src (transpose=1), all n columns, canonical residues. */
int blz_spmv(blz_ctx *ctx, int transpose, int src_block, int dst_block);

/* block_dot_products(), :443-453: vtAv = v^T*Av, vtAAv = Av^T*Av (kept on the device;
 * copied to the host arrays when they are not NULL).  With nranks > 1 the partial products are
 * summed over ranks (RCCL all-reduce of 2*n*n words). */
int blz_block_dot(blz_ctx *ctx, uint64_t *vtAv, uint64_t *vtAAv);

/* semi_inverse(vtAv, winv, d), :342-438, on the device copy of vtAv (set it with blz_set_small
 * or blz_block_dot).  Same pivot order, hence the same d and winv. */
int blz_semi_inverse(blz_ctx *ctx, int *npiv, uint64_t *winv, uint64_t *d);

/* orthogonalize(), :456-492, followed by the copy v <- tmp of :655-656 (done in place). */
int blz_orthogonalize(blz_ctx *ctx);

/* The loop body of block_lanczos(), :631-659, up to max_iters times without host round trips.
 * *done = iterations completed (the reference's n_iterations increments), *stopped = 1 once
 * semi_inverse returned 0 (:644-650); later calls are then no-ops.  *ms (optional) = device time
 * of this call measured with HIP events on the context's stream. */
int blz_iterate(blz_ctx *ctx, int max_iters, int *done, int *stopped, float *ms);
int64_t blz_iterations(const blz_ctx *ctx);
int blz_set_iterations(blz_ctx *ctx, int64_t iterations);	/* --load-checkpoint */

/* final_check(), :560-582, on V and on TMP (= M^T v of the last iteration). */
int blz_final_check(blz_ctx *ctx, int *v_nonzero, int *vtm_zero);

/* Asynchronous snapshot of (v, p, iteration count) for checkpoints (openMP/lanczos_modp.c:1013-1022 stops its loop
 * for them).  blz_snapshot_begin, called between two blz_iterate calls, enqueues the device-to-host copies of this
 * rank's rows on a stream of their own and returns; the GPU pauses for the transfer only and the caller goes on
 * iterating.  blz_snapshot_wait blocks until the copies have landed and writes this rank's rows into v and p (whole
 * blocks of rows(V) x n words, original numbering; other ranks' rows untouched).  It may be called from ANOTHER host
 * thread (the checkpoint writer) while the owner is inside blz_iterate -- the one exception to one thread per handle
 * (the in-flight mark is an atomic: begin in the owner's thread sees a wait that finished in the writer's).
 * v == p == NULL drops the snapshot: a writer that has failed (memory, one rank's copy) must still collect from EVERY
 * context, or their next blz_snapshot_begin is refused.  One snapshot in flight per context. */
int blz_snapshot_begin(blz_ctx *c);
int blz_snapshot_wait(blz_ctx *c, uint64_t *v, uint64_t *p, int64_t *iterations);

/* Measurement: run one hot-path kernel `reps` times between two HIP events on the context's
 * stream and return the mean time.  which: 0 = first SpMV of an iteration (:635), 1 = second (:636, without
 * the fused block products), 2 = block_dot + finalize, 3 = orthogonalize (updates V and P in place: call
 * blz_init_v or blz_set_block afterwards if the blocks are to be used again). */
int blz_time_kernel(blz_ctx *ctx, int which, int reps, float *ms_mean);
int blz_sync(blz_ctx *ctx);

/* Per-kernel HIP-event spans inside blz_iterate (on the stream the kernels run on).  blz_profile(ctx,1)
 * clears and starts collecting, blz_profile(ctx,0) stops; blz_profile_read sums the spans collected so
 * far into 8 classes: 0 first SpMV, 1 second SpMV, 2 block_dot (+finalize), 3 semi_inverse,
 * 4 orthogonalize, 5 all-gather of v, 6 all-gather of tmp, 7 all-reduce of the n x n products. */
#define BLZ_PROFILE_CLASSES 8
int blz_profile(blz_ctx *ctx, int enable);
int blz_profile_read(blz_ctx *ctx, double ms_sum[BLZ_PROFILE_CLASSES], int64_t launches[BLZ_PROFILE_CLASSES]);

/* Exchange mode of a multi-rank context: 0 (default) = the library issues the RCCL collectives itself
 * inside blz_iterate / blz_block_dot; 1 = external: collectives are skipped and the caller moves the
 * slabs between ranks with blz_get_block / blz_set_block / blz_get_small / blz_set_small (used by the
 * single-GPU emulation tests of the sharded schedule; blz_iterate is refused in this mode). */
int blz_set_exchange_mode(blz_ctx *ctx, int external);

/* Multi-GPU (one process per GPU).  id_bytes = ncclUniqueId from blz_comm_unique_id() on rank 0,
 * broadcast by the caller (bench.py uses torch.distributed for that and nothing else).  Call blz_comm_init before
 * blz_set_matrix.  Inside blz_iterate every product is pipelined against the exchange of its operand: the block is
 * all-gathered in K pieces on a second stream and the product runs piece by piece behind it (K = up to 4 pieces of
 * >= 2 MB per slab, or BLZ_AG_CHUNKS). */
int blz_comm_unique_id(void *id_out, size_t id_bytes);	/* needs id_bytes >= 128 */
int blz_comm_init(blz_ctx *ctx, const void *id, size_t id_bytes, int rank, int nranks);
/* What the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller passed in: -1, -1 when the
 * context has none.  bench.py prints it so that an N > 1 line proves N ranks really met inside RCCL
 * (the reference prints its MPI_Comm_size, mpi/lanczos_modp.c:1755-1757). */
int blz_comm_info(const blz_ctx *ctx, int *nranks_seen, int *rank_seen);
/* pieces the exchange of product `transpose`'s operand (and the product itself) is cut into; 0 in the short-side form,
 * where nothing is gathered */
/* Loopback communicator: the contexts of ONE process on ONE device, one host thread each, as the ranks of a job -- RCCL
 * refuses two ranks on one GPU, and every other part of the multi-rank path (slabs, gathered layouts, the piece pipeline on
 * two streams, the landing buffers of the collectives, what a batch does past the stop) is then the production code run
 * with real multi-rank sums on a one-GPU box.  Create one group, attach every context (instead of blz_comm_init), drive
 * each context from its own thread: the collectives inside blz_iterate / blz_final_check meet in the group (a rank that
 * does not arrive within 120 s fails all of them with BLZ_ECOMM).  Not a transport: nothing leaves the device. */
typedef struct blz_loop_group blz_loop_group;
int blz_loop_group_create(int nranks, blz_loop_group **out);	/* at most 16 ranks */
void blz_loop_group_destroy(blz_loop_group *g);			/* after every attached context has been destroyed */
int blz_comm_init_loopback(blz_ctx *ctx, blz_loop_group *g, int rank);

int blz_exchange_pieces(const blz_ctx *ctx, int transpose);
/* the number of pieces this context would have a matrix of that shape prepared in for nranks ranks (what blz_prepare_for
 * passes to blz_prepare): for callers that prepare a rank's share themselves (blz_prepare_rank) */
int blz_exchange_pieces_for(const blz_ctx *ctx, int64_t mrows, int64_t mcols, int64_t nnz, int nranks);

#ifdef __cplusplus
}
#endif
#endif
