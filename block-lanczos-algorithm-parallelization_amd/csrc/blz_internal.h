/* blz_internal.h -- shared between the plain-C host side and the HIP side of libblz_hip.so. */
#ifndef BLZ_INTERNAL_H
#define BLZ_INTERNAL_H

#include "blz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Helpers shared by the two halves of the library; not part of the ABI: libblz_hip.so exports what include/blz.h declares
 * and nothing else (tests/test_host_abi.py checks both directions). */
#define BLZ_LOCAL __attribute__((visibility("hidden")))

/* printf-style; returns `code` so that callers can `return blz_fail(BLZ_EINVAL, "...")`. */
BLZ_LOCAL int blz_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

/* Column indices of a CSR slab are rewritten from global rows of the source block to positions in the
 * gathered operand: a slab of `chunks` pieces of `piece` rows each; row q of rank g's slab lives at
 *     (q / piece) * (parts * piece) + g * piece + (q % piece)
 * i.e. piece-major, rank-major inside a piece, so that all-gather number k of equal pieces lands contiguously.
 * With chunks == 1 this is the rank-major padded layout; for one rank it is the identity. */
BLZ_LOCAL void blz_remap_columns(blz_csr *A, const int64_t *bounds, int parts, int64_t piece, int chunks);
BLZ_LOCAL int blz_csr_split_columns(const blz_csr *A, int64_t width, int chunks, blz_csr *out);

BLZ_LOCAL void blz_coo_relabel(const blz_coo *M, const int32_t *row_perm, const int32_t *col_perm, int32_t *new_i, int32_t *new_j);

/* Rows [r0, r1) of A as a standalone CSR (deep copy). */
BLZ_LOCAL int blz_csr_slab(const blz_csr *A, int64_t r0, int64_t r1, blz_csr *out);

/* the prepared matrix (include/blz.h: blz_prepare); arrays are malloc'ed, or live in the mmapped cache file when map != NULL */
struct blz_prepared {
	int64_t nrows, ncols, nnz;		/* of M */
	int right, nranks, chunks, order_kind, has_perm;
	int64_t hot[2];				/* densest rows / columns of M numbered first (0 = none) */
	double share[2], locality[2];
	int32_t *perm[2];			/* new index of every row / column of M (has_perm) */
	int64_t *bounds[2];			/* per SIDE (0: rows of v, 1: rows of tmp): nranks + 1 row bounds */
	int64_t stride[2];			/* per side: rows of a padded slab */
	blz_csr full[2];			/* CSR of M and of M^T in the solver's numbering, global column indices */
	int only_rank;				/* -1: full[] hold every row; >= 0 (blz_prepare_rank): full[t] holds only that rank's rows, */
	int64_t full_first[2];			/* ... row q of full[t] being global row full_first[t] + q */
	void *map;
	size_t map_len;
};

#ifdef __cplusplus
}
#endif
#endif
