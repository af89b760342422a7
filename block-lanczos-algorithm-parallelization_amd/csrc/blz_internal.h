/* blz_internal.h -- shared between the plain-C host side and the HIP side of libblz_hip.so. */
#ifndef BLZ_INTERNAL_H
#define BLZ_INTERNAL_H

#include "blz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* printf-style; returns `code` so that callers can `return blz_fail(BLZ_EINVAL, "...")`. */
int blz_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

/* Column indices of a CSR slab are rewritten from global rows of the source block to positions in
 * the rank-major padded ("gathered") layout: row r of rank g's slab [b_g, b_{g+1}) lives at
 * g*stride + (r - b_g).  For one rank this is the identity. */
void blz_remap_columns(blz_csr *A, const int64_t *bounds, int parts, int64_t stride);

void blz_coo_relabel(const blz_coo *M, const int32_t *row_perm, const int32_t *col_perm, int32_t *new_i, int32_t *new_j);

/* Rows [r0, r1) of A as a standalone CSR (deep copy). */
int blz_csr_slab(const blz_csr *A, int64_t r0, int64_t r1, blz_csr *out);

#ifdef __cplusplus
}
#endif
#endif
