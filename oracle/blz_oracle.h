/*
 * blz_oracle.h -- CPU oracle for the block-Lanczos-mod-p hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The shipped path (libblz_hip.so and the
 * lanczos_modp CLI) never links or calls it.
 *
 * It is a from-scratch restatement of the algorithm of the reference program
 * sequential/lanczos_modp.c (paths relative to /root/reference), widened from
 * u32 residues to u64 residues so that primes up to 2^62 can be checked.
 * For p <= 2^30-35 (the reference's own domain) and for p = 2^31-1 it is
 * pinned bit-for-bit against the reference compiled from its own sources
 * (oracle/_ref, see oracle/Makefile) through tests/golden/.
 *
 * All blocks are row-major rows x n arrays of uint64_t canonical residues.
 */
#ifndef BLZ_ORACLE_H
#define BLZ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* COO triplets, 0-based, file order.  sequential/lanczos_modp.c:55-62 */
typedef struct {
	int64_t nrows, ncols, nnz;
	int32_t *i, *j;
	uint32_t *x;		/* already canonicalised: (u32)value % p */
} orc_coo;

/* xoshiro-style generator, fixed seed.  sequential/lanczos_modp.c:67-87 */
void orc_rng_seed(uint64_t s[4]);
uint64_t orc_rng_next(uint64_t s[4]);

/* MatrixMarket "coordinate integer general" loader with the reference's
 * value canonicalisation ("%d" into a u32, then % p).
 * sequential/lanczos_modp.c:199-263.  Returns 0 or -1 (message in err). */
int orc_mm_load(const char *path, uint64_t p, orc_coo *out, char *err, size_t errlen);
void orc_coo_free(orc_coo *M);

/* y = M*x (transpose=0) or y = M^T*x (transpose=1), n columns.
 * sequential/lanczos_modp.c:266-287 */
void orc_spmv(uint64_t *y, const orc_coo *M, const uint64_t *x, int transpose, int n, uint64_t p);

/* vtAv = v^T*Av, vtAAv = Av^T*Av over the first N rows.
 * sequential/lanczos_modp.c:443-453 (+ :305-315) */
void orc_block_dot(uint64_t *vtAv, uint64_t *vtAAv, int64_t N, const uint64_t *Av,
		   const uint64_t *v, int n, uint64_t p);

/* a^-1 mod p by extended Euclid.  sequential/lanczos_modp.c:318-336 */
uint64_t orc_invmod(uint64_t a, uint64_t p);

/* Two-phase Gauss-Jordan; returns #pivots of phase 2.
 * sequential/lanczos_modp.c:342-438 */
int orc_semi_inverse(const uint64_t *M, uint64_t *winv, uint64_t *d, int n, uint64_t p);

/* Row-local update: v' (written to tmp rows < N) and p' (in place).
 * sequential/lanczos_modp.c:456-492 */
void orc_orthogonalize(const uint64_t *v, uint64_t *tmp, uint64_t *pblk, const uint64_t *d,
		       const uint64_t *vtAv, const uint64_t *vtAAv, const uint64_t *winv,
		       int64_t N, const uint64_t *Av, int n, uint64_t p);

/* Per-iteration trace record handed to the callback of orc_block_lanczos. */
typedef struct {
	int iteration;		/* 0-based index of the iteration just computed */
	int npiv;
	const uint64_t *vtAv, *vtAAv, *winv, *d;	/* n*n, n*n, n*n, n */
	const uint64_t *v, *tmp, *Av, *p;		/* blocks BEFORE orthogonalize */
} orc_trace;
typedef void (*orc_trace_fn)(const orc_trace *, void *user);

/* The driver loop.  sequential/lanczos_modp.c:585-669.
 * v_out: nrows*n words (nrows = right ? M->ncols : M->nrows).
 * tmp_out (optional): ncols*n words = M^T v of the last iteration.
 * stop_after <= 0 means "run to termination".  Returns #iterations done.
 * v_init/p_init/start_iter (optional) resume from a checkpoint
 * (openMP/lanczos_modp.c:933-940). */
int orc_block_lanczos(const orc_coo *M, int n, uint64_t p, int right, int stop_after,
		      uint64_t *v_out, uint64_t *tmp_out, uint64_t *p_out,
		      const uint64_t *v_init, const uint64_t *p_init, int start_iter,
		      orc_trace_fn cb, void *user);

/* final_check, sequential/lanczos_modp.c:560-582: bit0 = v != 0, bit1 = vtM == 0 */
int orc_final_check(int64_t nrows, int64_t ncols, int n, const uint64_t *v, const uint64_t *vtM);

/* MatrixMarket array writer, column-major.  sequential/lanczos_modp.c:673-686 */
int orc_save_block(const char *path, int64_t nrows, int n, const uint64_t *v);

/* checker_modp.c:143-204 widened: 0 = OK, 1 = all-zero kernel, 2 = y != 0,
 * <0 = I/O or format error. */
int orc_check_kernel(const char *matrix_path, const char *kernel_path, uint64_t p, int right,
		     char *err, size_t errlen);

/* OpenMP restatement of the openMP/lanczos_modp.c strategy (cpu_baseline):
 * thread-private lazily-reduced accumulators, openMP/lanczos_modp.c:329-374.
 * Safe for any values (128-bit accumulators). */
void orc_spmv_omp(uint64_t *y, const orc_coo *M, const uint64_t *x, int transpose, int n,
		  uint64_t p, int nthreads);
void orc_block_dot_omp(uint64_t *vtAv, uint64_t *vtAAv, int64_t N, const uint64_t *Av,
		       const uint64_t *v, int n, uint64_t p, int nthreads);
void orc_orthogonalize_omp(const uint64_t *v, uint64_t *tmp, uint64_t *pblk, const uint64_t *d,
			   const uint64_t *vtAv, const uint64_t *vtAAv, const uint64_t *winv,
			   int64_t N, const uint64_t *Av, int n, uint64_t p, int nthreads);
/* The same product by rows of a CSR built once (no per-thread copies of the output): the form bench.py times. */
typedef struct orc_csr orc_csr;
orc_csr *orc_csr_build(const orc_coo *M, int transpose);
void orc_csr_free(orc_csr *A);
void orc_spmv_csr_omp(uint64_t *y, const orc_csr *A, const uint64_t *x, int n, uint64_t p, int nthreads);
/* A = CSR of M, At = CSR of M^T; nrows = rows of v */
int orc_iteration_csr_omp(const orc_csr *A, const orc_csr *At, int64_t nrows, int n, uint64_t p, int right,
			  uint64_t *v, uint64_t *tmp, uint64_t *Av, uint64_t *pblk, int nthreads);
/* One full iteration with the OpenMP kernels; returns npiv. v is updated in place. */
int orc_iteration_omp(const orc_coo *M, int n, uint64_t p, int right, uint64_t *v, uint64_t *tmp,
		      uint64_t *Av, uint64_t *pblk, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
