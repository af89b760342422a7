/*
 * blz_oracle.c -- CPU oracle for the block-Lanczos-mod-p hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see blz_oracle.h).  A restatement, in this
 * project's own words, of the algorithm in /root/reference
 * sequential/lanczos_modp.c, with residues widened to u64 and products
 * taken in 128 bits so that any prime below 2^62 is in range.  Each
 * function cites the reference lines whose behaviour it must reproduce.
 *
 * Parity status: pinned.  tests/test_oracle_golden.py compares every
 * function here with vectors produced by the reference itself
 * (oracle/_ref/libref_seq.so, built from the reference's own sources by
 * oracle/Makefile; generator tests/golden/make_golden.py) for
 * p in {65537, 1073741789, 2^31-1}.  For p > 2^32 no reference
 * implementation exists (the reference stores u32 and caps p at 2^30-35,
 * sequential/lanczos_modp.c:189-193): there the oracle is the same code
 * path, cross-checked against exact Python integers on small cases.
 */
#define _POSIX_C_SOURCE 200809L
#include "blz_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

static inline u64 mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
static inline u64 macmod(u64 acc, u64 a, u64 b, u64 p) { return (u64)(((u128)a * b + acc) % p); }

/* ------------------------------------------------------------------ RNG */

/* sequential/lanczos_modp.c:67 -- the four fixed seed words. */
void orc_rng_seed(u64 s[4])
{
	s[0] = 0x1415926535ull;
	s[1] = 0x8979323846ull;
	s[2] = 0x2643383279ull;
	s[3] = 0x5028841971ull;
}

static inline u64 rot_left(u64 x, int k) { return (x << k) | (x >> (64 - k)); }

/* sequential/lanczos_modp.c:76-87: output is rotl(s0+s3,23)+s0, then the
 * xoshiro256 state transition with shift 17 and final rotation 45. */
u64 orc_rng_next(u64 s[4])
{
	const u64 out = rot_left(s[0] + s[3], 23) + s[0];
	const u64 carry = s[1] << 17;
	s[2] ^= s[0];
	s[3] ^= s[1];
	s[1] ^= s[2];
	s[0] ^= s[3];
	s[2] ^= carry;
	s[3] = rot_left(s[3], 45);
	return out;
}

/* --------------------------------------------------------------- loader */

static int fail(char *err, size_t n, const char *msg)
{
	if (err && n)
		snprintf(err, n, "%s", msg);
	return -1;
}

static void lower(char *s)
{
	for (; *s; s++)
		*s = (char)tolower((unsigned char)*s);
}

/* Banner rules of mmio.c:28-111 as used at sequential/lanczos_modp.c:214-221:
 * "%%MatrixMarket matrix <coordinate|array> <integer> <general>".
 * want_array selects the dense ("array") form used for kernel blocks. */
static int read_banner(FILE *f, int want_array, char *err, size_t errlen)
{
	char line[1100], w0[64], w1[64], w2[64], w3[64], w4[64];
	if (!fgets(line, sizeof line, f))
		return fail(err, errlen, "Could not process Matrix Market banner.");
	if (sscanf(line, "%63s %63s %63s %63s %63s", w0, w1, w2, w3, w4) != 5)
		return fail(err, errlen, "Could not process Matrix Market banner.");
	lower(w1); lower(w2); lower(w3); lower(w4);
	if (strncmp(w0, "%%MatrixMarket", 14) != 0 || strcmp(w1, "matrix") != 0)
		return fail(err, errlen, "Could not process Matrix Market banner.");
	if (strcmp(w2, "coordinate") != 0 && strcmp(w2, "array") != 0)
		return fail(err, errlen, "Could not process Matrix Market banner.");
	if (strcmp(w3, "real") && strcmp(w3, "complex") && strcmp(w3, "pattern") && strcmp(w3, "integer"))
		return fail(err, errlen, "Could not process Matrix Market banner.");
	if (strcmp(w4, "general") && strcmp(w4, "symmetric") && strcmp(w4, "hermitian") && strcmp(w4, "skew-symmetric"))
		return fail(err, errlen, "Could not process Matrix Market banner.");
	if (want_array ? strcmp(w2, "array") != 0 : strcmp(w2, "coordinate") != 0)
		return fail(err, errlen, want_array ? "only dense matrices are OK" : "only sparse matrices are OK");
	if (strcmp(w3, "integer") != 0 || strcmp(w4, "general") != 0)
		return fail(err, errlen, "only integer general are OK");
	return 0;
}

/* mmio.c:113-141: skip '%' lines, then the size line. */
static int read_size_line(FILE *f, char *line, size_t len)
{
	do {
		if (!fgets(line, (int)len, f))
			return -1;
	} while (line[0] == '%');
	return 0;
}

int orc_mm_load(const char *path, u64 p, orc_coo *out, char *err, size_t errlen)
{
	memset(out, 0, sizeof *out);
	FILE *f = fopen(path, "r");
	if (!f)
		return fail(err, errlen, "cannot open matrix file");
	if (read_banner(f, 0, err, errlen)) {
		fclose(f);
		return -1;
	}
	char line[1100];
	int nr, nc;
	long nz;
	if (read_size_line(f, line, sizeof line) || sscanf(line, "%d %d %ld", &nr, &nc, &nz) != 3) {
		fclose(f);
		return fail(err, errlen, "Cannot read matrix size");
	}
	out->nrows = nr;
	out->ncols = nc;
	out->nnz = nz;
	out->i = malloc(sizeof(int32_t) * (size_t)(nz ? nz : 1));
	out->j = malloc(sizeof(int32_t) * (size_t)(nz ? nz : 1));
	out->x = malloc(sizeof(uint32_t) * (size_t)(nz ? nz : 1));
	for (long u = 0; u < nz; u++) {
		int a, b, c;
		/* sequential/lanczos_modp.c:238-243: value scanned with %d into a
		 * u32, so a negative entry wraps to 2^32-|x| BEFORE the % p. */
		if (fscanf(f, "%d %d %d", &a, &b, &c) != 3) {
			fclose(f);
			orc_coo_free(out);
			return fail(err, errlen, "parse error in matrix entries");
		}
		out->i[u] = a - 1;
		out->j[u] = b - 1;
		out->x[u] = (uint32_t)((u64)(uint32_t)c % p);
	}
	fclose(f);
	return 0;
}

void orc_coo_free(orc_coo *M)
{
	free(M->i);
	free(M->j);
	free(M->x);
	memset(M, 0, sizeof *M);
}

/* ----------------------------------------------------------------- SpMV */

/* sequential/lanczos_modp.c:266-287.  The reference reduces after every
 * term; residues are canonical so any order gives the same words. */
void orc_spmv(u64 *y, const orc_coo *M, const u64 *x, int transpose, int n, u64 p)
{
	const int64_t rows_out = transpose ? M->ncols : M->nrows;
	memset(y, 0, sizeof(u64) * (size_t)(rows_out * n));
	for (int64_t k = 0; k < M->nnz; k++) {
		const int64_t r = transpose ? M->j[k] : M->i[k];
		const int64_t c = transpose ? M->i[k] : M->j[k];
		const u64 a = M->x[k];
		u64 *yr = y + r * n;
		const u64 *xc = x + c * n;
		for (int l = 0; l < n; l++)
			yr[l] = macmod(yr[l], a, xc[l], p);
	}
}

/* ------------------------------------------------------- block products */

/* sequential/lanczos_modp.c:443-453 with :305-315 folded in: the reference
 * walks n-row tiles (reading zero padding); summing row by row is the same
 * exact sum. */
void orc_block_dot(u64 *vtAv, u64 *vtAAv, int64_t N, const u64 *Av, const u64 *v, int n, u64 p)
{
	for (int e = 0; e < n * n; e++)
		vtAv[e] = vtAAv[e] = 0;
	for (int64_t r = 0; r < N; r++) {
		const u64 *vr = v + r * n, *ar = Av + r * n;
		for (int a = 0; a < n; a++)
			for (int b = 0; b < n; b++) {
				vtAv[a * n + b] = macmod(vtAv[a * n + b], vr[a], ar[b], p);
				vtAAv[a * n + b] = macmod(vtAAv[a * n + b], ar[a], ar[b], p);
			}
	}
}

/* sequential/lanczos_modp.c:318-336.  Same recurrence, 128-bit signed. */
u64 orc_invmod(u64 a, u64 p)
{
	__int128 t = 0, nt = 1, r = p, nr = a % p;
	while (nr != 0) {
		const __int128 q = r / nr;
		__int128 s = nt;
		nt = t - q * nt;
		t = s;
		s = nr;
		nr = r - q * nr;
		r = s;
	}
	if (t < 0)
		t += p;
	return (u64)t;
}

/*
 * One Gauss-Jordan sweep over `a` (n x n), optionally mirrored on `w`.
 * Pivot rule of sequential/lanczos_modp.c:351-381 / :393-436: for column j
 * take the FIRST non-zero entry in rows j..n-1; if there is none the column
 * is skipped (and row j is not reused later); otherwise scale the pivot row
 * to make the pivot 1, swap it into row j, eliminate column j everywhere
 * else.  Marks d[j] and returns the pivot count.
 */
static int gauss_sweep(u64 *a, u64 *w, u64 *d, int n, u64 p)
{
	int found = 0;
	for (int j = 0; j < n; j++) {
		int piv = -1;
		for (int r = j; r < n && piv < 0; r++)
			if (a[r * n + j] != 0)
				piv = r;
		if (piv < 0)
			continue;
		d[j] = 1;
		found++;
		const u64 inv = orc_invmod(a[piv * n + j], p);
		for (int k = 0; k < n; k++) {
			a[piv * n + k] = mulmod(a[piv * n + k], inv, p);
			if (w)
				w[piv * n + k] = mulmod(w[piv * n + k], inv, p);
		}
		for (int k = 0; k < n; k++) {
			u64 s = a[j * n + k];
			a[j * n + k] = a[piv * n + k];
			a[piv * n + k] = s;
			if (w) {
				s = w[j * n + k];
				w[j * n + k] = w[piv * n + k];
				w[piv * n + k] = s;
			}
		}
		for (int r = 0; r < n; r++) {
			if (r == j)
				continue;
			const u64 neg = p - a[r * n + j];	/* may equal p: harmless */
			for (int k = 0; k < n; k++) {
				a[r * n + k] = macmod(a[r * n + k], neg, a[j * n + k], p);
				if (w)
					w[r * n + k] = macmod(w[r * n + k], neg, w[j * n + k], p);
			}
		}
	}
	return found;
}

/* sequential/lanczos_modp.c:342-438 */
int orc_semi_inverse(const u64 *M, u64 *winv, u64 *d, int n, u64 p)
{
	u64 *a = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *sel = malloc(sizeof(u64) * (size_t)n);
	memcpy(a, M, sizeof(u64) * (size_t)(n * n));
	for (int k = 0; k < n; k++)
		sel[k] = 0;
	gauss_sweep(a, NULL, sel, n, p);			/* phase 1: which columns */
	for (int r = 0; r < n; r++)				/* :384-388 */
		for (int c = 0; c < n; c++) {
			a[r * n + c] = (sel[r] && sel[c]) ? M[r * n + c] : 0;
			winv[r * n + c] = (r == c && sel[r]) ? 1 : 0;
		}
	for (int k = 0; k < n; k++)
		d[k] = 0;
	const int npiv = gauss_sweep(a, winv, d, n, p);	/* phase 2 */
	free(a);
	free(sel);
	return npiv;
}

/* sequential/lanczos_modp.c:456-492.  c and vtAvd keep the reference's
 * "p - x" form (so an entry may be p rather than 0); outputs are canonical. */
static void ortho_coeffs(u64 *c, u64 *vtAvd, const u64 *d, const u64 *vtAv, const u64 *vtAAv,
			 const u64 *winv, int n, u64 p)
{
	for (int r = 0; r < n; r++)
		for (int col = 0; col < n; col++) {
			u64 acc = 0;
			for (int k = 0; k < n; k++) {
				const u64 s = d[col] ? vtAAv[k * n + col] : vtAv[k * n + col];
				acc = macmod(acc, winv[r * n + k], s, p);
			}
			c[r * n + col] = p - acc;
			vtAvd[r * n + col] = d[col] ? p - vtAv[r * n + col] : 0;
		}
}

static void ortho_row(const u64 *vr, const u64 *ar, u64 *tr, u64 *pr, const u64 *d, const u64 *c,
		      const u64 *vtAvd, const u64 *winv, int n, u64 p, u64 *scratch)
{
	for (int col = 0; col < n; col++) {
		u64 nv = d[col] ? ar[col] : vr[col];
		u64 np = d[col] ? 0 : pr[col];
		for (int k = 0; k < n; k++)
			nv = macmod(nv, vr[k], c[k * n + col], p);
		for (int k = 0; k < n; k++)
			nv = macmod(nv, pr[k], vtAvd[k * n + col], p);
		for (int k = 0; k < n; k++)
			np = macmod(np, vr[k], winv[k * n + col], p);
		tr[col] = nv;
		scratch[col] = np;
	}
	for (int col = 0; col < n; col++)
		pr[col] = scratch[col];
}

void orc_orthogonalize(const u64 *v, u64 *tmp, u64 *pblk, const u64 *d, const u64 *vtAv,
		       const u64 *vtAAv, const u64 *winv, int64_t N, const u64 *Av, int n, u64 p)
{
	u64 *c = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *vtAvd = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *scratch = malloc(sizeof(u64) * (size_t)n);
	ortho_coeffs(c, vtAvd, d, vtAv, vtAAv, winv, n, p);
	for (int64_t r = 0; r < N; r++)
		ortho_row(v + r * n, Av + r * n, tmp + r * n, pblk + r * n, d, c, vtAvd, winv, n, p, scratch);
	free(c);
	free(vtAvd);
	free(scratch);
}

/* ---------------------------------------------------------------- driver */

int orc_final_check(int64_t nrows, int64_t ncols, int n, const u64 *v, const u64 *vtM)
{
	int nonzero = 0, zero = 1;
	for (int64_t k = 0; k < nrows * n; k++)
		nonzero |= (v[k] != 0);
	for (int64_t k = 0; k < ncols * n; k++)
		zero &= (vtM[k] == 0);
	return nonzero | (zero << 1);
}

int orc_block_lanczos(const orc_coo *M, int n, u64 p, int right, int stop_after, u64 *v_out,
		      u64 *tmp_out, u64 *p_out, const u64 *v_init, const u64 *p_init, int start_iter,
		      orc_trace_fn cb, void *user)
{
	/* sequential/lanczos_modp.c:592-597: "rows" are those of the block v. */
	const int64_t nrows = right ? M->ncols : M->nrows;
	const int64_t ncols = right ? M->nrows : M->ncols;
	const int64_t big = (nrows > ncols ? nrows : ncols) * n;
	u64 *v = calloc((size_t)big, sizeof(u64));
	u64 *tmp = calloc((size_t)big, sizeof(u64));
	u64 *Av = calloc((size_t)big, sizeof(u64));
	u64 *pb = calloc((size_t)big, sizeof(u64));
	u64 *vtAv = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *vtAAv = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *winv = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *d = malloc(sizeof(u64) * (size_t)n);

	if (v_init) {
		memcpy(v, v_init, sizeof(u64) * (size_t)(nrows * n));
		if (p_init)
			memcpy(pb, p_init, sizeof(u64) * (size_t)(nrows * n));
	} else {
		/* :624-625: row-major draw order, one draw per word. */
		u64 s[4];
		orc_rng_seed(s);
		for (int64_t k = 0; k < nrows * n; k++)
			v[k] = orc_rng_next(s) % p;
	}

	int it = start_iter;
	for (;;) {
		if (stop_after > 0 && it == stop_after)		/* :632 */
			break;
		orc_spmv(tmp, M, v, !right, n, p);		/* :635 */
		orc_spmv(Av, M, tmp, right, n, p);		/* :636 */
		orc_block_dot(vtAv, vtAAv, nrows, Av, v, n, p);
		const int npiv = orc_semi_inverse(vtAv, winv, d, n, p);
		if (cb) {
			orc_trace t = { it, npiv, vtAv, vtAAv, winv, d, v, tmp, Av, pb };
			cb(&t, user);
		}
		if (npiv == 0)					/* :644,649 */
			break;
		orc_orthogonalize(v, tmp, pb, d, vtAv, vtAAv, winv, nrows, Av, n, p);
		memcpy(v, tmp, sizeof(u64) * (size_t)(nrows * n));	/* :655-656 */
		it++;						/* verbosity(), :496 */
	}
	memcpy(v_out, v, sizeof(u64) * (size_t)(nrows * n));
	if (tmp_out)
		memcpy(tmp_out, tmp, sizeof(u64) * (size_t)(ncols * n));
	if (p_out)
		memcpy(p_out, pb, sizeof(u64) * (size_t)(nrows * n));
	free(v); free(tmp); free(Av); free(pb);
	free(vtAv); free(vtAAv); free(winv); free(d);
	return it;
}

/* ------------------------------------------------------------------- I/O */

/* sequential/lanczos_modp.c:673-686.  The reference prints u32 words with
 * "%d" (so words >= 2^31 appear negative and checker_modp reads them back
 * the same way, checker_modp.c:148); words >= 2^32 only exist beyond its
 * domain and are printed as plain unsigned decimals. */
int orc_save_block(const char *path, int64_t nrows, int n, const u64 *v)
{
	FILE *f = fopen(path, "w");
	if (!f)
		return -1;
	fprintf(f, "%%%%MatrixMarket matrix array integer general\n");
	fprintf(f, "%%block of left-kernel vector computed by lanczos_modp\n");
	fprintf(f, "%ld %d\n", (long)nrows, n);
	for (int col = 0; col < n; col++)
		for (int64_t r = 0; r < nrows; r++) {
			const u64 w = v[r * n + col];
			if (w < 0x100000000ull)
				fprintf(f, "%d\n", (int)(uint32_t)w);
			else
				fprintf(f, "%llu\n", (unsigned long long)w);
		}
	return fclose(f) ? -1 : 0;
}

/* checker_modp.c:81-204, widened to u64 words.  Entries of the kernel file
 * are parsed as signed decimals; a negative one is taken as the u32 the
 * reference's "%d" would have stored. */
int orc_check_kernel(const char *matrix_path, const char *kernel_path, u64 p, int right, char *err,
		     size_t errlen)
{
	FILE *mf = fopen(matrix_path, "r");
	if (!mf)
		return fail(err, errlen, "cannot open matrix file");
	if (read_banner(mf, 0, err, errlen)) {
		fclose(mf);
		return -1;
	}
	char line[1100];
	long nr, nc, nz;
	if (read_size_line(mf, line, sizeof line) || sscanf(line, "%ld %ld %ld", &nr, &nc, &nz) != 3) {
		fclose(mf);
		return fail(err, errlen, "Cannot read matrix size");
	}
	if (right) {
		long t = nr;
		nr = nc;
		nc = t;
	}
	FILE *kf = fopen(kernel_path, "r");
	if (!kf) {
		fclose(mf);
		return fail(err, errlen, "cannot open kernel file");
	}
	long nk;
	int n;
	if (read_banner(kf, 1, err, errlen) || read_size_line(kf, line, sizeof line)
	    || sscanf(line, "%ld %d", &nk, &n) != 2) {
		fclose(mf);
		fclose(kf);
		return fail(err, errlen, "Cannot read kernel vector block size");
	}
	if (nk != nr) {
		fclose(mf);
		fclose(kf);
		return fail(err, errlen, "dimension mismatch");
	}
	u64 *x = malloc(sizeof(u64) * (size_t)(nr * n + 1));
	u64 *y = calloc((size_t)(nc * n + 1), sizeof(u64));
	int rc = 0, any = 0;
	for (int col = 0; col < n && rc == 0; col++)
		for (long r = 0; r < nr; r++) {
			long long w;
			if (fscanf(kf, "%lld", &w) != 1) {
				rc = fail(err, errlen, "parse error in kernel entries");
				break;
			}
			const u64 word = w < 0 ? (u64)(uint32_t)(int)w : (u64)w;
			if (word >= p) {
				rc = fail(err, errlen, "kernel entry out of bound");
				break;
			}
			x[r * n + col] = word;
			any |= (word != 0);
		}
	if (rc == 0 && !any)
		rc = 1;
	for (long u = 0; u < nz && rc == 0; u++) {
		int a, b, c;
		if (fscanf(mf, "%d %d %d", &a, &b, &c) != 3) {
			rc = fail(err, errlen, "parse error in matrix entries");
			break;
		}
		long r = a - 1, cc = b - 1;
		if (right) {
			long t = r;
			r = cc;
			cc = t;
		}
		const u64 val = (u64)(uint32_t)c % p;
		for (int k = 0; k < n; k++)
			y[cc * n + k] = macmod(y[cc * n + k], val, x[r * n + k], p);
	}
	if (rc == 0)
		for (long k = 0; k < nc * n; k++)
			if (y[k] != 0) {
				rc = 2;
				break;
			}
	free(x);
	free(y);
	fclose(mf);
	fclose(kf);
	return rc;
}

/* ------------------------------------------------- OpenMP baseline kernels */

static int pick_threads(int nthreads)
{
#ifdef _OPENMP
	return nthreads > 0 ? nthreads : omp_get_max_threads();
#else
	(void)nthreads;
	return 1;
#endif
}

/*
 * Strategy of openMP/lanczos_modp.c:329-374: the nnz loop is split across
 * threads in COO order, every thread accumulates UNREDUCED sums into a
 * private copy of the output block, the copies are summed and reduced once
 * at the end.  The reference keeps u64 sums (which overflow on long rows of
 * large values); this restatement keeps 128-bit sums, so it is exact for
 * every input.
 */
void orc_spmv_omp(u64 *y, const orc_coo *M, const u64 *x, int transpose, int n, u64 p, int nthreads)
{
	const int T = pick_threads(nthreads);
	const int64_t rows_out = transpose ? M->ncols : M->nrows;
	const int64_t words = rows_out * n;
	u128 *priv = calloc((size_t)(words * T), sizeof(u128));
#pragma omp parallel num_threads(T)
	{
#ifdef _OPENMP
		const int me = omp_get_thread_num();
#else
		const int me = 0;
#endif
		u128 *mine = priv + (int64_t)me * words;
		const int64_t lo = M->nnz * me / T, hi = M->nnz * (me + 1) / T;
		for (int64_t k = lo; k < hi; k++) {
			const int64_t r = transpose ? M->j[k] : M->i[k];
			const int64_t c = transpose ? M->i[k] : M->j[k];
			const u64 a = M->x[k];
			u128 *yr = mine + r * n;
			const u64 *xc = x + c * n;
			for (int l = 0; l < n; l++)
				yr[l] += (u128)a * xc[l];
		}
#pragma omp barrier
#pragma omp for
		for (int64_t w = 0; w < words; w++) {
			u64 acc = 0;
			for (int t = 0; t < T; t++) {
				acc += (u64)(priv[(int64_t)t * words + w] % p);
				if (acc >= p)
					acc -= p;
			}
			y[w] = acc;
		}
	}
	free(priv);
}

/*
 * The same product by rows: a CSR of M (or of M^T) built once, one output row per loop iteration, 128-bit sums
 * in registers, one reduction per word.  No private copies of the output (orc_spmv_omp zeroes and sums
 * nthreads x rows x n u128 words per call, which is what it spends most of its time on at 16 threads), so this
 * is the form bench.py times as cpu_baseline; openMP/lanczos_modp.c:329-374 stays restated above.
 */
struct orc_csr {
	int64_t rows, nnz;
	int64_t *row_ptr;
	int32_t *col;
	uint32_t *val;
};

void orc_csr_free(orc_csr *A)
{
	if (!A)
		return;
	free(A->row_ptr);
	free(A->col);
	free(A->val);
	free(A);
}

orc_csr *orc_csr_build(const orc_coo *M, int transpose)
{
	orc_csr *A = calloc(1, sizeof *A);
	if (!A)
		return NULL;
	A->rows = transpose ? M->ncols : M->nrows;
	A->nnz = M->nnz;
	A->row_ptr = calloc((size_t)A->rows + 2, sizeof *A->row_ptr);
	A->col = malloc(sizeof *A->col * (size_t)(M->nnz ? M->nnz : 1));
	A->val = malloc(sizeof *A->val * (size_t)(M->nnz ? M->nnz : 1));
	int64_t *fill = malloc(sizeof *fill * (size_t)(A->rows + 1));
	if (!A->row_ptr || !A->col || !A->val || !fill) {
		free(fill);
		orc_csr_free(A);
		return NULL;
	}
	for (int64_t k = 0; k < M->nnz; k++)
		A->row_ptr[(transpose ? M->j[k] : M->i[k]) + 1]++;
	for (int64_t r = 0; r < A->rows; r++)
		A->row_ptr[r + 1] += A->row_ptr[r];
	memcpy(fill, A->row_ptr, sizeof *fill * (size_t)(A->rows + 1));
	for (int64_t k = 0; k < M->nnz; k++) {
		const int64_t at = fill[transpose ? M->j[k] : M->i[k]]++;
		A->col[at] = (int32_t)(transpose ? M->i[k] : M->j[k]);
		A->val[at] = (uint32_t)M->x[k];
	}
	free(fill);
	return A;
}

void orc_spmv_csr_omp(u64 *y, const orc_csr *A, const u64 *x, int n, u64 p, int nthreads)
{
	const int T = pick_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 2048) num_threads(T)
	for (int64_t r = 0; r < A->rows; r++) {
		u128 acc[64];	/* n <= 64 (BLZ_MAX_N) */
		for (int l = 0; l < n; l++)
			acc[l] = 0;
		for (int64_t k = A->row_ptr[r]; k < A->row_ptr[r + 1]; k++) {
			const u64 a = A->val[k];
			const u64 *xc = x + (int64_t)A->col[k] * n;
			for (int l = 0; l < n; l++)
				acc[l] += (u128)a * xc[l];
		}
		for (int l = 0; l < n; l++)
			y[r * n + l] = (u64)(acc[l] % p);
	}
}

int orc_iteration_csr_omp(const orc_csr *A, const orc_csr *At, int64_t nrows, int n, u64 p, int right, u64 *v, u64 *tmp,
			  u64 *Av, u64 *pblk, int nthreads)
{
	/* sequential/lanczos_modp.c:635-656 with tmp = (right ? M : M^T) v, Av = (right ? M^T : M) tmp */
	u64 *sm = malloc(sizeof(u64) * (size_t)(3 * n * n + n));
	u64 *vtAv = sm, *vtAAv = sm + n * n, *winv = sm + 2 * n * n, *d = sm + 3 * n * n;
	orc_spmv_csr_omp(tmp, right ? A : At, v, n, p, nthreads);
	orc_spmv_csr_omp(Av, right ? At : A, tmp, n, p, nthreads);
	orc_block_dot_omp(vtAv, vtAAv, nrows, Av, v, n, p, nthreads);
	const int npiv = orc_semi_inverse(vtAv, winv, d, n, p);
	if (npiv) {
		orc_orthogonalize_omp(v, tmp, pblk, d, vtAv, vtAAv, winv, nrows, Av, n, p, nthreads);
		memcpy(v, tmp, sizeof(u64) * (size_t)(nrows * n));
	}
	free(sm);
	return npiv;
}

/* openMP/lanczos_modp.c:681-712: per-thread n x n partials, combined once. */
void orc_block_dot_omp(u64 *vtAv, u64 *vtAAv, int64_t N, const u64 *Av, const u64 *v, int n, u64 p,
		       int nthreads)
{
	const int T = pick_threads(nthreads);
	const int nn = n * n;
	u64 *part = calloc((size_t)(2 * nn * T), sizeof(u64));
#pragma omp parallel num_threads(T)
	{
#ifdef _OPENMP
		const int me = omp_get_thread_num();
#else
		const int me = 0;
#endif
		u64 *a1 = part + (size_t)me * 2 * nn, *a2 = a1 + nn;
		const int64_t lo = N * me / T, hi = N * (me + 1) / T;
		for (int64_t r = lo; r < hi; r++) {
			const u64 *vr = v + r * n, *ar = Av + r * n;
			for (int a = 0; a < n; a++)
				for (int b = 0; b < n; b++) {
					a1[a * n + b] = macmod(a1[a * n + b], vr[a], ar[b], p);
					a2[a * n + b] = macmod(a2[a * n + b], ar[a], ar[b], p);
				}
		}
	}
	for (int e = 0; e < nn; e++) {
		u64 s1 = 0, s2 = 0;
		for (int t = 0; t < T; t++) {
			s1 = (u64)(((u128)s1 + part[(size_t)t * 2 * nn + e]) % p);
			s2 = (u64)(((u128)s2 + part[(size_t)t * 2 * nn + nn + e]) % p);
		}
		vtAv[e] = s1;
		vtAAv[e] = s2;
	}
	free(part);
}

/* openMP/lanczos_modp.c:757-796: rows split across threads. */
void orc_orthogonalize_omp(const u64 *v, u64 *tmp, u64 *pblk, const u64 *d, const u64 *vtAv,
			   const u64 *vtAAv, const u64 *winv, int64_t N, const u64 *Av, int n, u64 p,
			   int nthreads)
{
	const int T = pick_threads(nthreads);
	u64 *c = malloc(sizeof(u64) * (size_t)(n * n));
	u64 *vtAvd = malloc(sizeof(u64) * (size_t)(n * n));
	ortho_coeffs(c, vtAvd, d, vtAv, vtAAv, winv, n, p);
#pragma omp parallel num_threads(T)
	{
		u64 scratch[64 > 1 ? 64 : 1];
		u64 *sc = n <= 64 ? scratch : malloc(sizeof(u64) * (size_t)n);
#pragma omp for
		for (int64_t r = 0; r < N; r++)
			ortho_row(v + r * n, Av + r * n, tmp + r * n, pblk + r * n, d, c, vtAvd, winv, n, p, sc);
		if (sc != scratch)
			free(sc);
	}
	free(c);
	free(vtAvd);
}

int orc_iteration_omp(const orc_coo *M, int n, u64 p, int right, u64 *v, u64 *tmp, u64 *Av,
		      u64 *pblk, int nthreads)
{
	const int64_t nrows = right ? M->ncols : M->nrows;
	u64 *sm = malloc(sizeof(u64) * (size_t)(3 * n * n + n));
	u64 *vtAv = sm, *vtAAv = sm + n * n, *winv = sm + 2 * n * n, *d = sm + 3 * n * n;
	orc_spmv_omp(tmp, M, v, !right, n, p, nthreads);
	orc_spmv_omp(Av, M, tmp, right, n, p, nthreads);
	orc_block_dot_omp(vtAv, vtAAv, nrows, Av, v, n, p, nthreads);
	const int npiv = orc_semi_inverse(vtAv, winv, d, n, p);
	if (npiv) {
		orc_orthogonalize_omp(v, tmp, pblk, d, vtAv, vtAAv, winv, nrows, Av, n, p, nthreads);
		memcpy(v, tmp, sizeof(u64) * (size_t)(nrows * n));
	}
	free(sm);
	return npiv;
}
