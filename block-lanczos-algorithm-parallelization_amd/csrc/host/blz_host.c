/*
 * blz_host.c -- the plain-C host side of libblz_hip.so: MatrixMarket ingest, CSR construction,
 * row partitioning, the fixed-seed generator, the result writer and checkpoint files.
 * Nothing here touches the GPU.  Reference citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#include "../blz_internal.h"

#include <ctype.h>
#include <errno.h>
#include <omp.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

/* ---------------------------------------------------------------------------- errors */

static __thread char g_err[512];

int blz_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

const char *blz_last_error(void) { return g_err; }
int blz_version(void) { return 100; }

/* ------------------------------------------------------------------ MatrixMarket ingest */

/* One pass over a memory-mapped file: no stdio, no per-entry locale work.  The grammar is
 * what fscanf("%d %d %d\n") accepts (sequential/lanczos_modp.c:239): optional blanks, optional
 * sign, decimal digits. */
typedef struct {
	const char *p, *end;
} cursor;

static int next_line(cursor *c, char *buf, size_t cap)
{
	if (c->p >= c->end)
		return -1;
	const char *nl = memchr(c->p, '\n', (size_t)(c->end - c->p));
	const char *stop = nl ? nl : c->end;
	size_t len = (size_t)(stop - c->p);
	if (len >= cap)
		len = cap - 1;
	memcpy(buf, c->p, len);
	buf[len] = 0;
	c->p = nl ? nl + 1 : c->end;
	return 0;
}

static inline int next_int(cursor *c, long long *out)
{
	const char *p = c->p, *end = c->end;
	while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f'))
		p++;
	if (p >= end)
		return -1;
	int neg = 0;
	if (*p == '-' || *p == '+') {
		neg = (*p == '-');
		p++;
	}
	if (p >= end || *p < '0' || *p > '9')
		return -1;
	unsigned long long acc = 0;
	while (p < end && *p >= '0' && *p <= '9')
		acc = acc * 10 + (unsigned)(*p++ - '0');
	*out = neg ? -(long long)acc : (long long)acc;
	c->p = p;
	return 0;
}

static inline int is_blank(char ch)
{
	return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f';
}

/* The entries of a large file, read by all cores: the text is cut into one piece per thread at token boundaries,
 * pass 1 counts the whitespace-separated tokens of each piece, pass 2 parses token g as field g % 3 of entry g / 3
 * (any layout fscanf("%d %d %d") accepts, not only one entry per line).  Returns 0 when every one of the 3*nz
 * tokens was a plain in-range integer; anything else returns 1 WITHOUT an error message and the caller re-reads
 * the file with the sequential reader, which defines the behaviour on irregular input. */
static int parse_entries_parallel(const char *p, const char *end, long long nr, long long nc, long long nz,
				  uint64_t prime, blz_coo *out)
{
	int T = omp_get_max_threads();
	if (T > 64)
		T = 64;
	const char *cut[65];
	long long first[65];
	cut[0] = p;
	cut[T] = end;
	for (int t = 1; t < T; t++) {
		const char *q = p + (size_t)(end - p) / (size_t)T * (size_t)t;
		while (q < end && !is_blank(*q))	/* finish the token the cut fell into */
			q++;
		cut[t] = q < cut[t - 1] ? cut[t - 1] : q;
	}
	int irregular = 0;
	first[0] = 0;
#pragma omp parallel for schedule(static, 1)
	for (int t = 0; t < T; t++) {
		long long cnt = 0;
		int in_tok = 0;
		for (const char *q = cut[t]; q < cut[t + 1]; q++) {
			const int bl = is_blank(*q);
			cnt += (!bl && !in_tok);
			in_tok = !bl;
		}
		first[t + 1] = cnt;
	}
	for (int k = 1; k <= T; k++)
		first[k] += first[k - 1];
	if (first[T] < 3 * nz)
		return 1;
#pragma omp parallel for schedule(static, 1) reduction(| : irregular)
	for (int t = 0; t < T; t++) {
		long long g = first[t];
		int bad = 0;
		const char *q = cut[t], *stop = cut[t + 1];
		while (!bad && g < 3 * nz) {
			while (q < stop && is_blank(*q))
				q++;
			if (q >= stop)
				break;
			int neg = 0;
			if (*q == '-' || *q == '+')
				neg = (*q++ == '-');
			if (q >= stop || *q < '0' || *q > '9') {
				bad = 1;
				break;
			}
			unsigned long long acc = 0;
			while (q < stop && *q >= '0' && *q <= '9')
				acc = acc * 10 + (unsigned)(*q++ - '0');
			if (q < stop && !is_blank(*q)) {
				bad = 1;
				break;
			}
			const long long val = neg ? -(long long)acc : (long long)acc, u = g / 3;
			switch (g % 3) {
			case 0:
				bad = val < 1 || val > nr;
				out->i[u] = (int32_t)(val - 1);
				break;
			case 1:
				bad = val < 1 || val > nc;
				out->j[u] = (int32_t)(val - 1);
				break;
			default:	/* sequential/lanczos_modp.c:238-243: "%d" into a u32, then % prime */
				out->x[u] = (uint32_t)((uint64_t)(uint32_t)(int32_t)val % prime);
			}
			g++;
		}
		irregular |= bad;
	}
	return irregular;
}

static void lowercase(char *s)
{
	for (; *s; s++)
		*s = (char)tolower((unsigned char)*s);
}

/* mm_read_banner(), mmio.c:28-111, and the type tests of sequential/lanczos_modp.c:214-221. */
static int check_banner(const char *line, int want_array)
{
	char tok[5][64];
	if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
		return blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
	for (int k = 1; k < 5; k++)
		lowercase(tok[k]);
	if (strncmp(tok[0], "%%MatrixMarket", 14) != 0 || strcmp(tok[1], "matrix") != 0)
		return blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
	const int sparse = strcmp(tok[2], "coordinate") == 0, dense = strcmp(tok[2], "array") == 0;
	const int known_type = !strcmp(tok[3], "real") || !strcmp(tok[3], "complex") || !strcmp(tok[3], "pattern")
	    || !strcmp(tok[3], "integer");
	const int known_sym = !strcmp(tok[4], "general") || !strcmp(tok[4], "symmetric")
	    || !strcmp(tok[4], "hermitian") || !strcmp(tok[4], "skew-symmetric");
	if ((!sparse && !dense) || !known_type || !known_sym)
		return blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
	if (want_array ? !dense : !sparse)
		return blz_fail(BLZ_EFORMAT, "Matrix Market type: [%s %s %s %s] not supported (only %s matrices are OK)",
				tok[1], tok[2], tok[3], tok[4], want_array ? "dense" : "sparse");
	if (strcmp(tok[3], "integer") != 0 || strcmp(tok[4], "general") != 0)
		return blz_fail(BLZ_EFORMAT, "Matrix type [%s %s %s %s] not supported (only integer general are OK)",
				tok[1], tok[2], tok[3], tok[4]);
	return BLZ_OK;
}

int blz_mm_load(const char *path, uint64_t prime, blz_coo *out)
{
	if (!path || !out || prime < 2)
		return blz_fail(BLZ_EINVAL, "blz_mm_load: bad argument");
	memset(out, 0, sizeof *out);
	int fd = open(path, O_RDONLY);
	if (fd < 0)
		return blz_fail(BLZ_EIO, "impossible d'ouvrir %s: %s", path, strerror(errno));
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size == 0) {
		close(fd);
		return blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
	}
	char *base = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (base == MAP_FAILED)
		return blz_fail(BLZ_EIO, "mmap %s: %s", path, strerror(errno));
	madvise(base, (size_t)st.st_size, MADV_SEQUENTIAL);
	cursor c = { base, base + st.st_size };
	char line[1100];
	int rc = BLZ_OK;
	if (next_line(&c, line, sizeof line) || (rc = check_banner(line, 0)) != BLZ_OK) {
		munmap(base, (size_t)st.st_size);
		return rc ? rc : blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
	}
	/* mm_read_mtx_crd_size(), mmio.c:113-141 */
	long long nr = 0, nc = 0, nz = 0;
	do {
		if (next_line(&c, line, sizeof line)) {
			munmap(base, (size_t)st.st_size);
			return blz_fail(BLZ_EIO, "Cannot read matrix size");
		}
	} while (line[0] == '%');
	if (sscanf(line, "%lld %lld %lld", &nr, &nc, &nz) != 3) {
		if (next_int(&c, &nr) || next_int(&c, &nc) || next_int(&c, &nz)) {
			munmap(base, (size_t)st.st_size);
			return blz_fail(BLZ_EIO, "Cannot read matrix size");
		}
	}
	if (nr < 0 || nc < 0 || nz < 0 || nr > INT32_MAX || nc > INT32_MAX) {
		munmap(base, (size_t)st.st_size);
		return blz_fail(BLZ_EIO, "Cannot read matrix size");
	}
	out->nrows = nr;
	out->ncols = nc;
	out->nnz = nz;
	const size_t cap = (size_t)(nz ? nz : 1);
	out->i = malloc(cap * sizeof *out->i);
	out->j = malloc(cap * sizeof *out->j);
	out->x = malloc(cap * sizeof *out->x);
	if (!out->i || !out->j || !out->x) {
		munmap(base, (size_t)st.st_size);
		blz_coo_free(out);
		return blz_fail(BLZ_ENOMEM, "Cannot allocate sparse matrix");
	}
	if (nz >= 200000 && parse_entries_parallel(c.p, c.end, nr, nc, nz, prime, out) == 0) {
		munmap(base, (size_t)st.st_size);
		return BLZ_OK;
	}
	/* small file, or something the fast path does not take (a malformed or out-of-range entry, tokens glued
	 * together): the one-token-at-a-time reader below is the definition, and it names the offending entry */
	for (long long u = 0; u < nz; u++) {
		long long a, b, v;
		if (next_int(&c, &a) || next_int(&c, &b) || next_int(&c, &v)) {
			munmap(base, (size_t)st.st_size);
			blz_coo_free(out);
			return blz_fail(BLZ_EIO, "parse error entry %lld", u);
		}
		if (a < 1 || a > nr || b < 1 || b > nc) {
			munmap(base, (size_t)st.st_size);
			blz_coo_free(out);
			return blz_fail(BLZ_EIO, "entry %lld: index (%lld, %lld) outside %lld x %lld", u, a, b, nr, nc);
		}
		out->i[u] = (int32_t)(a - 1);	/* MatrixMarket is 1-based, :241-242 */
		out->j[u] = (int32_t)(b - 1);
		/* :238-243: "%d" into a u32, then % prime */
		out->x[u] = (uint32_t)((uint64_t)(uint32_t)(int32_t)v % prime);
	}
	munmap(base, (size_t)st.st_size);
	return BLZ_OK;
}

int blz_mm_save_coo(const char *path, const blz_coo *M)
{
	if (!path || !M)
		return blz_fail(BLZ_EINVAL, "blz_mm_save_coo: bad argument");
	FILE *f = fopen(path, "w");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	static char iobuf[1 << 20];
	setvbuf(f, iobuf, _IOFBF, sizeof iobuf);
	fprintf(f, "%%%%MatrixMarket matrix coordinate integer general\n%ld %ld %ld\n", (long)M->nrows, (long)M->ncols,
		(long)M->nnz);
	for (int64_t k = 0; k < M->nnz; k++)
		fprintf(f, "%d %d %u\n", M->i[k] + 1, M->j[k] + 1, M->x[k]);
	if (fclose(f))
		return blz_fail(BLZ_EIO, "write error on %s", path);
	return BLZ_OK;
}

void blz_coo_free(blz_coo *M)
{
	if (!M)
		return;
	free(M->i);
	free(M->j);
	free(M->x);
	memset(M, 0, sizeof *M);
}

/* ------------------------------------------------------------------ synthetic matrices */

static inline uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

int blz_synth_coo(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern, uint64_t prime,
		  blz_coo *out)
{
	static const int32_t palette[7] = { 1, 1, 1, 2, 3, -1, -2 };
	if (!out || nrows <= 0 || ncols <= 0 || nnz < 0 || prime < 2 || nrows > INT32_MAX || ncols > INT32_MAX)
		return blz_fail(BLZ_EINVAL, "blz_synth_coo: bad shape");
	const int64_t base = nnz / nrows, extra = nnz % nrows;
	if (base + (extra ? 1 : 0) > ncols)
		return blz_fail(BLZ_EINVAL, "blz_synth_coo: more entries per row than columns");
	memset(out, 0, sizeof *out);
	out->nrows = nrows;
	out->ncols = ncols;
	out->nnz = nnz;
	const size_t cap = (size_t)(nnz ? nnz : 1);
	out->i = malloc(cap * sizeof *out->i);
	out->j = malloc(cap * sizeof *out->j);
	out->x = malloc(cap * sizeof *out->x);
	if (!out->i || !out->j || !out->x) {
		blz_coo_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_synth_coo: out of memory");
	}
#pragma omp parallel for schedule(static) if (nnz > 200000)
	for (int64_t r = 0; r < nrows; r++) {
		const int64_t cnt = base + (r < extra);
		int64_t at = r * base + (r < extra ? r : extra);
		uint64_t s = seed ^ ((uint64_t)r * 0xD1342543DE82EF95ull);
		for (int64_t k = 0; k < cnt; k++) {
			int32_t col;
			for (;;) {	/* distinct columns within the row */
				col = (int32_t)(splitmix64(&s) % (uint64_t)ncols);
				int dup = 0;
				for (int64_t q = at - k; q < at && !dup; q++)
					dup = (out->j[q] == col);
				if (!dup)
					break;
			}
			const int32_t v = pattern ? 1 : palette[splitmix64(&s) % 7];
			out->i[at] = (int32_t)r;
			out->j[at] = col;
			out->x[at] = (uint32_t)((uint64_t)(uint32_t)v % prime);
			at++;
		}
	}
	return BLZ_OK;
}

/*
 * The entries of blz_synth_coo's matrix (same shape, seed, values) that fall in rows [r0, r1) AND columns [c0, c1), with
 * their GLOBAL indices, rows ascending -- without ever holding the rest: every row is seeded by itself, so a rank of a
 * sharded solve can make its own rows (c0 = 0, c1 = ncols) and its own columns (r0 = 0, r1 = nrows) of config 5's
 * 50 M x 50 M / 2e9-entry matrix from 1/8 of the memory.  The reference's MPI loader hands out pieces of
 * the file the same way (mpi/lanczos_modp.c:1841-1845).  One pass: every thread keeps the wanted entries of a
 * contiguous range of rows in its own buffers, which are then laid end to end.
 */
int blz_synth_coo_part(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern, uint64_t prime,
		       int64_t r0, int64_t r1, int64_t c0, int64_t c1, blz_coo *out)
{
	static const int32_t palette[7] = { 1, 1, 1, 2, 3, -1, -2 };
	if (!out || nrows <= 0 || ncols <= 0 || nnz < 0 || prime < 2 || nrows > INT32_MAX || ncols > INT32_MAX ||
	    r0 < 0 || r1 < r0 || r1 > nrows || c0 < 0 || c1 < c0 || c1 > ncols)
		return blz_fail(BLZ_EINVAL, "blz_synth_coo_part: bad shape or range");
	const int64_t base = nnz / nrows, extra = nnz % nrows;
	if (base + (extra ? 1 : 0) > ncols || base + 1 > 4096)
		return blz_fail(BLZ_EINVAL, "blz_synth_coo_part: rows too long");
	memset(out, 0, sizeof *out);
	out->nrows = nrows;
	out->ncols = ncols;
	const int nth = omp_get_max_threads();
	int32_t **bi = calloc((size_t)nth, sizeof *bi), **bj = calloc((size_t)nth, sizeof *bj);
	uint32_t **bx = calloc((size_t)nth, sizeof *bx);
	int64_t *cnt = calloc((size_t)nth + 1, sizeof *cnt);
	int fail = !bi || !bj || !bx || !cnt;
	const int all_cols = c0 == 0 && c1 == ncols;
	if (!fail) {
#pragma omp parallel num_threads(nth)
		{
			const int t = omp_get_thread_num(), T = omp_get_num_threads();
			const int64_t span = r1 - r0, a = r0 + span * t / T, b = r0 + span * (t + 1) / T;
			/* expected share, with room; grown when it runs out */
			int64_t cap = (int64_t)((double)(b - a) * (double)(base + 1) * ((double)(c1 - c0) / (double)ncols) * 1.05) + 1024;
			int32_t *ti = malloc(sizeof *ti * (size_t)cap), *tj = malloc(sizeof *tj * (size_t)cap);
			uint32_t *tx = malloc(sizeof *tx * (size_t)cap);
			int64_t have = 0;
			int bad = !ti || !tj || !tx;
			int32_t row[4097];
			for (int64_t r = a; r < b && !bad; r++) {
				const int64_t n_r = base + (r < extra);
				uint64_t sd = seed ^ ((uint64_t)r * 0xD1342543DE82EF95ull);
				if (have + n_r > cap) {
					cap = cap + cap / 4 + n_r;
					int32_t *ni = realloc(ti, sizeof *ti * (size_t)cap), *nj = realloc(tj, sizeof *tj * (size_t)cap);
					uint32_t *nx = realloc(tx, sizeof *tx * (size_t)cap);
					ti = ni ? ni : ti;
					tj = nj ? nj : tj;
					tx = nx ? nx : tx;
					if (!ni || !nj || !nx) {
						bad = 1;
						break;
					}
				}
				for (int64_t k = 0; k < n_r; k++) {
					int32_t col;
					for (;;) {	/* distinct columns within the row: the draws of blz_synth_coo, in its order */
						col = (int32_t)(splitmix64(&sd) % (uint64_t)ncols);
						int dup = 0;
						for (int64_t q = 0; q < k && !dup; q++)
							dup = (row[q] == col);
						if (!dup)
							break;
					}
					row[k] = col;
					const int32_t v = pattern ? 1 : palette[splitmix64(&sd) % 7];
					if (all_cols || (col >= c0 && col < c1)) {
						ti[have] = (int32_t)r;
						tj[have] = col;
						tx[have] = (uint32_t)((uint64_t)(uint32_t)v % prime);
						have++;
					}
				}
			}
			bi[t] = ti;
			bj[t] = tj;
			bx[t] = tx;
			cnt[t + 1] = have;
			if (bad) {
#pragma omp atomic write
				fail = 1;
			}
		}
	}
	int rc = BLZ_OK;
	if (fail) {
		rc = blz_fail(BLZ_ENOMEM, "blz_synth_coo_part: out of memory");
	} else {
		for (int t = 0; t < nth; t++)
			cnt[t + 1] += cnt[t];
		const int64_t total = cnt[nth];
		out->nnz = total;
		out->i = malloc(sizeof *out->i * (size_t)(total ? total : 1));
		out->j = malloc(sizeof *out->j * (size_t)(total ? total : 1));
		out->x = malloc(sizeof *out->x * (size_t)(total ? total : 1));
		if (!out->i || !out->j || !out->x) {
			blz_coo_free(out);
			rc = blz_fail(BLZ_ENOMEM, "blz_synth_coo_part: out of memory");
		} else {
#pragma omp parallel for schedule(static, 1)
			for (int t = 0; t < nth; t++) {
				const size_t m = (size_t)(cnt[t + 1] - cnt[t]);
				if (m && bi[t]) {
					memcpy(out->i + cnt[t], bi[t], m * sizeof *out->i);
					memcpy(out->j + cnt[t], bj[t], m * sizeof *out->j);
					memcpy(out->x + cnt[t], bx[t], m * sizeof *out->x);
				}
			}
		}
	}
	for (int t = 0; t < nth; t++) {
		if (bi) free(bi[t]);
		if (bj) free(bj[t]);
		if (bx) free(bx[t]);
	}
	free(bi);
	free(bj);
	free(bx);
	free(cnt);
	return rc;
}

/*
 * A matrix WITH structure, for the measurements the uniform stand-ins cannot make (it is never the headline workload):
 * the shape of a sieve relation matrix.  Row r gets floor(nnz/R) (+1) distinct columns,
 *   hot_pct  % drawn with probability ~ 1/(c + 16): heavy-tailed column degrees (small primes),
 *   band_pct % uniform in a band of `band` columns centred on r*C/R: correlated supports of neighbouring rows,
 *   the rest uniform over all columns;
 * the first column numbers are the dense ones, as in a file sorted by prime.  Values as blz_synth_coo.
 */
int blz_synth_structured(int64_t nrows, int64_t ncols, int64_t nnz, uint64_t seed, int pattern, uint64_t prime,
			 int hot_pct, int band_pct, int64_t band, blz_coo *out)
{
	static const int32_t palette[7] = { 1, 1, 1, 2, 3, -1, -2 };
	if (!out || nrows <= 0 || ncols <= 0 || nnz < 0 || prime < 2 || nrows > INT32_MAX || ncols > INT32_MAX ||
	    hot_pct < 0 || band_pct < 0 || hot_pct + band_pct > 100 || band < 1)
		return blz_fail(BLZ_EINVAL, "blz_synth_structured: bad argument");
	const int64_t base = nnz / nrows, extra = nnz % nrows;
	if (2 * (base + 1) > ncols || 2 * (base + 1) > band)
		return blz_fail(BLZ_EINVAL, "blz_synth_structured: rows too long for the column range / band");
	memset(out, 0, sizeof *out);
	out->nrows = nrows;
	out->ncols = ncols;
	out->nnz = nnz;
	const size_t cap = (size_t)(nnz ? nnz : 1);
	out->i = malloc(cap * sizeof *out->i);
	out->j = malloc(cap * sizeof *out->j);
	out->x = malloc(cap * sizeof *out->x);
	if (!out->i || !out->j || !out->x) {
		blz_coo_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_synth_structured: out of memory");
	}
	const double c0 = 16.0, lnr = log(((double)ncols + c0) / c0);
	if (band > ncols)
		band = ncols;
#pragma omp parallel for schedule(static) if (nnz > 200000)
	for (int64_t r = 0; r < nrows; r++) {
		const int64_t cnt = base + (r < extra);
		int64_t at = r * base + (r < extra ? r : extra);
		uint64_t s = seed ^ ((uint64_t)r * 0xD1342543DE82EF95ull);
		int64_t lo = (int64_t)((double)r * (double)ncols / (double)nrows) - band / 2;
		lo = lo < 0 ? 0 : (lo + band > ncols ? ncols - band : lo);
		for (int64_t k = 0; k < cnt; k++) {
			const int kind = (int)(splitmix64(&s) % 100);
			int32_t col;
			for (int tries = 0;; tries++) {	/* distinct columns within the row */
				const uint64_t u = splitmix64(&s);
				if (kind < hot_pct && tries < 8) {
					const double x = c0 * (exp((double)(u >> 11) * (1.0 / 9007199254740992.0) * lnr) - 1.0);
					col = (int32_t)(x >= (double)ncols ? ncols - 1 : (int64_t)x);
				} else if (kind < hot_pct + band_pct && tries < 16) {
					col = (int32_t)(lo + (int64_t)(u % (uint64_t)band));
				} else {
					col = (int32_t)(u % (uint64_t)ncols);
				}
				int dup = 0;
				for (int64_t q = at - k; q < at && !dup; q++)
					dup = (out->j[q] == col);
				if (!dup)
					break;
			}
			const int32_t v = pattern ? 1 : palette[splitmix64(&s) % 7];
			out->i[at] = (int32_t)r;
			out->j[at] = col;
			out->x[at] = (uint32_t)((uint64_t)(uint32_t)v % prime);
			at++;
		}
	}
	return BLZ_OK;
}

/* ------------------------------------------------------------------------ CSR building */

void blz_csr_free(blz_csr *A)
{
	if (!A)
		return;
	free(A->row_ptr);
	free(A->col_idx);
	free(A->val);
	memset(A, 0, sizeof *A);
}

/* Counting sort by row (or by column for the transpose); stable, so entries of one row keep
 * file order.  Order is irrelevant to the result: every output word is the canonical residue
 * of an exact integer sum. */
/* rows [first, first + count) only (every entry of M must lie in them): the CSR's row q is row first + q */
static int csr_from_coo_window(const blz_coo *M, int transpose, int pattern, int64_t first, int64_t count, blz_csr *out);

int blz_csr_from_coo(const blz_coo *M, int transpose, int pattern, blz_csr *out)
{
	if (!M || !out)
		return blz_fail(BLZ_EINVAL, "blz_csr_from_coo: bad argument");
	return csr_from_coo_window(M, transpose, pattern, 0, transpose ? M->ncols : M->nrows, out);
}

static int csr_from_coo_window(const blz_coo *M, int transpose, int pattern, int64_t first, int64_t count, blz_csr *out)
{
	if (M->nnz >= (int64_t)UINT32_MAX)
		return blz_fail(BLZ_EINVAL, "blz_csr_from_coo: nnz >= 2^32 per slab is not supported");
	memset(out, 0, sizeof *out);
	const int64_t rows = count;
	const int32_t *ri = transpose ? M->j : M->i, *ci = transpose ? M->i : M->j;
	{
		int64_t outside = 0;
#pragma omp parallel for schedule(static) reduction(+ : outside) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++)
			outside += (ri[k] < first || ri[k] >= first + count);
		if (outside)
			return blz_fail(BLZ_EINVAL, "blz_csr_from_coo: %lld entries lie outside rows [%lld, %lld)", (long long)outside,
					(long long)first, (long long)(first + count));
	}
	out->rows = rows;
	out->cols = transpose ? M->nrows : M->ncols;
	out->nnz = M->nnz;
	out->row_ptr = calloc((size_t)rows + 2, sizeof *out->row_ptr);
	out->col_idx = malloc((size_t)(M->nnz ? M->nnz : 1) * sizeof *out->col_idx);
	int ones = pattern != 0;
	if (ones) {
		int64_t not_one = 0;
#pragma omp parallel for schedule(static) reduction(+ : not_one) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++)
			not_one += (M->x[k] != 1);
		ones = (not_one == 0);
	}
	out->val = ones ? NULL : malloc((size_t)(M->nnz ? M->nnz : 1) * sizeof *out->val);
	if (!out->row_ptr || !out->col_idx || (!ones && !out->val)) {
		blz_csr_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_csr_from_coo: out of memory");
	}
	uint32_t *next = out->row_ptr + 1;	/* next[r] will become the start of row r */
	/* histogram, prefix sums, scatter -- the two passes over the entries run on all host cores (atomic counters;
	 * the order of the entries inside a row is then arbitrary, which no result depends on) */
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
	for (int64_t k = 0; k < M->nnz; k++) {
#pragma omp atomic update
		next[ri[k] - first + 1]++;
	}
	for (int64_t r = 0; r < rows; r++)
		next[r + 1] += next[r];		/* next[r] = start(r), next[rows] = nnz */
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
	for (int64_t k = 0; k < M->nnz; k++) {
		uint32_t at;
#pragma omp atomic capture
		at = next[ri[k] - first]++;
		out->col_idx[at] = ci[k];
		if (out->val)
			out->val[at] = M->x[k];
	}
	/* after the scatter next[r] = start(r+1) = row_ptr[r+1]; row_ptr[0] is already 0 */
	return BLZ_OK;
}

int blz_partition_rows(const blz_csr *A, int parts, int64_t *bounds)
{
	if (!A || parts < 1 || !bounds)
		return blz_fail(BLZ_EINVAL, "blz_partition_rows: bad argument");
	/* weight of a row = its entries + 1 (so that empty rows are spread too) */
	const double total = (double)A->nnz + (double)A->rows;
	int64_t r = 0;
	bounds[0] = 0;
	for (int g = 1; g < parts; g++) {
		const double target = total * g / parts;
		while (r < A->rows && (double)A->row_ptr[r + 1] + (double)(r + 1) <= target)
			r++;
		bounds[g] = r;
	}
	bounds[parts] = A->rows;
	return BLZ_OK;
}

int blz_csr_slab(const blz_csr *A, int64_t r0, int64_t r1, blz_csr *out)
{
	memset(out, 0, sizeof *out);
	const uint32_t lo = A->row_ptr[r0], hi = A->row_ptr[r1];
	out->rows = r1 - r0;
	out->cols = A->cols;
	out->nnz = (int64_t)hi - lo;
	out->row_ptr = malloc((size_t)(out->rows + 1) * sizeof *out->row_ptr);
	out->col_idx = malloc((size_t)(out->nnz ? out->nnz : 1) * sizeof *out->col_idx);
	out->val = A->val ? malloc((size_t)(out->nnz ? out->nnz : 1) * sizeof *out->val) : NULL;
	if (!out->row_ptr || !out->col_idx || (A->val && !out->val)) {
		blz_csr_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_csr_slab: out of memory");
	}
	for (int64_t r = 0; r <= out->rows; r++)
		out->row_ptr[r] = A->row_ptr[r0 + r] - lo;
	memcpy(out->col_idx, A->col_idx + lo, (size_t)out->nnz * sizeof *out->col_idx);
	if (A->val)
		memcpy(out->val, A->val + lo, (size_t)out->nnz * sizeof *out->val);
	return BLZ_OK;
}

void blz_remap_columns(blz_csr *A, const int64_t *bounds, int parts, int64_t piece, int chunks)
{
	if (parts == 1)
		return;
#pragma omp parallel for schedule(static) if (A->nnz > 200000)
	for (int64_t k = 0; k < A->nnz; k++) {
		const int64_t c = A->col_idx[k];
		int lo = 0, hi = parts - 1;	/* largest g with bounds[g] <= c */
		while (lo < hi) {
			const int mid = (lo + hi + 1) / 2;
			if (bounds[mid] <= c)
				lo = mid;
			else
				hi = mid - 1;
		}
		const int64_t q = c - bounds[lo];	/* row inside rank lo's slab */
		(void)chunks;
		A->col_idx[k] = (int32_t)((q / piece) * (parts * piece) + lo * piece + (q % piece));
	}
}

/* Entries of A whose column lies in [k*width, (k+1)*width) become out[k] (same rows): the column-chunked form the
 * pipelined exchange multiplies piece by piece. */
int blz_csr_split_columns(const blz_csr *A, int64_t width, int chunks, blz_csr *out)
{
	memset(out, 0, (size_t)chunks * sizeof *out);
	for (int k = 0; k < chunks; k++) {
		out[k].rows = A->rows;
		out[k].cols = A->cols;
		out[k].row_ptr = calloc((size_t)A->rows + 1, sizeof(uint32_t));
		if (!out[k].row_ptr)
			return blz_fail(BLZ_ENOMEM, "blz_csr_split_columns: out of memory");
	}
	for (int64_t r = 0; r < A->rows; r++)
		for (uint32_t e = A->row_ptr[r]; e < A->row_ptr[r + 1]; e++)
			out[A->col_idx[e] / width].row_ptr[r + 1]++;
	for (int k = 0; k < chunks; k++) {
		for (int64_t r = 0; r < A->rows; r++)
			out[k].row_ptr[r + 1] += out[k].row_ptr[r];
		out[k].nnz = out[k].row_ptr[A->rows];
		out[k].col_idx = malloc((size_t)(out[k].nnz ? out[k].nnz : 1) * sizeof(int32_t));
		out[k].val = A->val ? malloc((size_t)(out[k].nnz ? out[k].nnz : 1) * sizeof(uint32_t)) : NULL;
		if (!out[k].col_idx || (A->val && !out[k].val))
			return blz_fail(BLZ_ENOMEM, "blz_csr_split_columns: out of memory");
	}
	uint32_t *fill = calloc((size_t)chunks, sizeof *fill);
	for (int64_t r = 0; r < A->rows; r++) {
		for (int k = 0; k < chunks; k++)
			fill[k] = out[k].row_ptr[r];
		for (uint32_t e = A->row_ptr[r]; e < A->row_ptr[r + 1]; e++) {
			const int k = (int)(A->col_idx[e] / width);
			out[k].col_idx[fill[k]] = A->col_idx[e];
			if (A->val)
				out[k].val[fill[k]] = A->val[e];
			fill[k]++;
		}
	}
	free(fill);
	return BLZ_OK;
}


static inline void atomic_min_i32(int32_t *addr, int32_t v)
{
	int32_t cur = __atomic_load_n(addr, __ATOMIC_RELAXED);
	while (v < cur && !__atomic_compare_exchange_n(addr, &cur, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
		;
}

/* new_i[k] = row_perm[i[k]], new_j[k] = col_perm[j[k]] (all host cores) */
void blz_coo_relabel(const blz_coo *M, const int32_t *row_perm, const int32_t *col_perm, int32_t *new_i, int32_t *new_j)
{
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
	for (int64_t k = 0; k < M->nnz; k++) {
		new_i[k] = row_perm[M->i[k]];
		new_j[k] = col_perm[M->j[k]];
	}
}

/* stable counting sort of `count` items by key in [0, nkeys]: perm[item] = new position */
static int sort_by_key(const int32_t *key, int64_t count, int64_t nkeys, int32_t *perm)
{
	int64_t *start = calloc((size_t)nkeys + 2, sizeof *start);
	if (!start)
		return blz_fail(BLZ_ENOMEM, "blz_reorder: out of memory");
	for (int64_t k = 0; k < count; k++)
		start[key[k] + 1]++;
	for (int64_t q = 0; q <= nkeys; q++)
		start[q + 1] += start[q];
	for (int64_t k = 0; k < count; k++)
		perm[k] = (int32_t)start[key[k]]++;
	free(start);
	return BLZ_OK;
}

int blz_reorder(const blz_coo *M, int32_t *row_perm, int32_t *col_perm)
{
	if (!M || !row_perm || !col_perm)
		return blz_fail(BLZ_EINVAL, "blz_reorder: bad argument");
	int32_t *key = malloc(sizeof *key * (size_t)((M->nrows > M->ncols ? M->nrows : M->ncols) + 1));
	if (!key)
		return blz_fail(BLZ_ENOMEM, "blz_reorder: out of memory");
	/* rows by smallest column (rows without entries last) */
#pragma omp parallel for schedule(static) if (M->nrows > 200000)
	for (int64_t r = 0; r < M->nrows; r++)
		key[r] = (int32_t)M->ncols;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
	for (int64_t k = 0; k < M->nnz; k++)
		atomic_min_i32(&key[M->i[k]], M->j[k]);
	int rc = sort_by_key(key, M->nrows, M->ncols, row_perm);
	/* columns by smallest NEW row */
	if (rc == BLZ_OK) {
#pragma omp parallel for schedule(static) if (M->ncols > 200000)
		for (int64_t c = 0; c < M->ncols; c++)
			key[c] = (int32_t)M->nrows;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++)
			atomic_min_i32(&key[M->j[k]], row_perm[M->i[k]]);
		rc = sort_by_key(key, M->ncols, M->nrows, col_perm);
	}
	free(key);
	return rc;
}

/* The `want` indices of largest degree (ties: lowest index), by descending degree; returns how many were chosen and the
 * number of entries they hold.  Degrees are bucketed up to 65535; above that the order inside the top bucket is by index. */
static int64_t pick_hot(const int32_t *deg, int64_t count, int64_t want, int32_t *hot_list, int64_t *hot_nnz)
{
	*hot_nnz = 0;
	if (want <= 0 || count <= 0)
		return 0;
	if (want > count)
		want = count;
	int64_t *hist = calloc(65536, sizeof *hist);
	if (!hist)
		return 0;
	for (int64_t q = 0; q < count; q++)
		hist[deg[q] > 65535 ? 65535 : deg[q]]++;
	int64_t need = want;
	int thr = 65535;
	for (; thr > 0; thr--) {
		if (hist[thr] >= need)
			break;
		need -= hist[thr];
	}
	free(hist);
	/* everything above thr, and the first `need` indices at thr (thr = 0: entries-less indices are never hot) */
	int64_t got = 0;
	for (int64_t q = 0; q < count && got < want; q++) {
		const int d = deg[q] > 65535 ? 65535 : deg[q];
		if (d == 0)
			continue;
		if (d > thr || (d == thr && need-- > 0))
			hot_list[got++] = (int32_t)q;
	}
	/* descending degree (insertion sort: a panel holds at most a few thousand rows) */
	for (int64_t a = 1; a < got; a++) {
		const int32_t x = hot_list[a];
		int64_t b = a;
		while (b > 0 && deg[hot_list[b - 1]] < deg[x]) {
			hot_list[b] = hot_list[b - 1];
			b--;
		}
		hot_list[b] = x;
	}
	for (int64_t a = 0; a < got; a++)
		*hot_nnz += deg[hot_list[a]];
	return got;
}

/*
 * blz_reorder with the densest rows and columns numbered first: hot[0] rows and hot[1] columns (in: the most a panel can
 * hold; out: how many were taken, 0 when they hold less than min_share of the entries -- a uniform matrix gets the plain
 * locality order).  The first hot[.] block rows of an operand are what the SpMV keeps in LDS (k_spmv_panel); the other
 * rows are ordered by their smallest NON-hot column (a dense column would otherwise be everybody's smallest) and the
 * other columns by their smallest non-hot new row.
 */
int blz_reorder_hot(const blz_coo *M, int32_t *row_perm, int32_t *col_perm, int64_t hot[2], double min_share, double share[2])
{
	if (!M || !row_perm || !col_perm || !hot || !share)
		return blz_fail(BLZ_EINVAL, "blz_reorder_hot: bad argument");
	const int64_t N[2] = { M->nrows, M->ncols };
	const int64_t big = N[0] > N[1] ? N[0] : N[1];
	int32_t *deg = calloc((size_t)big + 1, sizeof *deg), *key = malloc(sizeof *key * (size_t)(big + 1));
	int32_t *list[2] = { NULL, NULL };
	unsigned char *is_hot[2] = { calloc((size_t)N[0] + 1, 1), calloc((size_t)N[1] + 1, 1) };
	int rc = BLZ_OK;
	if (!deg || !key || !is_hot[0] || !is_hot[1])
		rc = blz_fail(BLZ_ENOMEM, "blz_reorder_hot: out of memory");
	for (int sd = 0; sd < 2 && rc == BLZ_OK; sd++) {
		const int32_t *idx = sd == 0 ? M->i : M->j;
		memset(deg, 0, sizeof *deg * (size_t)(N[sd] + 1));
		for (int64_t k = 0; k < M->nnz; k++)
			deg[idx[k]]++;
		list[sd] = malloc(sizeof(int32_t) * (size_t)(hot[sd] > 0 ? hot[sd] : 1));
		if (!list[sd]) {
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_hot: out of memory");
			break;
		}
		int64_t held = 0;
		int64_t got = pick_hot(deg, N[sd], hot[sd], list[sd], &held);
		share[sd] = M->nnz > 0 ? (double)held / (double)M->nnz : 0.0;
		if (share[sd] < min_share)
			got = 0;
		hot[sd] = got;
		for (int64_t a = 0; a < got; a++)
			is_hot[sd][list[sd][a]] = 1;
	}
	if (rc == BLZ_OK) {
		/* rows: hot ones first (by degree), then by smallest non-hot column */
		for (int64_t r = 0; r < M->nrows; r++)
			key[r] = (int32_t)M->ncols;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++)
			if (!is_hot[1][M->j[k]])
				atomic_min_i32(&key[M->i[k]], M->j[k]);
		rc = sort_by_key(key, M->nrows, M->ncols, row_perm);	/* positions 0..R-1 among ALL rows */
	}
	if (rc == BLZ_OK && hot[0] > 0) {
		/* squeeze the hot rows out of that order and put them in front */
		int32_t *order = malloc(sizeof *order * (size_t)M->nrows);
		if (!order) {
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_hot: out of memory");
		} else {
			for (int64_t r = 0; r < M->nrows; r++)
				order[row_perm[r]] = (int32_t)r;
			int64_t at = hot[0];
			for (int64_t q = 0; q < M->nrows; q++)
				if (!is_hot[0][order[q]])
					row_perm[order[q]] = (int32_t)at++;
			for (int64_t a = 0; a < hot[0]; a++)
				row_perm[list[0][a]] = (int32_t)a;
			free(order);
		}
	}
	if (rc == BLZ_OK) {
		for (int64_t c = 0; c < M->ncols; c++)
			key[c] = (int32_t)M->nrows;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++)
			if (!is_hot[0][M->i[k]])
				atomic_min_i32(&key[M->j[k]], row_perm[M->i[k]]);
		rc = sort_by_key(key, M->ncols, M->nrows, col_perm);
	}
	if (rc == BLZ_OK && hot[1] > 0) {
		int32_t *order = malloc(sizeof *order * (size_t)M->ncols);
		if (!order) {
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_hot: out of memory");
		} else {
			for (int64_t c = 0; c < M->ncols; c++)
				order[col_perm[c]] = (int32_t)c;
			int64_t at = hot[1];
			for (int64_t q = 0; q < M->ncols; q++)
				if (!is_hot[1][order[q]])
					col_perm[order[q]] = (int32_t)at++;
			for (int64_t a = 0; a < hot[1]; a++)
				col_perm[list[1][a]] = (int32_t)a;
			free(order);
		}
	}
	free(deg);
	free(key);
	free(list[0]);
	free(list[1]);
	free(is_hot[0]);
	free(is_hot[1]);
	return rc;
}

/* ---- choosing the renumbering by what it does to the gathers (round 2) ----
 *
 * Three candidate orders of the non-hot rows / columns:
 *   SMALLEST   rows by smallest (non-hot) column, columns by smallest new row: blz_reorder's order; on a matrix without
 *              structure it makes neighbouring rows share the line of their first entry (-8 % line fills, round 1)
 *   IDENTITY   the file's order: a matrix that arrives banded or block-structured stays so
 *   BARYCENTRE rows by the mean of their columns, columns by the mean of their new rows: tidies a file that is only
 *              roughly in band order (one sweep; it does not recover a band from a random shuffle)
 * Each is scored on a sample of windows of WIN consecutive (new) rows of each product: the number of DISTINCT 128-byte
 * lines of the operand that the window's entries touch -- what an XCD's L2 has to fetch while its wavefronts walk such
 * a window.  The order with the fewest lines wins; locality[t] = lines / entries of product t under it (1 = every
 * entry its own line: nothing to reuse; the per-XCD row ranges of the SpMV are switched on below 0.85).
 */
enum { ORD_SMALLEST = 0, ORD_IDENTITY = 1, ORD_BARYCENTRE = 2, ORD_SWEEPS = 3, ORD_KINDS = 4 };
#define ORD_SWEEP_COUNT 4

static int cmp_i32(const void *a, const void *b)
{
	const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
	return x < y ? -1 : x > y;
}

/* lines touched / entries, summed over `samples` windows of `win` consecutive (new) rows of the product whose rows are
 * `own` and whose operand rows are `other`.  Two parallel passes over the entries (count, fill) collect the operand lines
 * of the sampled windows; each window is then sorted and its distinct lines counted.  No index of the whole matrix is built. */
static void score_product(int64_t n_own, const int32_t *own_idx, const int32_t *perm_own, const int32_t *other_idx,
			  const int32_t *perm_other, int64_t nnz, int line_shift, int64_t win, int samples,
			  const unsigned char *skip_other, double *lines_out, double *entries_out)
{
	*lines_out = 0.0;
	*entries_out = 0.0;
	if (n_own <= 0 || nnz <= 0)
		return;
	if (win > n_own)
		win = n_own;
	const int64_t nwin = (n_own + win - 1) / win;
	if (samples > nwin)
		samples = (int)nwin;
	int32_t *slot = malloc(sizeof *slot * (size_t)nwin);
	int64_t *cnt = calloc((size_t)samples + 1, sizeof *cnt), *pos = calloc((size_t)samples + 1, sizeof *pos);
	if (!slot || !cnt || !pos) {
		free(slot);
		free(cnt);
		free(pos);
		return;
	}
	for (int64_t w = 0; w < nwin; w++)
		slot[w] = -1;
	for (int sidx = 0; sidx < samples; sidx++)
		slot[nwin * sidx / samples] = sidx;
#pragma omp parallel for schedule(static) if (nnz > 200000)
	for (int64_t k = 0; k < nnz; k++) {
		const int sl = slot[perm_own[own_idx[k]] / win];
		if (sl >= 0 && !(skip_other && skip_other[other_idx[k]])) {
#pragma omp atomic
			cnt[sl + 1]++;
		}
	}
	for (int sidx = 0; sidx < samples; sidx++)
		cnt[sidx + 1] += cnt[sidx];
	int32_t *buf = malloc(sizeof *buf * (size_t)(cnt[samples] ? cnt[samples] : 1));
	if (buf) {
#pragma omp parallel for schedule(static) if (nnz > 200000)
		for (int64_t k = 0; k < nnz; k++) {
			const int sl = slot[perm_own[own_idx[k]] / win];
			if (sl >= 0 && !(skip_other && skip_other[other_idx[k]])) {
				int64_t at;
#pragma omp atomic capture
				at = pos[sl]++;
				buf[cnt[sl] + at] = perm_other[other_idx[k]] >> line_shift;
			}
		}
		double lines = 0.0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : lines)
		for (int sidx = 0; sidx < samples; sidx++) {
			int32_t *b = buf + cnt[sidx];
			const int64_t m = cnt[sidx + 1] - cnt[sidx];
			qsort(b, (size_t)m, sizeof *b, cmp_i32);
			int64_t distinct = 0;
			for (int64_t k = 0; k < m; k++)
				distinct += (k == 0 || b[k] != b[k - 1]);
			lines += (double)distinct;
		}
		*lines_out = lines;
		*entries_out = (double)cnt[samples];
		free(buf);
	}
	free(slot);
	free(cnt);
	free(pos);
}

/* positions of the non-hot items in key order behind the hot ones (list[0..nhot) by descending degree) */
static int finish_perm(const int32_t *key, int64_t count, int64_t nkeys, const unsigned char *is_hot, const int32_t *list,
		       int64_t nhot, int32_t *perm)
{
	int rc = sort_by_key(key, count, nkeys, perm);
	if (rc != BLZ_OK || nhot <= 0)
		return rc;
	int32_t *order = malloc(sizeof *order * (size_t)count);
	if (!order)
		return blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
	for (int64_t r = 0; r < count; r++)
		order[perm[r]] = (int32_t)r;
	int64_t at = nhot;
	for (int64_t q = 0; q < count; q++)
		if (!is_hot[order[q]])
			perm[order[q]] = (int32_t)at++;
	for (int64_t a = 0; a < nhot; a++)
		perm[list[a]] = (int32_t)a;
	free(order);
	return BLZ_OK;
}

/* one half-sweep of the barycentre heuristic: key[own] = mean over own's entries of pos[other] (entries whose other end is
 * hot do not count), then positions by ascending key behind the hot items */
static int barycentre_half(const blz_coo *M, int own_is_row, const int32_t *pos_other, int64_t n_own, int64_t n_other,
			   unsigned char *const is_hot[2], int32_t *const list[2], const int64_t hot[2], double *sum, int32_t *cnt,
			   int32_t *key, int32_t *perm_own)
{
	const int32_t *own = own_is_row ? M->i : M->j, *oth = own_is_row ? M->j : M->i;
	const unsigned char *other_hot = is_hot[own_is_row ? 1 : 0];
	memset(sum, 0, sizeof *sum * (size_t)n_own);
	memset(cnt, 0, sizeof *cnt * (size_t)n_own);
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
	for (int64_t k = 0; k < M->nnz; k++)
		if (!other_hot[oth[k]]) {
#pragma omp atomic
			sum[own[k]] += (double)pos_other[oth[k]];
#pragma omp atomic
			cnt[own[k]]++;
		}
	for (int64_t r = 0; r < n_own; r++)
		key[r] = cnt[r] ? (int32_t)(sum[r] / cnt[r]) : (int32_t)n_other;
	const int sd = own_is_row ? 0 : 1;
	return finish_perm(key, n_own, n_other, is_hot[sd], list[sd], hot[sd], perm_own);
}

/* kind ORD_SWEEPS starts from the column positions in col_perm (the best earlier candidate) and overwrites both */
static int make_order(const blz_coo *M, int kind, unsigned char *const is_hot[2], int32_t *const list[2], const int64_t hot[2],
		      int32_t *row_perm, int32_t *col_perm)
{
	const int64_t big = (M->nrows > M->ncols ? M->nrows : M->ncols) + 1;
	int32_t *key = malloc(sizeof *key * (size_t)big);
	double *sum = NULL;
	int32_t *cnt = NULL;
	int rc = key ? BLZ_OK : blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
	if (rc == BLZ_OK && kind == ORD_SWEEPS) {
		/* Iterated barycentre sweeps (the classic ordering heuristic for a bipartite graph: rows to the mean position of
		 * their columns, columns to the mean position of their rows, a few times over): the stronger-than-one-pass
		 * candidate SURVEY 8(f)4 and the round-2 verdict ask for.  It is scored like the others and kept only if it
		 * leaves fewer lines to fetch. */
		sum = malloc(sizeof *sum * (size_t)big);
		cnt = malloc(sizeof *cnt * (size_t)big);
		if (!sum || !cnt)
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
		for (int sw = 0; sw < ORD_SWEEP_COUNT && rc == BLZ_OK; sw++) {
			rc = barycentre_half(M, 1, col_perm, M->nrows, M->ncols, is_hot, list, hot, sum, cnt, key, row_perm);
			if (rc == BLZ_OK)
				rc = barycentre_half(M, 0, row_perm, M->ncols, M->nrows, is_hot, list, hot, sum, cnt, key, col_perm);
		}
		free(key);
		free(sum);
		free(cnt);
		return rc;
	}
	if (rc == BLZ_OK && kind == ORD_BARYCENTRE) {
		sum = malloc(sizeof *sum * (size_t)big);
		cnt = malloc(sizeof *cnt * (size_t)big);
		if (!sum || !cnt)
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
	}
	if (rc == BLZ_OK) {
		if (kind == ORD_IDENTITY) {
			for (int64_t r = 0; r < M->nrows; r++)
				key[r] = (int32_t)r;
			rc = finish_perm(key, M->nrows, M->nrows, is_hot[0], list[0], hot[0], row_perm);
		} else if (kind == ORD_SMALLEST) {
			for (int64_t r = 0; r < M->nrows; r++)
				key[r] = (int32_t)M->ncols;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
			for (int64_t k = 0; k < M->nnz; k++)
				if (!is_hot[1][M->j[k]])
					atomic_min_i32(&key[M->i[k]], M->j[k]);
			rc = finish_perm(key, M->nrows, M->ncols, is_hot[0], list[0], hot[0], row_perm);
		} else {
			memset(sum, 0, sizeof *sum * (size_t)M->nrows);
			memset(cnt, 0, sizeof *cnt * (size_t)M->nrows);
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
			for (int64_t k = 0; k < M->nnz; k++)
				if (!is_hot[1][M->j[k]]) {
#pragma omp atomic
					sum[M->i[k]] += (double)M->j[k];
#pragma omp atomic
					cnt[M->i[k]]++;
				}
			for (int64_t r = 0; r < M->nrows; r++)
				key[r] = cnt[r] ? (int32_t)(sum[r] / cnt[r]) : (int32_t)M->ncols;
			rc = finish_perm(key, M->nrows, M->ncols, is_hot[0], list[0], hot[0], row_perm);
		}
	}
	if (rc == BLZ_OK) {
		if (kind == ORD_IDENTITY) {
			for (int64_t c = 0; c < M->ncols; c++)
				key[c] = (int32_t)c;
			rc = finish_perm(key, M->ncols, M->ncols, is_hot[1], list[1], hot[1], col_perm);
		} else if (kind == ORD_SMALLEST) {
			for (int64_t c = 0; c < M->ncols; c++)
				key[c] = (int32_t)M->nrows;
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
			for (int64_t k = 0; k < M->nnz; k++)
				if (!is_hot[0][M->i[k]])
					atomic_min_i32(&key[M->j[k]], row_perm[M->i[k]]);
			rc = finish_perm(key, M->ncols, M->nrows, is_hot[1], list[1], hot[1], col_perm);
		} else {
			memset(sum, 0, sizeof *sum * (size_t)M->ncols);
			memset(cnt, 0, sizeof *cnt * (size_t)M->ncols);
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
			for (int64_t k = 0; k < M->nnz; k++)
				if (!is_hot[0][M->i[k]]) {
#pragma omp atomic
					sum[M->j[k]] += (double)row_perm[M->i[k]];
#pragma omp atomic
					cnt[M->j[k]]++;
				}
			for (int64_t c = 0; c < M->ncols; c++)
				key[c] = cnt[c] ? (int32_t)(sum[c] / cnt[c]) : (int32_t)M->nrows;
			rc = finish_perm(key, M->ncols, M->nrows, is_hot[1], list[1], hot[1], col_perm);
		}
	}
	free(key);
	free(sum);
	free(cnt);
	return rc;
}

int blz_reorder_auto(const blz_coo *M, int32_t *row_perm, int32_t *col_perm, int64_t hot[2], double min_share,
		     double share[2], int rows_per_line, double locality[2], int *kind_out)
{
	if (!M || !row_perm || !col_perm || !hot || !share || !locality || rows_per_line < 1)
		return blz_fail(BLZ_EINVAL, "blz_reorder_auto: bad argument");
	locality[0] = locality[1] = 1.0;
	if (kind_out)
		*kind_out = ORD_SMALLEST;
	if (M->nnz >= INT32_MAX || M->nnz == 0) {	/* entry numbers must fit an int32 here: keep round 1's order */
		return blz_reorder_hot(M, row_perm, col_perm, hot, min_share, share);
	}
	const int64_t N[2] = { M->nrows, M->ncols };
	const int64_t big = N[0] > N[1] ? N[0] : N[1];
	int line_shift = 0;
	while ((1 << (line_shift + 1)) <= rows_per_line)
		line_shift++;
	int32_t *deg = calloc((size_t)big + 1, sizeof *deg);
	int32_t *list[2] = { NULL, NULL };
	unsigned char *is_hot[2] = { calloc((size_t)N[0] + 1, 1), calloc((size_t)N[1] + 1, 1) };
	int32_t *cand_r = malloc(sizeof *cand_r * (size_t)(N[0] ? N[0] : 1)), *cand_c = malloc(sizeof *cand_c * (size_t)(N[1] ? N[1] : 1));
	int rc = (deg && is_hot[0] && is_hot[1] && cand_r && cand_c) ? BLZ_OK : blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
	for (int sd = 0; sd < 2 && rc == BLZ_OK; sd++) {
		list[sd] = malloc(sizeof(int32_t) * (size_t)(hot[sd] > 0 ? hot[sd] : 1));
		if (!list[sd]) {
			rc = blz_fail(BLZ_ENOMEM, "blz_reorder_auto: out of memory");
			break;
		}
		share[sd] = 0.0;
		if (hot[sd] <= 0) {		/* no panel asked for (several ranks): no degrees needed */
			hot[sd] = 0;
			continue;
		}
		const int32_t *idx = sd == 0 ? M->i : M->j;
		memset(deg, 0, sizeof *deg * (size_t)(N[sd] + 1));
#pragma omp parallel for schedule(static) if (M->nnz > 200000)
		for (int64_t k = 0; k < M->nnz; k++) {
#pragma omp atomic
			deg[idx[k]]++;
		}
		int64_t held = 0;
		int64_t got = pick_hot(deg, N[sd], hot[sd], list[sd], &held);
		share[sd] = (double)held / (double)M->nnz;
		if (share[sd] < min_share)
			got = 0;
		hot[sd] = got;
		for (int64_t a = 0; a < got; a++)
			is_hot[sd][list[sd][a]] = 1;
	}
	/* round 1's order and the file's own are always compared; the mean order only when the file's order already beats
	 * round 1's by 10 % (there is structure to tidy) and the matrix is not huge (each candidate is two passes over it) */
	double best = -1.0, score[ORD_KINDS] = { -1.0, -1.0, -1.0, -1.0 };
	const char *verbose = getenv("BLZ_REORDER_VERBOSE");
	const char *sw_env = getenv("BLZ_REORDER_SWEEPS");	/* 0: without the iterated barycentre sweeps (A/B) */
	const int sweeps_off = sw_env && sw_env[0] == '0';
	double best_frac = 1.0;		/* lines per entry of the best candidate so far, both products together */
	for (int kind = 0; kind < ORD_KINDS && rc == BLZ_OK; kind++) {
		/* the mean order and the sweeps only when the file's order already beats round 1's by 10 % (there is structure to
		 * tidy) and the matrix is not huge (each candidate is several passes over it) */
		if (kind == ORD_BARYCENTRE && !(score[ORD_IDENTITY] < 0.9 * score[ORD_SMALLEST] && M->nnz < 500000000))
			continue;
		/* the sweeps when an earlier candidate has found ANY locality (< 0.85 lines per entry over both products: the uniform
		 * stand-ins sit at 0.89 under round 1's order and skip them) and the matrix is not huge.  Round 3, profiles/r03_reorder_sweeps.txt: a band
		 * matrix whose rows and columns were scrambled goes from 0.72 / 0.45 (best earlier candidate) to 0.44 / 0.44 lines
		 * per entry; the structured workload keeps its file order (0.55 / 0.23 against 0.80 / 0.95 for the sweeps: 30 % of
		 * its entries are uniform and drag every mean to the middle). */
		if (kind == ORD_SWEEPS && (sweeps_off || best_frac >= 0.85 || M->nnz >= 500000000))
			continue;
		if (kind == ORD_SWEEPS)		/* start from the best candidate so far */
			memcpy(cand_c, col_perm, sizeof *cand_c * (size_t)N[1]);
		if ((rc = make_order(M, kind, is_hot, list, hot, cand_r, cand_c)) != BLZ_OK)
			break;
		double ln[2], en[2];
		/* product 0: rows of M gather block rows by column; product 1: rows of M^T (columns of M) gather by row */
		score_product(N[0], M->i, cand_r, M->j, cand_c, M->nnz, line_shift, 4096, 32, is_hot[1], &ln[0], &en[0]);
		score_product(N[1], M->j, cand_c, M->i, cand_r, M->nnz, line_shift, 4096, 32, is_hot[0], &ln[1], &en[1]);
		const double tot = ln[0] + ln[1];
		score[kind] = tot;
		if (verbose && verbose[0] == '1')
			fprintf(stderr, "blz_reorder_auto: order %d: %.3f / %.3f lines per entry (M x / M^T x)\n", kind,
				en[0] > 0.0 ? ln[0] / en[0] : 1.0, en[1] > 0.0 ? ln[1] / en[1] : 1.0);
		if (best < 0.0 || tot < best * 0.995) {	/* a later candidate must win by more than the sampling noise */
			best = tot;
			best_frac = en[0] + en[1] > 0.0 ? tot / (en[0] + en[1]) : 1.0;
			memcpy(row_perm, cand_r, sizeof *cand_r * (size_t)N[0]);
			memcpy(col_perm, cand_c, sizeof *cand_c * (size_t)N[1]);
			locality[0] = en[0] > 0.0 ? ln[0] / en[0] : 1.0;
			locality[1] = en[1] > 0.0 ? ln[1] / en[1] : 1.0;
			if (kind_out)
				*kind_out = kind;
		}
	}
	free(deg);
	free(cand_r);
	free(cand_c);
	for (int sd = 0; sd < 2; sd++) {
		free(list[sd]);
		free(is_hot[sd]);
	}
	return rc;
}

/* Entries of every row in ascending column order (values follow): with the dense columns numbered first, a row's
 * panel entries then come before its gathered ones, so the batches of a lane group are of one kind. */
void blz_csr_sort_rows(blz_csr *A)
{
#pragma omp parallel for schedule(dynamic, 4096) if (A->nnz > 200000)
	for (int64_t r = 0; r < A->rows; r++) {
		const uint32_t k0 = A->row_ptr[r], k1 = A->row_ptr[r + 1];
		for (uint32_t a = k0 + 1; a < k1; a++) {	/* insertion sort: rows are short or nearly sorted */
			const int32_t c = A->col_idx[a];
			const uint32_t v = A->val ? A->val[a] : 0;
			uint32_t b = a;
			while (b > k0 && A->col_idx[b - 1] > c) {
				A->col_idx[b] = A->col_idx[b - 1];
				if (A->val)
					A->val[b] = A->val[b - 1];
				b--;
			}
			A->col_idx[b] = c;
			if (A->val)
				A->val[b] = v;
		}
	}
}

int blz_shard_matrix(const blz_coo *M, int right, int rank, int nranks, int chunks, blz_csr slabs[2], int64_t *bounds0,
		     int64_t *bounds1, int64_t stride[2])
{
	if (!M || !slabs || !bounds0 || !bounds1 || !stride || nranks < 1 || rank < 0 || rank >= nranks || chunks < 1)
		return blz_fail(BLZ_EINVAL, "blz_shard_matrix: bad argument");
	memset(slabs, 0, 2 * sizeof slabs[0]);
	blz_csr full[2];
	int rc;
	if ((rc = blz_csr_from_coo(M, 0, 1, &full[0])) != BLZ_OK)
		return rc;
	if ((rc = blz_csr_from_coo(M, 1, 1, &full[1])) != BLZ_OK) {
		blz_csr_free(&full[0]);
		return rc;
	}
	/* rows of M live on side 0 for a left kernel and on side 1 for a right kernel; M^T the other way */
	int64_t *bounds[2] = { bounds0, bounds1 };
	const int row_side[2] = { right ? 1 : 0, right ? 0 : 1 };
	for (int t = 0; t < 2; t++) {
		const int sd = row_side[t];
		blz_partition_rows(&full[t], nranks, bounds[sd]);
		int64_t mx = 0;
		for (int g = 0; g < nranks; g++)
			if (bounds[sd][g + 1] - bounds[sd][g] > mx)
				mx = bounds[sd][g + 1] - bounds[sd][g];
		stride[sd] = (mx + chunks - 1) / chunks * chunks;	/* slab rows, padded to a whole number of pieces */
	}
	for (int t = 0; t < 2 && rc == BLZ_OK; t++) {
		const int rs = row_side[t], cs = 1 - rs;
		if (nranks == 1) {
			slabs[t] = full[t];		/* hand the arrays over */
			memset(&full[t], 0, sizeof full[t]);
		} else {
			rc = blz_csr_slab(&full[t], bounds[rs][rank], bounds[rs][rank + 1], &slabs[t]);
			if (rc == BLZ_OK) {
				blz_remap_columns(&slabs[t], bounds[cs], nranks, stride[cs] / chunks, chunks);
				slabs[t].cols = stride[cs] * nranks;
			}
		}
	}
	blz_csr_free(&full[0]);
	blz_csr_free(&full[1]);
	if (rc != BLZ_OK) {
		blz_csr_free(&slabs[0]);
		blz_csr_free(&slabs[1]);
	}
	return rc;
}

/* ----------------------------------------------------- the prepared matrix: set-up done once, shared, cacheable */

/*
 * Everything blz_set_matrix needs that does NOT depend on the rank: the renumbering, CSR(M) and CSR(M^T) in the
 * solver's numbering, the nnz-balanced row partition of both sides.  Round 1 redid all of it in every rank (and kept
 * 1/N of the result); now it is made once -- by the CLI's main thread for its G contexts, by rank 0 of a
 * multi-process job -- and each rank cuts its slab out of it (blz_prepared_slab).  Saved to a file it is the binary
 * cache SURVEY 8(f)1 asks for: the text parser is fast (0.5 s for 724 MB), what a second run saves is the renumbering
 * and the two CSR builds (GL7d19 shape ~1 s, config 5 ~48 s per rank).  A loaded cache is mmapped: N processes on one
 * node share its pages.
 */
#define PREP_MAGIC "BLZPREP3"

/* FNV-1a over 8-byte words, 1 MB pieces hashed in parallel and chained (the file hash of the CLI's cache key, and the
 * payload checksum of the cache itself) */
static uint64_t hash_bytes(const unsigned char *m, size_t len)
{
	const size_t piece = 1u << 20, np = (len + piece - 1) / piece;
	uint64_t *hp = malloc(sizeof *hp * (np ? np : 1));
	uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)len;
	if (!hp)
		return 0;
#pragma omp parallel for schedule(static)
	for (int64_t q = 0; q < (int64_t)np; q++) {
		const size_t lo = (size_t)q * piece, hi = lo + piece < len ? lo + piece : len;
		uint64_t x = 0xcbf29ce484222325ull;
		size_t k = lo;
		for (; k + 8 <= hi; k += 8) {
			uint64_t wd;
			memcpy(&wd, m + k, 8);
			x = (x ^ wd) * 0x100000001b3ull;
		}
		for (; k < hi; k++)
			x = (x ^ m[k]) * 0x100000001b3ull;
		hp[q] = x;
	}
	for (size_t q = 0; q < np; q++)
		h = (h ^ hp[q]) * 0x100000001b3ull;
	free(hp);
	return h ? h : 1;
}

typedef struct {
	char magic[8];
	uint64_t key;
	int64_t nrows, ncols, nnz;
	int32_t right, nranks, chunks, order_kind, has_perm, has_val[2];
	int64_t hot[2], stride[2];
	double share[2], locality[2];
	uint64_t off_perm[2], off_bounds[2], off_rp[2], off_ci[2], off_va[2];
	uint64_t total;
	uint64_t payload_hash;		/* hash_bytes of everything behind the header (bit rot, a torn or foreign file) */
} prep_header;

void blz_prepared_free(blz_prepared *P)
{
	if (!P)
		return;
	if (P->map) {
		munmap(P->map, P->map_len);
	} else {
		for (int q = 0; q < 2; q++) {
			free(P->perm[q]);
			free(P->bounds[q]);
			blz_csr_free(&P->full[q]);
		}
	}
	free(P);
}

int blz_prepare(const blz_coo *M, int right, int nranks, int chunks, int reorder, int rows_per_line, int64_t hot_cap,
		double min_share, blz_prepared **out)
{
	if (!M || !out || nranks < 1 || chunks < 1 || rows_per_line < 1)
		return blz_fail(BLZ_EINVAL, "blz_prepare: bad argument");
	*out = NULL;
	blz_prepared *P = calloc(1, sizeof *P);
	if (!P)
		return blz_fail(BLZ_ENOMEM, "blz_prepare: out of memory");
	P->nrows = M->nrows;
	P->ncols = M->ncols;
	P->nnz = M->nnz;
	P->right = right ? 1 : 0;
	P->nranks = nranks;
	P->chunks = chunks;
	P->only_rank = -1;
	P->locality[0] = P->locality[1] = 1.0;
	int rc = BLZ_OK;
	blz_coo R = *M;
	int32_t *ni = NULL, *nj = NULL;
	if (reorder && M->nnz > 0) {
		P->perm[0] = malloc(sizeof(int32_t) * (size_t)(M->nrows ? M->nrows : 1));
		P->perm[1] = malloc(sizeof(int32_t) * (size_t)(M->ncols ? M->ncols : 1));
		ni = malloc(sizeof *ni * (size_t)M->nnz);
		nj = malloc(sizeof *nj * (size_t)M->nnz);
		if (!P->perm[0] || !P->perm[1] || !ni || !nj)
			rc = blz_fail(BLZ_ENOMEM, "blz_prepare: out of memory");
		if (rc == BLZ_OK) {
			P->hot[0] = P->hot[1] = (nranks == 1 && chunks == 1) ? hot_cap : 0;
			if (reorder == 2)	/* round 1's order, no scored choice (A/B) */
				rc = blz_reorder_hot(M, P->perm[0], P->perm[1], P->hot, min_share, P->share);
			else
				rc = blz_reorder_auto(M, P->perm[0], P->perm[1], P->hot, min_share, P->share, rows_per_line,
						      P->locality, &P->order_kind);
		}
		if (rc == BLZ_OK) {
			P->has_perm = 1;
			blz_coo_relabel(M, P->perm[0], P->perm[1], ni, nj);
			R.i = ni;
			R.j = nj;
		}
	}
	for (int t = 0; t < 2 && rc == BLZ_OK; t++)
		rc = blz_csr_from_coo(&R, t, 1, &P->full[t]);
	free(ni);
	free(nj);
	/* rows of M live on side 0 for a left kernel and on side 1 for a right kernel; M^T the other way */
	for (int t = 0; t < 2 && rc == BLZ_OK; t++) {
		const int sd = t == 0 ? (right ? 1 : 0) : (right ? 0 : 1);
		P->bounds[sd] = malloc(sizeof(int64_t) * (size_t)(nranks + 1));
		if (!P->bounds[sd]) {
			rc = blz_fail(BLZ_ENOMEM, "blz_prepare: out of memory");
			break;
		}
		blz_partition_rows(&P->full[t], nranks, P->bounds[sd]);
		int64_t mx = 0;
		for (int g = 0; g < nranks; g++)
			if (P->bounds[sd][g + 1] - P->bounds[sd][g] > mx)
				mx = P->bounds[sd][g + 1] - P->bounds[sd][g];
		P->stride[sd] = (mx + chunks - 1) / chunks * chunks;	/* slab rows, padded to a whole number of pieces */
		/* product t gathers by the column index of full[t]: columns of M for t = 0 (hot[1]), rows of M for t = 1 */
		if (P->hot[t == 0 ? 1 : 0] > 0)
			blz_csr_sort_rows(&P->full[t]);
	}
	if (rc != BLZ_OK) {
		blz_prepared_free(P);
		return rc;
	}
	*out = P;
	return BLZ_OK;
}

/*
 * The prepared matrix of ONE rank, made from that rank's share alone: `row_part` = the entries of M in its rows
 * [row_bounds[rank], row_bounds[rank+1]), `col_part` = the entries in its columns [col_bounds[rank], ...), both with
 * global indices (blz_synth_coo_part makes them; a distributed loader would read them).  No process ever holds the
 * whole matrix: config 5 (2e9 entries, 24 GB of triplets) costs each of 8 ranks 2 x 2.5e8 entries.  The partition
 * is the caller's (bounds ascending from 0 to the dimension) and the numbering is kept -- a caller that wants the
 * locality renumbering applies it before cutting.  The reference's MPI variant distributes the matrix the same way
 * and never gathers it (mpi/lanczos_modp.c:1841-1900).  Not savable (blz_prepared_save refuses it).
 */
int blz_prepare_rank(const blz_coo *row_part, const blz_coo *col_part, int64_t nrows, int64_t ncols, int64_t nnz_total,
		     int right, int rank, int nranks, int chunks, const int64_t *row_bounds, const int64_t *col_bounds,
		     blz_prepared **out)
{
	if (!row_part || !col_part || !out || !row_bounds || !col_bounds || nranks < 1 || rank < 0 || rank >= nranks ||
	    chunks < 1 || nrows <= 0 || ncols <= 0 || nrows > INT32_MAX || ncols > INT32_MAX)
		return blz_fail(BLZ_EINVAL, "blz_prepare_rank: bad argument");
	*out = NULL;
	for (int g = 0; g < nranks; g++)
		if (row_bounds[g] > row_bounds[g + 1] || col_bounds[g] > col_bounds[g + 1])
			return blz_fail(BLZ_EINVAL, "blz_prepare_rank: bounds must ascend");
	if (row_bounds[0] != 0 || col_bounds[0] != 0 || row_bounds[nranks] != nrows || col_bounds[nranks] != ncols)
		return blz_fail(BLZ_EINVAL, "blz_prepare_rank: bounds must run from 0 to the dimension");
	blz_prepared *P = calloc(1, sizeof *P);
	if (!P)
		return blz_fail(BLZ_ENOMEM, "blz_prepare_rank: out of memory");
	P->nrows = nrows;
	P->ncols = ncols;
	P->nnz = nnz_total;
	P->right = right ? 1 : 0;
	P->nranks = nranks;
	P->chunks = chunks;
	P->only_rank = rank;
	P->locality[0] = P->locality[1] = 1.0;
	int rc = BLZ_OK;
	for (int t = 0; t < 2 && rc == BLZ_OK; t++) {
		const int sd = t == 0 ? (right ? 1 : 0) : (right ? 0 : 1);
		const int64_t *b = t == 0 ? row_bounds : col_bounds;
		P->bounds[sd] = malloc(sizeof(int64_t) * (size_t)(nranks + 1));
		if (!P->bounds[sd]) {
			rc = blz_fail(BLZ_ENOMEM, "blz_prepare_rank: out of memory");
			break;
		}
		memcpy(P->bounds[sd], b, sizeof(int64_t) * (size_t)(nranks + 1));
		int64_t mx = 0;
		for (int g = 0; g < nranks; g++)
			if (b[g + 1] - b[g] > mx)
				mx = b[g + 1] - b[g];
		P->stride[sd] = (mx + chunks - 1) / chunks * chunks;
		P->full_first[t] = b[rank];
		blz_coo part = t == 0 ? *row_part : *col_part;
		part.nrows = nrows;
		part.ncols = ncols;
		rc = csr_from_coo_window(&part, t, 1, b[rank], b[rank + 1] - b[rank], &P->full[t]);
	}
	if (rc != BLZ_OK) {
		blz_prepared_free(P);
		return rc;
	}
	*out = P;
	return BLZ_OK;
}

/* rank `rank`'s rows of M (t = 0) or M^T (t = 1), columns rewritten to positions in the gathered operand */
int blz_prepared_slab(const blz_prepared *P, int rank, int t, blz_csr *slab)
{
	if (!P || !slab || rank < 0 || rank >= P->nranks || t < 0 || t > 1)
		return blz_fail(BLZ_EINVAL, "blz_prepared_slab: bad argument");
	const int rs = t == 0 ? (P->right ? 1 : 0) : (P->right ? 0 : 1), cs = 1 - rs;
	if (P->only_rank >= 0 && rank != P->only_rank)
		return blz_fail(BLZ_EINVAL, "blz_prepared_slab: this object holds rank %d's rows only", P->only_rank);
	const int64_t off = P->only_rank >= 0 ? P->full_first[t] : 0;
	int rc = blz_csr_slab(&P->full[t], P->bounds[rs][rank] - off, P->bounds[rs][rank + 1] - off, slab);
	if (rc == BLZ_OK && P->nranks > 1) {
		blz_remap_columns(slab, P->bounds[cs], P->nranks, P->stride[cs] / P->chunks, P->chunks);
		slab->cols = P->stride[cs] * P->nranks;
	}
	return rc;
}

/*
 * Short-side exchange (tall or wide matrices, several ranks).  Product t of the iteration normally all-gathers its
 * operand -- the block on side cs -- and every rank multiplies its rows of the matrix by all of it.  When side cs is much
 * longer than the output side rs (relat9: 12.36 M rows against 0.55 M), that all-gather moves almost the whole long
 * block into every GPU.  The other way round costs a block of the SHORT side: rank g multiplies the transpose of ITS
 * rows of the other orientation, (full[1-t][own rows of side cs, :])^T, by its own slab of the operand -- no exchange
 * at all going in -- which gives a full-length partial product on side rs, and a reduce-scatter (u64 sums, then mod p)
 * hands every rank its rows of the sum.  mpi/lanczos_modp.c:1108-1124 is the reference's (hub-shaped) version of that
 * reduction.  This builds the matrix of that product: rows = the padded rank-major numbering of side rs
 * (nranks x stride[rs], so that equal-count reduce-scatter pieces are the ranks' slabs), columns = row numbers inside
 * this rank's slab of side cs.
 */
int blz_prepared_slab_short(const blz_prepared *P, int rank, int t, blz_csr *out)
{
	if (!P || !out || rank < 0 || rank >= P->nranks || t < 0 || t > 1)
		return blz_fail(BLZ_EINVAL, "blz_prepared_slab_short: bad argument");
	memset(out, 0, sizeof *out);
	const int rs = t == 0 ? (P->right ? 1 : 0) : (P->right ? 0 : 1), cs = 1 - rs;
	const blz_csr *O = &P->full[1 - t];		/* rows on side cs, columns on side rs */
	if (P->only_rank >= 0 && rank != P->only_rank)
		return blz_fail(BLZ_EINVAL, "blz_prepared_slab_short: this object holds rank %d's rows only", P->only_rank);
	const int64_t off = P->only_rank >= 0 ? P->full_first[1 - t] : 0;	/* O's row q is global row off + q */
	const int64_t b0 = P->bounds[cs][rank] - off, b1 = P->bounds[cs][rank + 1] - off;
	const uint32_t k0 = O->row_ptr[b0], k1 = O->row_ptr[b1];
	const int64_t nnz = (int64_t)k1 - k0, rows = (int64_t)P->nranks * P->stride[rs];
	if (rows > (int64_t)INT32_MAX)		/* padded output positions are kept in int32 (pos[] below) */
		return blz_fail(BLZ_EINVAL, "blz_prepared_slab_short: too many rows");
	out->rows = rows;
	out->cols = b1 - b0;
	out->nnz = nnz;
	out->row_ptr = calloc((size_t)rows + 2, sizeof *out->row_ptr);
	out->col_idx = malloc(sizeof *out->col_idx * (size_t)(nnz ? nnz : 1));
	out->val = O->val ? malloc(sizeof *out->val * (size_t)(nnz ? nnz : 1)) : NULL;
	int32_t *pos = malloc(sizeof *pos * (size_t)(nnz ? nnz : 1));
	if (!out->row_ptr || !out->col_idx || (O->val && !out->val) || !pos) {
		free(pos);
		blz_csr_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_prepared_slab_short: out of memory");
	}
	/* padded position of every entry's output row */
#pragma omp parallel for schedule(static) if (nnz > 200000)
	for (int64_t k = 0; k < nnz; k++) {
		const int64_t c = O->col_idx[k0 + k];
		int lo = 0, hi = P->nranks - 1;		/* largest g with bounds[rs][g] <= c */
		while (lo < hi) {
			const int mid = (lo + hi + 1) / 2;
			if (P->bounds[rs][mid] <= c)
				lo = mid;
			else
				hi = mid - 1;
		}
		pos[k] = (int32_t)((int64_t)lo * P->stride[rs] + (c - P->bounds[rs][lo]));
	}
	for (int64_t k = 0; k < nnz; k++)
		out->row_ptr[pos[k] + 1]++;
	for (int64_t r = 0; r < rows; r++)
		out->row_ptr[r + 1] += out->row_ptr[r];
	uint32_t *fill = malloc(sizeof *fill * (size_t)(rows + 1));
	if (!fill) {
		free(pos);
		blz_csr_free(out);
		return blz_fail(BLZ_ENOMEM, "blz_prepared_slab_short: out of memory");
	}
	memcpy(fill, out->row_ptr, sizeof *fill * (size_t)(rows + 1));
	for (int64_t r = b0; r < b1; r++)
		for (uint32_t k = O->row_ptr[r]; k < O->row_ptr[r + 1]; k++) {
			const uint32_t at = fill[pos[k - k0]]++;
			out->col_idx[at] = (int32_t)(r - b0);
			if (O->val)
				out->val[at] = O->val[k];
		}
	free(fill);
	free(pos);
	return BLZ_OK;
}

int blz_prepared_layout(const blz_prepared *P, int64_t *bounds0, int64_t *bounds1, int64_t stride[2])
{
	if (!P || !bounds0 || !bounds1 || !stride)
		return blz_fail(BLZ_EINVAL, "blz_prepared_layout: NULL argument");
	memcpy(bounds0, P->bounds[0], sizeof(int64_t) * (size_t)(P->nranks + 1));
	memcpy(bounds1, P->bounds[1], sizeof(int64_t) * (size_t)(P->nranks + 1));
	stride[0] = P->stride[0];
	stride[1] = P->stride[1];
	return BLZ_OK;
}

int blz_prepared_describe(const blz_prepared *P, int *right, int *nranks, int *chunks)
{
	if (!P)
		return blz_fail(BLZ_EINVAL, "blz_prepared_describe: NULL argument");
	if (right)
		*right = P->right;
	if (nranks)
		*nranks = P->nranks;
	if (chunks)
		*chunks = P->chunks;
	return BLZ_OK;
}

static uint64_t up64(uint64_t x) { return (x + 63u) & ~(uint64_t)63u; }

int blz_prepared_save(const blz_prepared *P, const char *path, uint64_t key)
{
	if (!P || !path)
		return blz_fail(BLZ_EINVAL, "blz_prepared_save: bad argument");
	if (P->only_rank >= 0)
		return blz_fail(BLZ_EINVAL, "blz_prepared_save: a one-rank object (blz_prepare_rank) is not a cache of the matrix");
	prep_header h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, PREP_MAGIC, 8);
	h.key = key;
	h.nrows = P->nrows;
	h.ncols = P->ncols;
	h.nnz = P->nnz;
	h.right = P->right;
	h.nranks = P->nranks;
	h.chunks = P->chunks;
	h.order_kind = P->order_kind;
	h.has_perm = P->has_perm;
	uint64_t at = up64(sizeof h);
	const int64_t plen[2] = { P->nrows, P->ncols };
	for (int q = 0; q < 2; q++) {
		h.hot[q] = P->hot[q];
		h.stride[q] = P->stride[q];
		h.share[q] = P->share[q];
		h.locality[q] = P->locality[q];
		h.has_val[q] = P->full[q].val != NULL;
		if (P->has_perm) {
			h.off_perm[q] = at;
			at = up64(at + sizeof(int32_t) * (uint64_t)plen[q]);
		}
		h.off_bounds[q] = at;
		at = up64(at + sizeof(int64_t) * (uint64_t)(P->nranks + 1));
		h.off_rp[q] = at;
		at = up64(at + sizeof(uint32_t) * (uint64_t)(P->full[q].rows + 1));
		h.off_ci[q] = at;
		at = up64(at + sizeof(int32_t) * (uint64_t)P->full[q].nnz);
		if (h.has_val[q]) {
			h.off_va[q] = at;
			at = up64(at + sizeof(uint32_t) * (uint64_t)P->full[q].nnz);
		}
	}
	h.total = at;
	char tmp[4096];
	snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
	const int fd = open(tmp, O_RDWR | O_CREAT | O_TRUNC, 0644);
	if (fd < 0)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", tmp, strerror(errno));
	if (ftruncate(fd, (off_t)h.total) != 0) {
		close(fd);
		unlink(tmp);
		return blz_fail(BLZ_EIO, "cannot size %s: %s", tmp, strerror(errno));
	}
	char *m = mmap(NULL, h.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (m == MAP_FAILED) {
		unlink(tmp);
		return blz_fail(BLZ_EIO, "cannot map %s: %s", tmp, strerror(errno));
	}
	for (int q = 0; q < 2; q++) {
		if (P->has_perm)
			memcpy(m + h.off_perm[q], P->perm[q], sizeof(int32_t) * (size_t)plen[q]);
		memcpy(m + h.off_bounds[q], P->bounds[q], sizeof(int64_t) * (size_t)(P->nranks + 1));
		memcpy(m + h.off_rp[q], P->full[q].row_ptr, sizeof(uint32_t) * (size_t)(P->full[q].rows + 1));
		memcpy(m + h.off_ci[q], P->full[q].col_idx, sizeof(int32_t) * (size_t)P->full[q].nnz);
		if (h.has_val[q])
			memcpy(m + h.off_va[q], P->full[q].val, sizeof(uint32_t) * (size_t)P->full[q].nnz);
	}
	h.payload_hash = hash_bytes((const unsigned char *)m + up64(sizeof h), (size_t)(h.total - up64(sizeof h)));
	memcpy(m, &h, sizeof h);
	const int bad = msync(m, h.total, MS_SYNC) != 0;
	munmap(m, h.total);
	if (bad || rename(tmp, path) != 0) {	/* atomic: a reader sees the old cache or the whole new one */
		unlink(tmp);
		return blz_fail(BLZ_EIO, "cannot write %s: %s", path, strerror(errno));
	}
	return BLZ_OK;
}

int blz_prepared_load(const char *path, uint64_t key, blz_prepared **out)
{
	if (!path || !out)
		return blz_fail(BLZ_EINVAL, "blz_prepared_load: bad argument");
	*out = NULL;
	const int fd = open(path, O_RDONLY);
	if (fd < 0)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	struct stat st;
	if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(prep_header)) {
		close(fd);
		return blz_fail(BLZ_EFORMAT, "%s is not a prepared-matrix cache", path);
	}
	char *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
	close(fd);
	if (m == MAP_FAILED)
		return blz_fail(BLZ_EIO, "cannot map %s: %s", path, strerror(errno));
	prep_header h;
	memcpy(&h, m, sizeof h);
	if (memcmp(h.magic, PREP_MAGIC, 8) != 0 || h.total != (uint64_t)st.st_size || h.key != key || h.nranks < 1) {
		munmap(m, (size_t)st.st_size);
		return blz_fail(BLZ_EFORMAT, "%s: not a cache of this matrix / prime / width / rank count (or an older format)", path);
	}
	if ((uint64_t)st.st_size < up64(sizeof h) ||
	    hash_bytes((const unsigned char *)m + up64(sizeof h), (size_t)st.st_size - up64(sizeof h)) != h.payload_hash) {
		munmap(m, (size_t)st.st_size);
		return blz_fail(BLZ_EFORMAT, "%s: damaged cache (checksum)", path);
	}
	blz_prepared *P = calloc(1, sizeof *P);
	if (!P) {
		munmap(m, (size_t)st.st_size);
		return blz_fail(BLZ_ENOMEM, "blz_prepared_load: out of memory");
	}
	P->map = m;
	P->map_len = (size_t)st.st_size;
	P->only_rank = -1;
	P->nrows = h.nrows;
	P->ncols = h.ncols;
	P->nnz = h.nnz;
	P->right = h.right;
	P->nranks = h.nranks;
	P->chunks = h.chunks;
	P->order_kind = h.order_kind;
	P->has_perm = h.has_perm;
	for (int q = 0; q < 2; q++) {
		P->hot[q] = h.hot[q];
		P->stride[q] = h.stride[q];
		P->share[q] = h.share[q];
		P->locality[q] = h.locality[q];
		P->perm[q] = h.has_perm ? (int32_t *)(m + h.off_perm[q]) : NULL;
		P->bounds[q] = (int64_t *)(m + h.off_bounds[q]);
		P->full[q].rows = q == 0 ? h.nrows : h.ncols;
		P->full[q].cols = q == 0 ? h.ncols : h.nrows;
		P->full[q].nnz = h.nnz;
		P->full[q].row_ptr = (uint32_t *)(m + h.off_rp[q]);
		P->full[q].col_idx = (int32_t *)(m + h.off_ci[q]);
		P->full[q].val = h.has_val[q] ? (uint32_t *)(m + h.off_va[q]) : NULL;
	}
	/* The header is input like any other (a truncated, stale or foreign FILENAME.<key>.blzcache next to the matrix): every
	 * array must lie inside the file, and what the set-up indexes with -- row pointers, bounds, permutations, column
	 * indices -- must be in range, or the caller prepares afresh (BLZ_EFORMAT).  ADVICE round 2. */
	const char *why = NULL;
	const uint64_t len = (uint64_t)st.st_size;
#define INSIDE(off, bytes) ((off) >= sizeof(prep_header) && (off) <= len && (uint64_t)(bytes) <= len - (off) && ((off) & 7u) == 0)
	if (h.nrows < 0 || h.ncols < 0 || h.nnz < 0 || h.nrows > INT32_MAX || h.ncols > INT32_MAX || h.nnz >= (int64_t)UINT32_MAX ||
	    h.chunks < 1 || h.nranks > 65536)
		why = "dimensions";
	for (int q = 0; q < 2 && !why; q++) {
		const int64_t rows = q == 0 ? h.nrows : h.ncols, cols = q == 0 ? h.ncols : h.nrows;
		if ((h.has_perm && !INSIDE(h.off_perm[q], 4 * (uint64_t)rows)) || !INSIDE(h.off_bounds[q], 8 * (uint64_t)(h.nranks + 1)) ||
		    !INSIDE(h.off_rp[q], 4 * (uint64_t)(rows + 1)) || !INSIDE(h.off_ci[q], 4 * (uint64_t)h.nnz) ||
		    (h.has_val[q] && !INSIDE(h.off_va[q], 4 * (uint64_t)h.nnz))) {
			why = "an array lies outside the file";
			break;
		}
		const uint32_t *rp = P->full[q].row_ptr;
		int64_t bad = rp[0] != 0 || rp[rows] != (uint32_t)h.nnz;
#pragma omp parallel for schedule(static) reduction(+ : bad) if (rows > 200000)
		for (int64_t r = 0; r < rows; r++)
			bad += rp[r] > rp[r + 1];
		const int32_t *ci = P->full[q].col_idx;
#pragma omp parallel for schedule(static) reduction(+ : bad) if (h.nnz > 200000)
		for (int64_t k = 0; k < h.nnz; k++)
			bad += ci[k] < 0 || ci[k] >= cols;
		if (bad) {
			why = "row pointers or column indices out of range";
			break;
		}
		if (h.has_perm) {
			const int32_t *pm = P->perm[q];
			int64_t oob = 0;
#pragma omp parallel for schedule(static) reduction(+ : oob) if (rows > 200000)
			for (int64_t r = 0; r < rows; r++)
				oob += pm[r] < 0 || pm[r] >= rows;
			if (oob) {
				why = "permutation out of range";
				break;
			}
		}
	}
	for (int sd = 0; sd < 2 && !why; sd++) {
		/* side 0 = rows of v: rows of M for a left kernel, columns for a right one */
		const int64_t dim = (sd == 0) == (h.right == 0) ? h.nrows : h.ncols;
		const int64_t *b = P->bounds[sd];
		int64_t mx = 0;
		if (b[0] != 0 || b[h.nranks] != dim)
			why = "partition bounds";
		for (int g = 0; g < h.nranks && !why; g++) {
			if (b[g] > b[g + 1])
				why = "partition bounds";
			else if (b[g + 1] - b[g] > mx)
				mx = b[g + 1] - b[g];
		}
		if (!why && (h.stride[sd] < mx || h.stride[sd] % h.chunks != 0 || h.stride[sd] > mx + h.chunks))
			why = "slab stride";
	}
#undef INSIDE
	if (why) {
		blz_prepared_free(P);
		return blz_fail(BLZ_EFORMAT, "%s: damaged cache (%s)", path, why);
	}
	*out = P;
	return BLZ_OK;
}

/* 64-bit content hash of a file (FNV-1a over 8-byte words, 1 MB pieces hashed in parallel and chained): the cache key
 * of the CLI.  0 on error. */
uint64_t blz_file_hash(const char *path)
{
	const int fd = open(path, O_RDONLY);
	if (fd < 0)
		return 0;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size <= 0) {
		close(fd);
		return 0;
	}
	const size_t len = (size_t)st.st_size;
	const unsigned char *m = mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED)
		return 0;
	const uint64_t h = hash_bytes(m, len);
	munmap((void *)m, len);
	return h ? h : 1;
}

/* --------------------------------------------------------------------------------- RNG */

/* sequential/lanczos_modp.c:67 */
void blz_rng_seed(uint64_t s[4])
{
	s[0] = 0x1415926535ull;
	s[1] = 0x8979323846ull;
	s[2] = 0x2643383279ull;
	s[3] = 0x5028841971ull;
}

/* sequential/lanczos_modp.c:69-87 */
uint64_t blz_rng_next(uint64_t s[4])
{
	const uint64_t sum = s[0] + s[3];
	const uint64_t result = ((sum << 23) | (sum >> 41)) + s[0];
	const uint64_t t = s[1] << 17;
	s[2] ^= s[0];
	s[3] ^= s[1];
	s[1] ^= s[2];
	s[0] ^= s[3];
	s[2] ^= t;
	s[3] = (s[3] << 45) | (s[3] >> 19);
	return result;
}

int blz_rng_fill(uint64_t *v, int64_t words, uint64_t prime)
{
	if (!v || words < 0 || prime < 2)
		return blz_fail(BLZ_EINVAL, "blz_rng_fill: bad argument");
	uint64_t s[4];
	blz_rng_seed(s);
	for (int64_t k = 0; k < words; k++)
		v[k] = blz_rng_next(s) % prime;
	return BLZ_OK;
}

/* ---------------------------------------------------------------------- result writer */

/* one line of the kernel file: the reference's "%d" of a u32 (negative from 2^31 on) below 2^32, plain decimal above */
static inline size_t format_word(char *dst, uint64_t w)
{
	char tmp[24];
	size_t len = 0, k = 0;
	if (w < 0x100000000ull && (w & 0x80000000ull)) {
		dst[len++] = '-';
		w = 0x100000000ull - w;
	}
	do {
		tmp[k++] = (char)('0' + w % 10);
		w /= 10;
	} while (w);
	while (k)
		dst[len++] = tmp[--k];
	dst[len++] = '\n';
	return len;
}

/* save_vector_block(), sequential/lanczos_modp.c:673-686: column-major, one word per line.  The N*n lines are
 * formatted by all cores, a few million at a time, and written in order. */
int blz_save_block(const char *path, int64_t nrows, int n, const uint64_t *v)
{
	if (!path || nrows < 0 || n < 1 || (!v && nrows > 0))
		return blz_fail(BLZ_EINVAL, "blz_save_block: bad argument");
	FILE *f = fopen(path, "w");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	fprintf(f, "%%%%MatrixMarket matrix array integer general\n");
	fprintf(f, "%%block of left-kernel vector computed by lanczos_modp\n");
	fprintf(f, "%ld %d\n", (long)nrows, n);
	enum { LINE_MAX_BYTES = 21, BATCH = 1 << 22 };
	const int64_t lines = nrows * n;
	int T = omp_get_max_threads();
	if (T > 64)
		T = 64;
	if (lines < 100000)
		T = 1;
	const int64_t batch = lines < BATCH ? (lines ? lines : 1) : BATCH;
	const int64_t per = (batch + T - 1) / T;
	char *buf = malloc((size_t)per * T * LINE_MAX_BYTES);
	if (!buf) {
		fclose(f);
		return blz_fail(BLZ_ENOMEM, "blz_save_block: out of memory");
	}
	int rc = BLZ_OK;
	for (int64_t l0 = 0; l0 < lines && rc == BLZ_OK; l0 += batch) {
		const int64_t l1 = l0 + batch < lines ? l0 + batch : lines;
		size_t used[64];
#pragma omp parallel for schedule(static, 1) num_threads(T)
		for (int t = 0; t < T; t++) {
			const int64_t a = l0 + (int64_t)t * per, b = a + per < l1 ? a + per : l1;
			char *dst = buf + (size_t)t * per * LINE_MAX_BYTES;
			size_t len = 0;
			if (a < b) {
				int64_t col = a / nrows, r = a % nrows;
				for (int64_t l = a; l < b; l++) {
					len += format_word(dst + len, v[r * n + col]);
					if (++r == nrows) {
						r = 0;
						col++;
					}
				}
			}
			used[t] = len;
		}
		for (int t = 0; t < T; t++)
			if (used[t] && fwrite(buf + (size_t)t * per * LINE_MAX_BYTES, 1, used[t], f) != used[t])
				rc = blz_fail(BLZ_EIO, "write error on %s", path);
	}
	free(buf);
	if (fclose(f) && rc == BLZ_OK)
		rc = blz_fail(BLZ_EIO, "write error on %s", path);
	return rc;
}

/* ------------------------------------------------------------------------ kernel checker */

int blz_check_kernel(const char *matrix_path, const char *kernel_path, uint64_t prime, int right, int64_t *bad_row,
		     int *bad_col)
{
	if (!matrix_path || !kernel_path || prime < 2)
		return blz_fail(BLZ_EINVAL, "blz_check_kernel: bad argument");
	blz_coo M;
	int rc = blz_mm_load(matrix_path, prime, &M);
	if (rc != BLZ_OK)
		return rc;
	const int64_t nrows = right ? M.ncols : M.nrows, ncols = right ? M.nrows : M.ncols;	/* :99-104 */

	int fd = open(kernel_path, O_RDONLY);
	struct stat st;
	if (fd < 0 || fstat(fd, &st) != 0 || st.st_size == 0) {
		if (fd >= 0)
			close(fd);
		blz_coo_free(&M);
		return blz_fail(BLZ_EIO, "cannot open %s", kernel_path);
	}
	char *base = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (base == MAP_FAILED) {
		blz_coo_free(&M);
		return blz_fail(BLZ_EIO, "mmap %s: %s", kernel_path, strerror(errno));
	}
	cursor c = { base, base + st.st_size };
	char line[1100];
	long long nk = 0, n = 0;
	uint64_t *x = NULL;
	unsigned __int128 *y = NULL;
	rc = BLZ_OK;
	if (next_line(&c, line, sizeof line) || (rc = check_banner(line, 1)) != BLZ_OK) {
		if (rc == BLZ_OK)
			rc = blz_fail(BLZ_EFORMAT, "Could not process Matrix Market banner.");
		goto done;
	}
	do {
		if (next_line(&c, line, sizeof line)) {
			rc = blz_fail(BLZ_EIO, "Cannot read kernel vector block size size");
			goto done;
		}
	} while (line[0] == '%');
	if (sscanf(line, "%lld %lld", &nk, &n) != 2 || n < 1) {
		rc = blz_fail(BLZ_EIO, "Cannot read kernel vector block size size");
		goto done;
	}
	if (nk != nrows) {
		rc = blz_fail(BLZ_EINVAL, "dimension mismatch");
		goto done;
	}
	x = malloc(sizeof *x * (size_t)(nrows * n + 1));
	y = calloc((size_t)(ncols * n + 1), sizeof *y);
	if (!x || !y) {
		rc = blz_fail(BLZ_ENOMEM, "cannot allocate vector blocks");
		goto done;
	}
	int any = 0;
	for (long long k = 0; k < n; k++)		/* column-major, :146-153 */
		for (long long i = 0; i < nrows; i++) {
			long long w;
			if (next_int(&c, &w)) {
				rc = blz_fail(BLZ_EIO, "parse error entry %lld, %lld", i, k);
				goto done;
			}
			/* a negative entry is the reference writer's "%d" of a u32 >= 2^31 */
			const uint64_t word = w < 0 ? (uint64_t)(uint32_t)(int32_t)w : (uint64_t)w;
			if (word >= prime) {
				rc = blz_fail(BLZ_EINVAL, "entry %lld, %lld out of bound", i, k);
				goto done;
			}
			x[i * n + k] = word;
			any |= (word != 0);
		}
	if (!any) {
		rc = 1;
		goto done;
	}
	/* unreduced 128-bit sums, one reduction per word: value < 2^32, x < 2^62, nnz per column < 2^34 */
	for (int64_t u = 0; u < M.nnz; u++) {
		const int64_t i = right ? M.j[u] : M.i[u], j = right ? M.i[u] : M.j[u];
		const uint64_t v = M.x[u];
		for (long long k = 0; k < n; k++)
			y[j * n + k] += (unsigned __int128)v * x[i * n + k];
	}
	for (int64_t j = 0; j < ncols && rc == BLZ_OK; j++)
		for (long long k = 0; k < n; k++)
			if (y[j * n + k] % prime != 0) {
				if (bad_row)
					*bad_row = j;
				if (bad_col)
					*bad_col = (int)k;
				rc = 2;
				break;
			}
done:
	free(x);
	free(y);
	munmap(base, (size_t)st.st_size);
	blz_coo_free(&M);
	return rc;
}

/* -------------------------------------------------------------------------- checkpoints */

typedef struct {
	char magic[8];		/* "BLZCKPT1" */
	uint64_t prime;
	int64_t nrows, iterations;
	int32_t n, right;
} ckpt_header;

int blz_checkpoint_save(const char *path, uint64_t prime, int n, int right, int64_t nrows, int64_t iterations,
			const uint64_t *v, const uint64_t *p)
{
	char tmp[4096];
	snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
	FILE *f = fopen(tmp, "wb");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", tmp, strerror(errno));
	ckpt_header h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, "BLZCKPT1", 8);
	h.prime = prime;
	h.nrows = nrows;
	h.iterations = iterations;
	h.n = n;
	h.right = right;
	const size_t words = (size_t)(nrows * n);
	int ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(v, sizeof *v, words, f) == words
	    && fwrite(p, sizeof *p, words, f) == words && fflush(f) == 0 && fsync(fileno(f)) == 0;
	ok = (fclose(f) == 0) && ok;
	if (!ok || rename(tmp, path) != 0) {
		unlink(tmp);
		return blz_fail(BLZ_EIO, "cannot write checkpoint %s: %s", path, strerror(errno));
	}
	return BLZ_OK;
}

int blz_checkpoint_load(const char *path, uint64_t prime, int n, int right, int64_t nrows, int64_t *iterations,
			uint64_t *v, uint64_t *p)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	ckpt_header h;
	const size_t words = (size_t)(nrows * n);
	int ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "BLZCKPT1", 8) == 0;
	if (ok && (h.prime != prime || h.n != n || h.right != right || h.nrows != nrows)) {
		fclose(f);
		return blz_fail(BLZ_EINVAL, "checkpoint %s was written for another prime/n/orientation/shape", path);
	}
	ok = ok && fread(v, sizeof *v, words, f) == words && fread(p, sizeof *p, words, f) == words;
	fclose(f);
	if (!ok)
		return blz_fail(BLZ_EIO, "checkpoint %s is truncated or not a checkpoint", path);
	*iterations = h.iterations;
	return BLZ_OK;
}

/* The reference's text snapshot, openMP/lanczos_modp.c:571-601: one "%d" per line,
 * block_size_pad lines per vector (padding rows are zero). */
static int64_t ref_block_size_pad(int n, int64_t nrows, int64_t ncols)
{
	const int64_t a = (nrows + n - 1) / n * n, b = (ncols + n - 1) / n * n;
	return (a > b ? a : b) * n;		/* sequential/lanczos_modp.c:595-597 */
}

static int write_ref_vector(const char *dir, const char *name, int64_t pad, int64_t words, const uint64_t *v)
{
	char path[4096];
	snprintf(path, sizeof path, "%s/%s", dir, name);
	FILE *f = fopen(path, "w");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	for (int64_t k = 0; k < pad; k++)
		fprintf(f, "%d\n", k < words ? (int)(uint32_t)v[k] : 0);
	return fclose(f) ? blz_fail(BLZ_EIO, "write error on %s", path) : BLZ_OK;
}

int blz_checkpoint_save_ref_text(const char *dir, int n, int64_t nrows, int64_t ncols, int64_t iterations,
				 double start, double now, const uint64_t *v, const uint64_t *tmp,
				 const uint64_t *Av, const uint64_t *p)
{
	const int64_t pad = ref_block_size_pad(n, nrows, ncols);
	char path[4096];
	snprintf(path, sizeof path, "%s/verbosity.txt", dir);
	FILE *f = fopen(path, "w");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	fprintf(f, "%d\n%f\n%f\n", (int)iterations, start, now);	/* openMP/lanczos_modp.c:603-620 */
	if (fclose(f))
		return blz_fail(BLZ_EIO, "write error on %s", path);
	int rc;
	if ((rc = write_ref_vector(dir, "v.txt", pad, nrows * n, v)) ||
	    (rc = write_ref_vector(dir, "tmp.txt", pad, ncols * n, tmp)) ||
	    (rc = write_ref_vector(dir, "Av.txt", pad, nrows * n, Av)) ||
	    (rc = write_ref_vector(dir, "p.txt", pad, nrows * n, p)))
		return rc;
	return BLZ_OK;
}

static int read_ref_vector(const char *dir, const char *name, int64_t words, uint64_t *v)
{
	char path[4096], line[128];
	snprintf(path, sizeof path, "%s/%s", dir, name);
	FILE *f = fopen(path, "r");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	int64_t k = 0;
	while (fgets(line, sizeof line, f)) {	/* openMP/lanczos_modp.c:622-647: atoi per line */
		if (k < words)
			v[k] = (uint64_t)(uint32_t)atoi(line);
		k++;
	}
	fclose(f);
	if (k < words)
		return blz_fail(BLZ_EIO, "%s holds %ld words, %ld needed", path, (long)k, (long)words);
	return BLZ_OK;
}

int blz_checkpoint_load_ref_text(const char *dir, int n, int64_t nrows, int64_t ncols, int64_t *iterations,
				 uint64_t *v, uint64_t *p)
{
	(void)ncols;
	char path[4096], line[128];
	snprintf(path, sizeof path, "%s/verbosity.txt", dir);
	FILE *f = fopen(path, "r");
	if (!f)
		return blz_fail(BLZ_EIO, "cannot open %s: %s", path, strerror(errno));
	*iterations = fgets(line, sizeof line, f) ? atoi(line) : 0;	/* openMP/lanczos_modp.c:661-664 */
	fclose(f);
	int rc;
	if ((rc = read_ref_vector(dir, "v.txt", nrows * n, v)) || (rc = read_ref_vector(dir, "p.txt", nrows * n, p)))
		return rc;
	return BLZ_OK;
}
