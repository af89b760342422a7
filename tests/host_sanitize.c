/* Host half of libblz_hip.so under AddressSanitizer + UBSan (CPU build; the GPU pool has no sanitizer runs).
 * Compiled and run by tests/test_host_sanitize.py:  host_sanitize <golden dir> <scratch dir>
 * Walks every host-side entry point of include/blz.h on small and on multi-threaded-path-sized inputs, including the
 * error paths; any invalid access aborts the process with a sanitizer report. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "blz.h"

#define REQUIRE(cond)                                                                         \
	do {                                                                                  \
		if (!(cond)) {                                                                \
			fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond, blz_last_error()); \
			exit(2);                                                              \
		}                                                                             \
	} while (0)

static void matrix_round(const blz_coo *M, uint64_t prime, const char *scratch)
{
	for (int t = 0; t < 2; t++)
		for (int pattern = 0; pattern < 2; pattern++) {
			blz_csr A;
			REQUIRE(blz_csr_from_coo(M, t, pattern, &A) == BLZ_OK);
			REQUIRE(A.nnz == M->nnz && A.row_ptr[A.rows] == (uint32_t)M->nnz);
			int64_t bounds[9];
			for (int parts = 1; parts <= 8; parts += 3) {
				REQUIRE(blz_partition_rows(&A, parts, bounds) == BLZ_OK);
				REQUIRE(bounds[0] == 0 && bounds[parts] == A.rows);
			}
			blz_csr_free(&A);
		}
	int32_t *rp = malloc(sizeof *rp * (size_t)(M->nrows + 1)), *cp = malloc(sizeof *cp * (size_t)(M->ncols + 1));
	REQUIRE(rp && cp);
	REQUIRE(blz_reorder(M, rp, cp) == BLZ_OK);
	free(rp);
	free(cp);
	for (int right = 0; right < 2; right++)
		for (int nranks = 1; nranks <= 5; nranks += 2)
			for (int chunks = 1; chunks <= 4; chunks += 3)
				for (int rank = 0; rank < nranks; rank++) {
					blz_csr slabs[2];
					int64_t b0[6], b1[6], stride[2];
					REQUIRE(blz_shard_matrix(M, right, rank, nranks, chunks, slabs, b0, b1, stride) == BLZ_OK);
					REQUIRE(stride[0] % chunks == 0 && stride[1] % chunks == 0);
					for (int s = 0; s < 2; s++) {
						/* rows of M gather from tmp's side for a left kernel, from v's side for a right one */
						const int64_t width = stride[s == 0 ? (right ? 0 : 1) : (right ? 1 : 0)] * nranks;
						for (int64_t k = 0; k < slabs[s].nnz; k++)
							REQUIRE(slabs[s].col_idx[k] >= 0 && slabs[s].col_idx[k] < width);
						blz_csr_free(&slabs[s]);
					}
				}
	/* round 2: scored renumbering with and without a hot panel, row sort, the prepared matrix (every rank's slabs in
	 * both forms), its cache file (save, map, wrong key, truncated), the file hash */
	{
		int32_t *r2 = malloc(sizeof *r2 * (size_t)(M->nrows + 1)), *c2 = malloc(sizeof *c2 * (size_t)(M->ncols + 1));
		REQUIRE(r2 && c2);
		int64_t hot[2] = { 7, 300 };
		double share[2], loc[2];
		int kind = -1;
		REQUIRE(blz_reorder_hot(M, r2, c2, hot, 0.0, share) == BLZ_OK);
		REQUIRE(hot[0] <= 7 && hot[1] <= 300);
		hot[0] = 5;
		hot[1] = 64;
		REQUIRE(blz_reorder_auto(M, r2, c2, hot, 0.01, share, 2, loc, &kind) == BLZ_OK && kind >= 0 && kind <= 3);
		for (int64_t r = 0; r < M->nrows; r++)
			REQUIRE(r2[r] >= 0 && r2[r] < M->nrows);
		free(r2);
		free(c2);
		blz_csr A;
		REQUIRE(blz_csr_from_coo(M, 1, 0, &A) == BLZ_OK);
		blz_csr_sort_rows(&A);
		for (int64_t r = 0; r < A.rows; r++)
			for (uint32_t k = A.row_ptr[r] + 1; k < A.row_ptr[r + 1]; k++)
				REQUIRE(A.col_idx[k - 1] <= A.col_idx[k]);
		blz_csr_free(&A);
	}
	for (int right = 0; right < 2; right++)
		for (int nranks = 1; nranks <= 4; nranks += 3) {
			const int chunks = nranks == 1 ? 1 : 2;
			blz_prepared *P = NULL, *Q = NULL;
			REQUIRE(blz_prepare(M, right, nranks, chunks, 1, 2, nranks == 1 ? 50 : 0, 0.05, &P) == BLZ_OK);
			int pr, pn, pc;
			REQUIRE(blz_prepared_describe(P, &pr, &pn, &pc) == BLZ_OK && pr == right && pn == nranks && pc == chunks);
			char cpath[1024];
			snprintf(cpath, sizeof cpath, "%s/prep.blzcache", scratch);
			REQUIRE(blz_prepared_save(P, cpath, 77 + (uint64_t)nranks) == BLZ_OK);
			REQUIRE(blz_prepared_load(cpath, 78 + (uint64_t)nranks, &Q) != BLZ_OK && Q == NULL);
			REQUIRE(blz_prepared_load(cpath, 77 + (uint64_t)nranks, &Q) == BLZ_OK);
			for (int rank = 0; rank < nranks; rank++)
				for (int t = 0; t < 2; t++) {
					blz_csr a, b, sh;
					REQUIRE(blz_prepared_slab(P, rank, t, &a) == BLZ_OK && blz_prepared_slab(Q, rank, t, &b) == BLZ_OK);
					REQUIRE(a.rows == b.rows && a.nnz == b.nnz);
					REQUIRE(!memcmp(a.row_ptr, b.row_ptr, sizeof *a.row_ptr * (size_t)(a.rows + 1)));
					REQUIRE(!memcmp(a.col_idx, b.col_idx, sizeof *a.col_idx * (size_t)a.nnz));
					REQUIRE(blz_prepared_slab_short(Q, rank, t, &sh) == BLZ_OK);
					for (int64_t k = 0; k < sh.nnz; k++)
						REQUIRE(sh.col_idx[k] >= 0 && sh.col_idx[k] < (sh.cols > 0 ? sh.cols : 1));
					REQUIRE(sh.row_ptr[sh.rows] == (uint32_t)sh.nnz);
					blz_csr_free(&a);
					blz_csr_free(&b);
					blz_csr_free(&sh);
				}
			blz_prepared_free(Q);
			blz_prepared_free(P);
			/* damaged caches (round 3): a flipped word behind the header (checksum), a header whose offsets point past the
			 * file, a truncated file: refused, nothing read out of bounds */
			{
				FILE *fd = fopen(cpath, "r+");
				REQUIRE(fd != NULL);
				REQUIRE(fseek(fd, 0, SEEK_END) == 0);
				const long len = ftell(fd);
				uint32_t junk = 0x7FFFFFF0u;
				REQUIRE(len > 400 && fseek(fd, (320 + (len - 320) / 2) & ~3l, SEEK_SET) == 0 && fwrite(&junk, 4, 1, fd) == 1);
				fclose(fd);
				Q = NULL;
				REQUIRE(blz_prepared_load(cpath, 77 + (uint64_t)nranks, &Q) != BLZ_OK && Q == NULL);
				fd = fopen(cpath, "r+");
				uint64_t huge = (uint64_t)1 << 40;
				for (long off = 8 * 20; off < 8 * 34; off += 8) {	/* the offset words of the header */
					REQUIRE(fseek(fd, off, SEEK_SET) == 0 && fwrite(&huge, 8, 1, fd) == 1);
				}
				fclose(fd);
				REQUIRE(blz_prepared_load(cpath, 77 + (uint64_t)nranks, &Q) != BLZ_OK && Q == NULL);
			}
			FILE *fc = fopen(cpath, "r+");
			REQUIRE(fc && ftruncate(fileno(fc), 100) == 0);
			fclose(fc);
			Q = NULL;
			REQUIRE(blz_prepared_load(cpath, 77 + (uint64_t)nranks, &Q) != BLZ_OK);
			REQUIRE(blz_file_hash(cpath) != 0 && blz_file_hash("/nonexistent-file") == 0);
		}
	char path[1024];
	snprintf(path, sizeof path, "%s/copy.mtx", scratch);
	REQUIRE(blz_mm_save_coo(path, M) == BLZ_OK);
	blz_coo L;
	REQUIRE(blz_mm_load(path, prime, &L) == BLZ_OK);
	REQUIRE(L.nnz == M->nnz && L.nrows == M->nrows && L.ncols == M->ncols);
	for (int64_t k = 0; k < L.nnz; k++)
		REQUIRE(L.i[k] == M->i[k] && L.j[k] == M->j[k] && L.x[k] == M->x[k] % prime);
	blz_coo_free(&L);
}

int main(int argc, char **argv)
{
	if (argc < 3)
		return 64;
	const char *golden = argv[1], *scratch = argv[2];
	char path[1024], kpath[1024];
	const uint64_t primes[3] = { 65537, 1073741789, ((uint64_t)1 << 61) - 1 };
	const char *names[4] = { "trefethen20", "quirks40x30", "wide120x260", "rand3000x2000" };
	for (int m = 0; m < 4; m++)
		for (int q = 0; q < 3; q++) {
			blz_coo M;
			snprintf(path, sizeof path, "%s/%s.mtx", golden, names[m]);
			REQUIRE(blz_mm_load(path, primes[q], &M) == BLZ_OK);
			matrix_round(&M, primes[q], scratch);
			blz_coo_free(&M);
		}
	/* sizes that take the OpenMP paths (parser pieces, atomics CSR build, parallel writer) */
	blz_coo S;
	REQUIRE(blz_synth_coo(70000, 50000, 400000, 0x5A17, 0, primes[1], &S) == BLZ_OK);
	matrix_round(&S, primes[1], scratch);
	blz_coo P;
	REQUIRE(blz_synth_coo(50000, 60000, 300000, 9, 1, primes[2], &P) == BLZ_OK);
	matrix_round(&P, primes[2], scratch);
	blz_coo_free(&P);
	REQUIRE(blz_synth_coo(10, 3, 40, 1, 0, 7, &P) != BLZ_OK);		/* more entries per row than columns */
	REQUIRE(blz_synth_structured(40000, 42000, 400000, 3, 0, primes[2], 40, 30, 512, &P) == BLZ_OK);
	matrix_round(&P, primes[2], scratch);
	blz_coo_free(&P);
	REQUIRE(blz_synth_structured(100, 100, 400, 3, 0, 7, 60, 60, 64, &P) != BLZ_OK);	/* percentages above 100 */

	/* round 3: a rank's share generated and prepared alone (blz_synth_coo_part, blz_prepare_rank): every rank of 1, 3 and 8,
	 * both orientations, gathering and short-side slabs; bad ranges and foreign entries refused */
	{
		const int64_t R = 30000, C = 4000, NZ = 150000;
		for (int nranks = 1; nranks <= 8; nranks += (nranks == 1 ? 2 : 5))
			for (int right = 0; right < 2; right++) {
				int64_t rb[9], cb[9];
				for (int g = 0; g <= nranks; g++) {
					rb[g] = R * g / nranks;
					cb[g] = C * g / nranks;
				}
				for (int rank = 0; rank < nranks; rank++) {
					blz_coo rp_, cp_;
					REQUIRE(blz_synth_coo_part(R, C, NZ, 0xBEEF, 0, primes[2], rb[rank], rb[rank + 1], 0, C, &rp_) == BLZ_OK);
					REQUIRE(blz_synth_coo_part(R, C, NZ, 0xBEEF, 0, primes[2], 0, R, cb[rank], cb[rank + 1], &cp_) == BLZ_OK);
					blz_prepared *Q = NULL;
					REQUIRE(blz_prepare_rank(&rp_, &cp_, R, C, NZ, right, rank, nranks, nranks > 1 ? 2 : 1, rb, cb, &Q) == BLZ_OK);
					for (int t = 0; t < 2; t++) {
						blz_csr a, sh;
						REQUIRE(blz_prepared_slab(Q, rank, t, &a) == BLZ_OK);
						REQUIRE(a.row_ptr[a.rows] == (uint32_t)a.nnz);
						for (int64_t k = 0; k < a.nnz; k++)
							REQUIRE(a.col_idx[k] >= 0 && a.col_idx[k] < a.cols);
						blz_csr_free(&a);
						REQUIRE(blz_prepared_slab_short(Q, rank, t, &sh) == BLZ_OK);
						blz_csr_free(&sh);
						if (nranks > 1)
							REQUIRE(blz_prepared_slab(Q, (rank + 1) % nranks, t, &a) != BLZ_OK);
					}
					REQUIRE(blz_prepared_save(Q, "/dev/null", 1) != BLZ_OK);
					blz_prepared_free(Q);
					if (nranks > 1) {	/* the rows of another rank are refused, not misfiled */
						Q = NULL;
						REQUIRE(blz_prepare_rank(&rp_, &cp_, R, C, NZ, right, (rank + 1) % nranks, nranks, 1, rb, cb, &Q) != BLZ_OK && Q == NULL);
					}
					blz_coo_free(&rp_);
					blz_coo_free(&cp_);
				}
			}
		blz_coo bad;
		REQUIRE(blz_synth_coo_part(R, C, NZ, 1, 0, 7, 10, R + 1, 0, C, &bad) != BLZ_OK);
		REQUIRE(blz_synth_coo_part(R, C, NZ, 1, 0, 7, 5, 5, 0, C, &bad) == BLZ_OK && bad.nnz == 0);
		blz_coo_free(&bad);
	}
	/* round 3: the iterated-sweeps ordering, chosen on a band matrix whose rows and columns are scrambled */
	{
		const int64_t R = 20000, per = 12, band = 300;
		blz_coo B = { R, R, R * per, malloc(sizeof(int32_t) * (size_t)(R * per)), malloc(sizeof(int32_t) * (size_t)(R * per)),
			      malloc(sizeof(uint32_t) * (size_t)(R * per)) };
		int32_t *pr = malloc(sizeof(int32_t) * (size_t)R), *pc = malloc(sizeof(int32_t) * (size_t)R);
		REQUIRE(B.i && B.j && B.x && pr && pc);
		uint64_t st = 12345;
		for (int64_t r = 0; r < R; r++)
			pr[r] = pc[r] = (int32_t)r;
		for (int64_t r = R - 1; r > 0; r--) {	/* two shuffles */
			st = st * 6364136223846793005ull + 1442695040888963407ull;
			int64_t q = (int64_t)((st >> 33) % (uint64_t)(r + 1));
			int32_t t_ = pr[r]; pr[r] = pr[q]; pr[q] = t_;
			st = st * 6364136223846793005ull + 1442695040888963407ull;
			q = (int64_t)((st >> 33) % (uint64_t)(r + 1));
			t_ = pc[r]; pc[r] = pc[q]; pc[q] = t_;
		}
		for (int64_t r = 0; r < R; r++)
			for (int64_t k = 0; k < per; k++) {
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				const int64_t c = (r + (int64_t)((st >> 33) % (uint64_t)band) - band / 2 + R) % R;
				B.i[r * per + k] = pr[r];
				B.j[r * per + k] = pc[c];
				B.x[r * per + k] = 1;
			}
		int32_t *r2 = malloc(sizeof(int32_t) * (size_t)R), *c2 = malloc(sizeof(int32_t) * (size_t)R);
		int64_t hot[2] = { 0, 0 };
		double share[2], loc[2];
		int kind = -1;
		REQUIRE(r2 && c2 && blz_reorder_auto(&B, r2, c2, hot, 0.25, share, 2, loc, &kind) == BLZ_OK);
		REQUIRE(kind == 3 && loc[0] < 0.7 && loc[1] < 0.7);
		free(r2);
		free(c2);
		free(pr);
		free(pc);
		blz_coo_free(&B);
	}

	/* RNG, kernel writer, checker (zero block, wrong block, bad shapes), both word widths */
	const int n = 4;
	const int64_t rows = S.nrows;
	uint64_t *v = malloc(sizeof *v * (size_t)(rows * n)), *pb = malloc(sizeof *pb * (size_t)(rows * n));
	REQUIRE(v && pb);
	REQUIRE(blz_rng_fill(v, rows * n, primes[1]) == BLZ_OK);
	snprintf(path, sizeof path, "%s/copy.mtx", scratch);
	REQUIRE(blz_mm_save_coo(path, &S) == BLZ_OK);
	snprintf(kpath, sizeof kpath, "%s/k.mtx", scratch);
	REQUIRE(blz_save_block(kpath, rows, n, v) == BLZ_OK);
	int64_t bad_row = -1;
	int bad_col = -1;
	REQUIRE(blz_check_kernel(path, kpath, primes[1], 0, &bad_row, &bad_col) == 2 && bad_row >= 0);	/* random block: y != 0 */
	REQUIRE(blz_check_kernel(path, kpath, primes[1], 1, &bad_row, &bad_col) < 0);			/* dimension mismatch */
	REQUIRE(blz_check_kernel(path, kpath, 65537, 0, &bad_row, &bad_col) < 0);				/* entries >= prime */
	memset(v, 0, sizeof *v * (size_t)(rows * n));
	REQUIRE(blz_save_block(kpath, rows, n, v) == BLZ_OK);
	REQUIRE(blz_check_kernel(path, kpath, primes[1], 0, &bad_row, &bad_col) == 1);			/* all zero */
	REQUIRE(blz_check_kernel(path, path, primes[1], 0, &bad_row, &bad_col) < 0);			/* not an array file */
	REQUIRE(blz_rng_fill(v, rows * n, primes[2]) == BLZ_OK);						/* words >= 2^32 */
	REQUIRE(blz_save_block(kpath, rows, n, v) == BLZ_OK);
	REQUIRE(blz_check_kernel(path, kpath, primes[2], 0, &bad_row, &bad_col) == 2);

	/* checkpoints: binary round trip, mismatch, truncated file; the reference's text files */
	REQUIRE(blz_rng_fill(pb, rows * n, primes[1]) == BLZ_OK);
	REQUIRE(blz_rng_fill(v, rows * n, primes[1]) == BLZ_OK);
	snprintf(kpath, sizeof kpath, "%s/ck.bin", scratch);
	REQUIRE(blz_checkpoint_save(kpath, primes[1], n, 0, rows, 1234, v, pb) == BLZ_OK);
	uint64_t *v2 = malloc(sizeof *v2 * (size_t)(rows * n)), *p2 = malloc(sizeof *p2 * (size_t)(rows * n));
	int64_t its = 0;
	REQUIRE(v2 && p2);
	REQUIRE(blz_checkpoint_load(kpath, primes[1], n, 0, rows, &its, v2, p2) == BLZ_OK && its == 1234);
	REQUIRE(!memcmp(v, v2, sizeof *v * (size_t)(rows * n)) && !memcmp(pb, p2, sizeof *pb * (size_t)(rows * n)));
	REQUIRE(blz_checkpoint_load(kpath, primes[1], n + 1, 0, rows, &its, v2, p2) != BLZ_OK);
	REQUIRE(blz_checkpoint_load(kpath, primes[1], n, 0, rows - 1, &its, v2, p2) != BLZ_OK);
	FILE *f = fopen(kpath, "r+");
	REQUIRE(f && ftruncate(fileno(f), 4096) == 0);
	fclose(f);
	REQUIRE(blz_checkpoint_load(kpath, primes[1], n, 0, rows, &its, v2, p2) != BLZ_OK);
	const int64_t small_rows = 300, small_cols = 200;
	REQUIRE(blz_checkpoint_save_ref_text(scratch, n, small_rows, small_cols, 17, 0.0, 1.0, v, pb, v, pb) == BLZ_OK);
	REQUIRE(blz_checkpoint_load_ref_text(scratch, n, small_rows, small_cols, &its, v2, p2) == BLZ_OK && its == 17);
	REQUIRE(!memcmp(v, v2, sizeof *v * (size_t)(small_rows * n)));
	REQUIRE(blz_checkpoint_load_ref_text("/nonexistent-dir", n, small_rows, small_cols, &its, v2, p2) != BLZ_OK);

	/* malformed matrix files */
	static const char *bad_files[] = {
		"", "%%MatrixMarket matrix coordinate integer general\n", "%%MatrixMarket matrix coordinate integer general\n3 3\n",
		"%%MatrixMarket matrix coordinate integer general\n3 3 2\n1 1 1\n", "%%MatrixMarket matrix coordinate integer general\n3 3 1\n4 1 1\n",
		"%%MatrixMarket matrix coordinate integer general\n-3 3 1\n1 1 1\n", "%%MatrixMarket matrix coordinate integer general\n3 3 1\n1 1 x\n",
		"%%MatrixMarket matrix array real general\n3 3\n1\n", "garbage\n",
	};
	for (size_t k = 0; k < sizeof bad_files / sizeof *bad_files; k++) {
		snprintf(path, sizeof path, "%s/bad.mtx", scratch);
		f = fopen(path, "w");
		REQUIRE(f);
		fputs(bad_files[k], f);
		fclose(f);
		blz_coo B;
		REQUIRE(blz_mm_load(path, 65537, &B) != BLZ_OK);
		REQUIRE(blz_check_kernel(path, path, 65537, 0, &bad_row, &bad_col) < 0);
	}
	free(v);
	free(pb);
	free(v2);
	free(p2);
	blz_coo_free(&S);
	puts("host half clean under ASan + UBSan");
	return 0;
}
