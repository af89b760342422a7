/*
 * lanczos_modp -- drop-in for the reference's command-line program, driving the MI355X kernels
 * through the C ABI of include/blz.h.  Plain C host code; no GPU code in this file.
 *
 * Same flags, same MatrixMarket input rules, same output file and the same key stdout lines as
 * sequential/lanczos_modp.c (main :690-705, options :141-194) plus the checkpoint flags of
 * openMP/lanczos_modp.c:189-243.  Differences, all stated in --help:
 *   - the prime may be anything in [2, 2^62) (the reference stops at 2^30-35, :189-193);
 *   - checkpoints are one binary file written atomically (lanczos_modp.ckpt in the CWD) by a helper thread from an
 *     asynchronous snapshot: the main loop does not stop for them (openMP/lanczos_modp.c:1013-1022 does);
 *     with BLZ_REF_CHECKPOINT=1 and p < 2^32 the reference's five text files are written too, and
 *     --load-checkpoint falls back to them when the binary file is absent.
 */
#define _GNU_SOURCE
#include <err.h>
#include <getopt.h>
#include <inttypes.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include "blz.h"

static long n = 1;
static uint64_t prime;
static char *matrix_filename, *kernel_filename;
static bool right_kernel, checkpoints, load_checkpoint, verify, use_cache;
static int stop_after = -1, checkpoint_timer = 60, device, gpus = 1;

static double wtime(void)
{
	struct timeval ts;
	gettimeofday(&ts, NULL);
	return (double)ts.tv_sec + ts.tv_usec / 1e6;
}

/* human_format(), sequential/lanczos_modp.c:99-120 */
static void human_format(char *target, long v)
{
	if (v < 1000)
		sprintf(target, "%ld", v);
	else if (v < 1000000)
		sprintf(target, "%.1fK", v / 1e3);
	else if (v < 1000000000)
		sprintf(target, "%.1fM", v / 1e6);
	else if (v < 1000000000000ll)
		sprintf(target, "%.1fG", v / 1e9);
	else
		sprintf(target, "%.1fT", v / 1e12);
}

static void usage(char **argv)
{
	printf("%s [OPTIONS]\n\n", argv[0]);
	printf("Options:\n");
	printf("--matrix FILENAME           MatrixMarket file containing the spasre matrix\n");
	printf("--prime P                   compute modulo P (2 <= P < 2**62; the reference's cap of 2**30-35 is lifted)\n");
	printf("--n N                       blocking factor [default 1, at most %d]\n", BLZ_MAX_N);
	printf("--output-file FILENAME      store the block of kernel vectors\n");
	printf("--right                     compute right kernel vectors\n");
	printf("--left                      compute left kernel vectors [default]\n");
	printf("--stop-after N              stop the algorithm after N iterations\n");
	printf("--checkpoint [cp]           make a checkpoint every cp seconds [default 60] (lanczos_modp.ckpt)\n");
	printf("--load-checkpoint           restart from the checkpoint in the current directory\n");
	printf("--verify                    check the reference's per-iteration invariants (correctness_tests) on the host;\n");
	printf("                            one host round trip per iteration, for debugging\n");
	printf("--cache                     keep the renumbered CSR(M), CSR(M^T) and the row partition next to the matrix\n");
	printf("                            (FILENAME.<key>.blzcache, keyed by content hash, prime, n, orientation, GPUs);\n");
	printf("                            a later run with the same arguments skips the renumbering and the CSR builds\n");
	printf("--device D                  first HIP device to run on [default 0]\n");
	printf("--gpus G                    row-partition the matrix over G GPUs of this node (devices D..D+G-1), RCCL\n");
	printf("                            all-gather of the block before each product [default 1]\n");
	printf("\n");
	printf("The --matrix and --prime arguments are required\n");
	printf("The --stop-after and --output-file arguments mutually exclusive\n");
	exit(0);
}

static void process_command_line_options(int argc, char **argv)
{
	struct option longopts[] = {
		{"matrix", required_argument, NULL, 'm'}, {"prime", required_argument, NULL, 'p'},
		{"n", required_argument, NULL, 'n'}, {"output-file", required_argument, NULL, 'o'},
		{"right", no_argument, NULL, 'r'}, {"left", no_argument, NULL, 'l'},
		{"stop-after", required_argument, NULL, 's'}, {"checkpoint", optional_argument, NULL, 'c'},
		{"load-checkpoint", no_argument, NULL, 'L'}, {"device", required_argument, NULL, 'd'},
		{"gpus", required_argument, NULL, 'g'}, {"verify", no_argument, NULL, 'V'},
		{"cache", no_argument, NULL, 'C'}, {"help", no_argument, NULL, 'h'}, {NULL, 0, NULL, 0}
	};
	int ch;
	while ((ch = getopt_long(argc, argv, "", longopts, NULL)) != -1) {
		switch (ch) {
		case 'm': matrix_filename = optarg; break;
		case 'n': n = atoi(optarg); break;
		case 'p': prime = strtoull(optarg, NULL, 10); break;
		case 'o': kernel_filename = optarg; break;
		case 'r': right_kernel = true; break;
		case 'l': right_kernel = false; break;
		case 's': stop_after = (int)atoll(optarg); break;
		case 'c':	/* optional, possibly space-separated value: openMP/lanczos_modp.c:225-235 */
			checkpoints = true;
			if (optarg == NULL && optind < argc && argv[optind][0] != '-')
				optarg = argv[optind++];
			if (optarg)
				checkpoint_timer = atoi(optarg);
			break;
		case 'L': load_checkpoint = true; break;
		case 'd': device = atoi(optarg); break;
		case 'g': gpus = atoi(optarg); break;
		case 'V': verify = true; break;
		case 'C': use_cache = true; break;
		case 'h': usage(argv); break;
		default: errx(1, "Unknown option\n");
		}
	}
	if (matrix_filename == NULL || prime == 0)	/* sequential/lanczos_modp.c:183-187 */
		usage(argv);
	if (kernel_filename != NULL && stop_after > 0)
		usage(argv);
	if (prime >= (1ull << 62))
		errx(1, "p is capped at 2**62 - 1.");
	if (n < 1 || n > BLZ_MAX_N)
		errx(1, "n must be between 1 and %d", BLZ_MAX_N);
	if (gpus < 1 || gpus > 64)
		errx(1, "--gpus must be between 1 and 64");
}

#define CHECK(call)                                                \
	do {                                                       \
		if ((call) != BLZ_OK)                              \
			errx(1, "%s", blz_last_error());           \
	} while (0)

static int n_iterations, expected_iterations;
static double start, last_print, extra_time;
static bool eta_flag;

/* verbosity(), sequential/lanczos_modp.c:494-529 (at most one line per second). */
static void verbosity(void)
{
	const double elapsed = wtime() - start;
	if (elapsed - last_print < 1)
		return;
	last_print = elapsed;
	const double per_iteration = (elapsed + extra_time) / (n_iterations > 0 ? n_iterations : 1);
	double estimated_length = expected_iterations * per_iteration;
	time_t end = (time_t)(start - extra_time + estimated_length);
	if (!eta_flag) {
		int d = (int)(estimated_length / 86400);
		estimated_length -= d * 86400.0;
		int h = (int)(estimated_length / 3600);
		estimated_length -= h * 3600.0;
		int m = (int)(estimated_length / 60);
		estimated_length -= m * 60.0;
		printf("    - Expected duration : ");
		if (d > 0) printf("%d j ", d);
		if (h > 0) printf("%d h ", h);
		if (m > 0) printf("%d min ", m);
		printf("%d s\n", (int)estimated_length);
		eta_flag = true;
	}
	char eta[30];
	ctime_r(&end, eta);
	eta[strlen(eta) - 1] = 0;
	printf("\r    - iteration %d / %d. %.3fs per iteration. ETA: %s", n_iterations, expected_iterations,
	       per_iteration, eta);
	fflush(stdout);
}

/*
 * correctness_tests(), sequential/lanczos_modp.c:532-557, on the n x n operands of the iteration just computed:
 * vtAv, vtAAv, winv symmetric; winv[i][j] != 0 only if d[i] or d[j]; winv * (vtAv restricted to the selected
 * columns) == diag(d).  The reference asserts these every iteration ("disable in production"); here they are the
 * --verify mode.
 */
static void correctness_tests(blz_ctx *ctx)
{
	const int nn = (int)(n * n);
	uint64_t *A = malloc(sizeof *A * (size_t)(3 * nn + n)), *B = A + nn, *W = B + nn, *d = W + nn;
	CHECK(blz_get_small(ctx, BLZ_VTAV, A));
	CHECK(blz_get_small(ctx, BLZ_VTAAV, B));
	CHECK(blz_get_small(ctx, BLZ_WINV, W));
	CHECK(blz_get_small(ctx, BLZ_D, d));
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) {
			if (A[i * n + j] != A[j * n + i] || B[i * n + j] != B[j * n + i] || W[i * n + j] != W[j * n + i])
				errx(1, "--verify: iteration %d: an n x n operand is not symmetric at (%d,%d)", n_iterations, i, j);
			if (W[i * n + j] != 0 && !d[i] && !d[j])
				errx(1, "--verify: iteration %d: winv[%d][%d] != 0 outside the selected block", n_iterations, i, j);
			unsigned __int128 acc = 0;
			for (int k = 0; k < n; k++)
				acc = (acc + (unsigned __int128)W[i * n + k] * (d[j] ? A[k * n + j] : 0)) % prime;
			if ((uint64_t)acc != (uint64_t)(i == j ? d[i] : 0))
				errx(1, "--verify: iteration %d: winv * vtAv * D != D at (%d,%d)", n_iterations, i, j);
		}
	free(A);
}

/*
 * One context per GPU, one host thread per context for the duration of each operation (the RCCL calls inside
 * blz_comm_init / blz_iterate / blz_final_check must be entered by all ranks concurrently).  With --gpus 1 the
 * operation runs on the calling thread.
 */
enum { OP_CREATE, OP_COMM, OP_MATRIX, OP_INIT, OP_SET_VP, OP_ITERATE, OP_GET, OP_FINAL, OP_SNAP, OP_DESTROY };

static struct {
	blz_ctx *ctx[64];
	const blz_prepared *P;
	char uid[128];
	blz_loop_group *loop;	/* BLZ_LOOPBACK=1: the G contexts share ONE device and meet in a loopback communicator (tests on a one-GPU box) */
	int op, todo, block;
	uint64_t *host, *host2;
	int64_t its;
	int done[64], stopped[64], nonzero[64], zero[64], rc[64];
	float ms[64];
	char err[64][512];
} team;

static void *team_worker(void *arg)
{
	const int g = (int)(intptr_t)arg;
	int rc = BLZ_OK;
	switch (team.op) {
	/* set-up in three phases with a join after each: a rank that fails in one of them (no memory on its device, say) is
	 * reported before its peers enter the next, instead of leaving them inside ncclCommInitRank for ever */
	case OP_CREATE:
		rc = blz_create(&team.ctx[g], team.loop ? device : device + g, prime, (int)n);
		break;
	case OP_COMM:
		rc = team.loop ? blz_comm_init_loopback(team.ctx[g], team.loop, g)
			       : blz_comm_init(team.ctx[g], team.uid, sizeof team.uid, g, gpus);
		break;
	case OP_MATRIX:
		rc = blz_set_matrix_prepared(team.ctx[g], team.P, g);
		break;
	case OP_INIT:
		rc = blz_init_v(team.ctx[g]);
		break;
	case OP_SET_VP:
		rc = blz_set_block(team.ctx[g], BLZ_V, team.host);
		if (rc == BLZ_OK)
			rc = blz_set_block(team.ctx[g], BLZ_P, team.host2);
		if (rc == BLZ_OK)
			rc = blz_set_iterations(team.ctx[g], team.its);
		break;
	case OP_ITERATE:
		rc = blz_iterate(team.ctx[g], team.todo, &team.done[g], &team.stopped[g], &team.ms[g]);
		break;
	case OP_GET:
		rc = blz_get_block(team.ctx[g], team.block, team.host);	/* writes only the rows this rank owns */
		break;
	case OP_FINAL:
		rc = blz_final_check(team.ctx[g], &team.nonzero[g], &team.zero[g]);
		break;
	case OP_SNAP:
		rc = blz_snapshot_begin(team.ctx[g]);
		break;
	case OP_DESTROY:
		blz_destroy(team.ctx[g]);
		break;
	}
	team.rc[g] = rc;
	if (rc != BLZ_OK)
		snprintf(team.err[g], sizeof team.err[g], "%s", blz_last_error());	/* blz_last_error is thread-local */
	return NULL;
}

static void team_run(int op)
{
	team.op = op;
	if (gpus == 1) {
		team_worker((void *)(intptr_t)0);
	} else {
		pthread_t th[64];
		for (int g = 0; g < gpus; g++) {
			team.rc[g] = BLZ_OK;
			if (pthread_create(&th[g], NULL, team_worker, (void *)(intptr_t)g))
				errx(1, "cannot start a host thread for GPU %d", g);
		}
		/* the communicator's rendez-vous is the one step where a missing rank leaves the others waiting inside RCCL:
		 * give it a deadline and leave from the main thread instead of joining for ever */
		struct timespec dl;
		clock_gettime(CLOCK_REALTIME, &dl);
		dl.tv_sec += 300;
		for (int g = 0; g < gpus; g++) {
			if (op == OP_COMM) {
				if (pthread_timedjoin_np(th[g], NULL, &dl) != 0) {
					for (int q = 0; q < gpus; q++)
						if (team.rc[q] != BLZ_OK)
							errx(1, "GPU %d: %s", device + q, team.err[q]);
					errx(1, "GPU %d did not join the RCCL communicator within 300 s", device + g);
				}
			} else {
				pthread_join(th[g], NULL);
			}
		}
	}
	for (int g = 0; g < gpus; g++)
		if (team.rc[g] != BLZ_OK)
			errx(1, "GPU %d: %s", device + g, team.err[g]);
}

static void team_get(int block, uint64_t *host)
{
	team.block = block;
	team.host = host;
	team_run(OP_GET);
}

/* the checkpoint writer: collects the snapshot every context has begun (blz_snapshot_wait is the one call that may come
 * from another thread than the context's owner), writes lanczos_modp.ckpt atomically */
static struct {
	pthread_t th;
	volatile bool busy;
	bool started;
	int64_t nrows;
	uint64_t *v, *p;
} writer;

static void *checkpoint_writer(void *arg)
{
	(void)arg;
	const size_t words = (size_t)(writer.nrows * n + 1);
	if (!writer.v) {
		writer.v = malloc(sizeof(uint64_t) * words);
		writer.p = malloc(sizeof(uint64_t) * words);
	}
	int64_t its = 0;
	int rc = (writer.v && writer.p) ? BLZ_OK : BLZ_ENOMEM;
	char why[512] = "";
	/* every context's snapshot is collected whatever happens -- a context left with one in flight refuses the next
	 * checkpoint, and a transient failure here must not end the solve: after the first failure the
	 * remaining ones are dropped (NULL buffers) */
	for (int g = 0; g < gpus; g++) {
		const int r = rc == BLZ_OK ? blz_snapshot_wait(team.ctx[g], writer.v, writer.p, &its)
					   : blz_snapshot_wait(team.ctx[g], NULL, NULL, NULL);
		if (r != BLZ_OK && rc == BLZ_OK) {
			rc = r;
			snprintf(why, sizeof why, "%s", blz_last_error());
		}
	}
	if (rc == BLZ_OK)
		rc = blz_checkpoint_save("lanczos_modp.ckpt", prime, (int)n, right_kernel, writer.nrows, its, writer.v, writer.p);
	if (rc == BLZ_OK)
		printf("\n		>> Snapshot written to lanczos_modp.ckpt (iteration %" PRId64 ")\n", its);
	else
		fprintf(stderr, "\ncheckpoint NOT written: %s\n", rc == BLZ_ENOMEM ? "out of memory" : (why[0] ? why : blz_last_error()));
	fflush(stdout);
	writer.busy = false;
	return NULL;
}

int main(int argc, char **argv)
{
	process_command_line_options(argc, argv);
	setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);	/* dmabuf IPC for RCCL; must precede the first HIP call */

	printf("Loading matrix from %s\n", matrix_filename);
	fflush(stdout);
	blz_coo M;
	const double t_load = wtime();
	CHECK(blz_mm_load(matrix_filename, prime, &M));
	fprintf(stderr, "  - [matrix coordinate integer general] %ld x %ld with %ld nz\n", (long)M.nrows, (long)M.ncols,
		(long)M.nnz);
	fprintf(stderr, "  - Read in %.2fs\n", wtime() - t_load);

	{
		const char *lb = getenv("BLZ_LOOPBACK");
		if (gpus > 1 && lb && lb[0] == '1') {
			CHECK(blz_loop_group_create(gpus, &team.loop));
			fprintf(stderr, "  - BLZ_LOOPBACK=1: %d ranks on GPU %d through the loopback communicator (a test mode, not a speed-up)\n", gpus,
				device);
		}
	}
	if (device < 0 || (team.loop ? device + 1 : device + gpus) > blz_device_count())
		errx(1, "GPU %d..%d requested but %d HIP device(s) are visible", device, device + gpus - 1, blz_device_count());
	team_run(OP_CREATE);
	if (gpus > 1) {
		if (!team.loop)
			CHECK(blz_comm_unique_id(team.uid, sizeof team.uid));
		team_run(OP_COMM);
	}
	blz_ctx *ctx = team.ctx[0];
	/* the rank-independent set-up (renumbering, CSR(M), CSR(M^T), row partition) is done ONCE here -- round 1 redid it in
	 * every context's thread -- or mapped from the cache of an earlier run (--cache) */
	blz_prepared *P = NULL;
	char cache_path[4096];
	uint64_t key = 0;
	const double t_prep = wtime();
	if (use_cache) {
		const uint64_t fh = blz_file_hash(matrix_filename);
		key = blz_prepare_key(ctx, fh, M.nrows, M.ncols, M.nnz, right_kernel, gpus);
		snprintf(cache_path, sizeof cache_path, "%s.%016" PRIx64 ".blzcache", matrix_filename, key);
		if (fh && blz_prepared_load(cache_path, key, &P) == BLZ_OK)
			fprintf(stderr, "  - Set-up mapped from %s in %.2fs\n", cache_path, wtime() - t_prep);
	}
	if (!P) {
		CHECK(blz_prepare_for(ctx, &M, right_kernel, gpus, &P));
		fprintf(stderr, "  - Renumbering, CSR(M), CSR(M^T), partition: %.2fs\n", wtime() - t_prep);
		if (use_cache) {
			if (blz_prepared_save(P, cache_path, key) == BLZ_OK)
				fprintf(stderr, "  - Set-up saved to %s\n", cache_path);
			else
				fprintf(stderr, "  - (cache not written: %s)\n", blz_last_error());
		}
	}
	team.P = P;
	team_run(OP_MATRIX);
	blz_prepared_free(P);
	const int64_t nrows = right_kernel ? M.ncols : M.nrows;
	const int64_t ncols = right_kernel ? M.nrows : M.ncols;
	blz_coo_free(&M);

	printf("Block Lanczos\n");
	const long npad = (long)((nrows + n - 1) / n * n), mpad = (long)((ncols + n - 1) / n * n);
	const long block_size_pad = (npad > mpad ? npad : mpad) * n;
	char human[32];
	human_format(human, 4L * blz_word_bytes(ctx) * block_size_pad);
	printf("  - Extra storage needed: %sB\n", human);
	expected_iterations = 1 + (int)(ncols / n);	/* sequential/lanczos_modp.c:610 */

	uint64_t *v = malloc(sizeof(uint64_t) * (size_t)(nrows * n + 1));
	uint64_t *p = malloc(sizeof(uint64_t) * (size_t)(nrows * n + 1));
	if (!v || !p)
		errx(1, "impossible d'allouer les blocs de vecteur");
	if (load_checkpoint) {
		int64_t its = 0;
		if (access("lanczos_modp.ckpt", R_OK) == 0)
			CHECK(blz_checkpoint_load("lanczos_modp.ckpt", prime, (int)n, right_kernel, nrows, &its, v, p));
		else
			CHECK(blz_checkpoint_load_ref_text(".", (int)n, nrows, ncols, &its, v, p));
		team_run(OP_INIT);
		team.host = v;
		team.host2 = p;
		team.its = its;
		team_run(OP_SET_VP);
		n_iterations = (int)its;
		expected_iterations -= n_iterations;	/* openMP/lanczos_modp.c:971-972 */
	} else {
		team_run(OP_INIT);
	}
	human_format(human, expected_iterations);
	printf("  - Expecting %s iterations\n", human);

	printf("  - Main loop\n");
	start = wtime();
	double checkpoint_start = wtime();
	int batch = 1, stopped = 0;
	while (!stopped) {
		int todo = verify ? 1 : batch;
		if (stop_after > 0) {
			if (n_iterations >= stop_after)
				break;
			if (todo > stop_after - n_iterations)
				todo = stop_after - n_iterations;
		}
		team.todo = todo;
		team_run(OP_ITERATE);
		const int done = team.done[0];
		const float ms = team.ms[0];
		stopped = team.stopped[0];
		n_iterations += done;
		if (verify)
			correctness_tests(ctx);
		verbosity();
		/* keep the host out of the loop: grow the batch until one batch takes ~0.25 s */
		if (ms < 250.0f && batch < 4096)
			batch *= 2;
		if (checkpoints && !stopped && (wtime() - checkpoint_start) >= checkpoint_timer) {
			const char *ref = getenv("BLZ_REF_CHECKPOINT");
			if (ref && ref[0] == '1' && prime < (1ull << 32)) {
				/* the reference's five text files need tmp and Av too: the synchronous way, as round 1 did */
				printf("\n");
				team_get(BLZ_V, v);
				team_get(BLZ_P, p);
				printf("		>> Making a snapshot in lanczos_modp.ckpt (iteration %d)\n", n_iterations);
				CHECK(blz_checkpoint_save("lanczos_modp.ckpt", prime, (int)n, right_kernel, nrows, n_iterations, v, p));
				uint64_t *t = calloc((size_t)(ncols * n + 1), sizeof *t), *a = calloc((size_t)(nrows * n + 1), sizeof *a);
				team_get(BLZ_TMP, t);
				team_get(BLZ_AV, a);
				CHECK(blz_checkpoint_save_ref_text(".", (int)n, nrows, ncols, n_iterations, start, wtime(), v, t, a, p));
				free(t);
				free(a);
				checkpoint_start = wtime();
			} else if (!writer.busy) {
				/* asynchronous: the copies of v and p are enqueued on a side stream (the GPU pauses for the PCIe
				 * transfer only), a helper thread waits for them, writes the file and renames it; the loop goes on.
				 * A checkpoint that comes due while the previous one is still being written is skipped. */
				if (writer.started) {
					pthread_join(writer.th, NULL);
					writer.started = false;
				}
				team_run(OP_SNAP);
				writer.nrows = nrows;
				writer.busy = true;
				writer.started = true;
				if (pthread_create(&writer.th, NULL, checkpoint_writer, NULL))
					errx(1, "cannot start the checkpoint writer");
				checkpoint_start = wtime();
			}
		}
	}
	if (writer.started) {
		pthread_join(writer.th, NULL);
		writer.started = false;
	}
	printf("\n");

	if (stop_after < 0) {		/* final_check(), sequential/lanczos_modp.c:560-582 */
		team_run(OP_FINAL);
		const int nonzero = team.nonzero[0], zero = team.zero[0];
		printf("Final check:\n");
		printf(nonzero ? "  - OK:    v != 0\n" : "  - KO:    v == 0\n");
		printf(zero ? "  - OK: vt*M == 0\n" : "  - KO: vt*M != 0\n");
	}
	printf("  - Terminated in %.1fs after %d iterations\n", wtime() - start, n_iterations);

	if (kernel_filename) {
		team_get(BLZ_V, v);
		printf("Saving result in %s\n", kernel_filename);
		CHECK(blz_save_block(kernel_filename, nrows, (int)n, v));
	} else {
		printf("Not saving result (no --output given)\n");
	}
	free(v);
	free(p);
	team_run(OP_DESTROY);
	if (team.loop)
		blz_loop_group_destroy(team.loop);
	exit(EXIT_SUCCESS);
}
