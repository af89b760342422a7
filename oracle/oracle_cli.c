/*
 * lanczos_modp_oracle -- the oracle behind the reference's command line
 * (sequential/lanczos_modp.c:141-194, :690-705).  TEST INFRASTRUCTURE ONLY:
 * it exists so that end-to-end outputs of the HIP solver can be compared
 * byte for byte with a CPU run for primes the reference refuses (p > 2^30-35).
 */
#define _POSIX_C_SOURCE 200809L
#include "blz_oracle.h"
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
	const struct option opts[] = {
		{"matrix", required_argument, NULL, 'm'}, {"prime", required_argument, NULL, 'p'},
		{"n", required_argument, NULL, 'n'}, {"output-file", required_argument, NULL, 'o'},
		{"right", no_argument, NULL, 'r'}, {"left", no_argument, NULL, 'l'},
		{"stop-after", required_argument, NULL, 's'}, {NULL, 0, NULL, 0}
	};
	const char *matrix = NULL, *out = NULL;
	uint64_t prime = 0;
	int n = 1, right = 0, stop_after = -1, ch;
	while ((ch = getopt_long(argc, argv, "", opts, NULL)) != -1)
		switch (ch) {
		case 'm': matrix = optarg; break;
		case 'p': prime = strtoull(optarg, NULL, 10); break;
		case 'n': n = atoi(optarg); break;
		case 'o': out = optarg; break;
		case 'r': right = 1; break;
		case 'l': right = 0; break;
		case 's': stop_after = atoi(optarg); break;
		default: fprintf(stderr, "Unknown option\n"); return 1;
		}
	if (!matrix || !prime || (out && stop_after > 0)) {
		printf("%s --matrix FILE --prime P [--n N] [--output-file FILE] [--right|--left] [--stop-after N]\n", argv[0]);
		return 0;
	}
	char err[256];
	orc_coo M;
	if (orc_mm_load(matrix, prime, &M, err, sizeof err)) {
		fprintf(stderr, "%s\n", err);
		return 1;
	}
	const int64_t nrows = right ? M.ncols : M.nrows, ncols = right ? M.nrows : M.ncols;
	uint64_t *v = malloc(sizeof(uint64_t) * (size_t)(nrows * n + 1));
	uint64_t *t = malloc(sizeof(uint64_t) * (size_t)(ncols * n + 1));
	const int its = orc_block_lanczos(&M, n, prime, right, stop_after, v, t, NULL, NULL, NULL, 0, NULL, NULL);
	if (stop_after < 0) {
		const int fc = orc_final_check(nrows, ncols, n, v, t);
		printf("Final check:\n");
		printf(fc & 1 ? "  - OK:    v != 0\n" : "  - KO:    v == 0\n");
		printf(fc & 2 ? "  - OK: vt*M == 0\n" : "  - KO: vt*M != 0\n");
	}
	printf("  - Terminated after %d iterations\n", its);
	if (out) {
		printf("Saving result in %s\n", out);
		if (orc_save_block(out, nrows, n, v))
			return 1;
	}
	return 0;
}
