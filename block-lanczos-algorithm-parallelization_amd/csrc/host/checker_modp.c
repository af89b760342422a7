/*
 * checker_modp -- drop-in for the reference's verifier (checker_modp.c) with residues widened to 64 bits, so
 * that kernels computed modulo primes above 2^31-1 can be checked too.  Same flags (:43-76), same verdict
 * lines and exit codes: "OK" + exit 0, or a KO message + exit 1.  Plain C, no GPU.
 */
#define _GNU_SOURCE
#include <err.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

#include "blz.h"

int main(int argc, char **argv)
{
	struct option longopts[] = {
		{"matrix", required_argument, NULL, 'm'}, {"kernel", required_argument, NULL, 'k'},
		{"prime", required_argument, NULL, 'p'}, {"right", no_argument, NULL, 'r'},
		{"left", no_argument, NULL, 'l'}, {NULL, 0, NULL, 0}
	};
	char *matrix = NULL, *kernel = NULL;
	unsigned long long prime = 0;
	int right = 0, ch;
	while ((ch = getopt_long(argc, argv, "", longopts, NULL)) != -1) {
		switch (ch) {
		case 'm': matrix = optarg; break;
		case 'k': kernel = optarg; break;
		case 'p': prime = strtoull(optarg, NULL, 10); break;
		case 'r': right = 1; break;
		case 'l': right = 0; break;
		default: errx(1, "Unknown option\n");
		}
	}
	if (matrix == NULL || kernel == NULL || prime == 0) {
		printf("%s [OPTIONS]\n\n", argv[0]);
		printf("Options:\n");
		printf("--matrix FILENAME           MatrixMarket file containing the sparse matrix\n");
		printf("--kernel FILENAME           MatrixMarket file containing the kernel vectors\n");
		printf("--prime P                   compute modulo P (up to 2**62)\n");
		printf("--right                     check right kernel vectors\n");
		printf("--left                      check left kernel vectors [default]\n");
		exit(0);
	}
	printf("Reading Matrix from %s and kernel from %s\n", matrix, kernel);
	long long row = 0;
	int col = 0;
	const int rc = blz_check_kernel(matrix, kernel, prime, right, (int64_t *)&row, &col);
	if (rc == 0) {
		printf("OK\n");
		exit(EXIT_SUCCESS);
	}
	if (rc == 1)
		errx(1, "KO: kernel vectors are all zero");
	if (rc == 2)
		errx(1, "KO: y[%lld, %d] != 0\n", row, col);
	errx(1, "%s", blz_last_error());
}
