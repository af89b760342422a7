/*
 * checker_modp_oracle -- command line over orc_check_kernel, with the flags and
 * exit codes of the reference's checker_modp.c:43-76 (exit 0 + "OK", exit 1 on
 * any failure).  TEST INFRASTRUCTURE ONLY.
 */
#define _POSIX_C_SOURCE 200809L
#include "blz_oracle.h"
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
	const struct option opts[] = {
		{"matrix", required_argument, NULL, 'm'}, {"kernel", required_argument, NULL, 'k'},
		{"prime", required_argument, NULL, 'p'}, {"right", no_argument, NULL, 'r'},
		{"left", no_argument, NULL, 'l'}, {NULL, 0, NULL, 0}
	};
	const char *matrix = NULL, *kernel = NULL;
	uint64_t prime = 0;
	int right = 0, ch;
	while ((ch = getopt_long(argc, argv, "", opts, NULL)) != -1)
		switch (ch) {
		case 'm': matrix = optarg; break;
		case 'k': kernel = optarg; break;
		case 'p': prime = strtoull(optarg, NULL, 10); break;
		case 'r': right = 1; break;
		case 'l': right = 0; break;
		default: fprintf(stderr, "Unknown option\n"); return 1;
		}
	if (!matrix || !kernel || !prime) {
		printf("%s --matrix FILE --kernel FILE --prime P [--right|--left]\n", argv[0]);
		return 0;
	}
	char err[256] = "";
	const int rc = orc_check_kernel(matrix, kernel, prime, right, err, sizeof err);
	if (rc == 0) {
		printf("OK\n");
		return 0;
	}
	if (rc == 1)
		fprintf(stderr, "KO: kernel vectors are all zero\n");
	else if (rc == 2)
		fprintf(stderr, "KO: y != 0\n");
	else
		fprintf(stderr, "%s\n", err);
	return 1;
}
