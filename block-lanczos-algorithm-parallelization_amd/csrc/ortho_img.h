/*
 * ortho_img.h -- layout of the coefficient image the matrix-core block update reads (k_ortho_mfma, blz_dense_mfma.hip)
 * and the device code that builds it.  Two kernels build it: k_ortho_mfma_prep (stand-alone blz_orthogonalize) and the
 * semi-inverse kernel of the iteration, which has just computed the coefficients and still holds them in LDS
 * (blz_kernels.hip; round 3: one launch instead of two).
 *
 * orthogonalize(), sequential/lanczos_modp.c:456-492: c and vtAvd (:460-475) and winv are the n x n coefficients.
 */
#ifndef BLZ_ORTHO_IMG_H
#define BLZ_ORTHO_IMG_H

#include "modp.h"

#if defined(__HIPCC__)

template <int NT, bool ST = true>
struct OG {
	static constexpr int KS1 = NT == 16 ? 4 : 2, KS2 = NT == 16 ? 2 : 0, KS = KS1 + KS2, NCH = NT == 16 ? 2 : 1;
	static constexpr int ND = 8;		/* digit positions of a multiplier */
	static constexpr size_t B_BYTES = (size_t)KS * ND * 64 * 16, INIT_BYTES = (size_t)NCH * ND * 16 * 4;
	static constexpr size_t IMG_BYTES = B_BYTES + INIT_BYTES;
	static constexpr int NE = NCH * ND * 16;	/* accumulator-start entries */
	/* n = 16 stages the rows through LDS; per wavefront 16 rows of v and 16 of p, row stride = row bytes + 16 (bank spread) */
	static constexpr bool STAGED = ST;
	static constexpr bool PREFETCH = ST;		/* load the next tile during the arithmetic: the direct form lost by it (150 vs 124 us) */
	static constexpr int RSTR = NT * 8 + 16;
	static constexpr size_t STAGE_BYTES = STAGED ? (size_t)3 * 16 * RSTR : 0;	/* v, p, Av */
	static constexpr size_t lds_bytes(int threads) { return IMG_BYTES + (size_t)(threads / 64) * STAGE_BYTES; }
	/* threads per workgroup (BLZ_MFMA_BLOCK takes fewer).  n = 16: ONE workgroup of 16 wavefronts per CU next to the 49 KB image
	 * and their 108 KB of staging areas -- 1.77 ms with 2 x 4 wavefronts, 1.56 with 12, 1.49 with 16 on the config-5 quarter
	 * shape, which is what a bare streaming kernel with this traffic reaches.  n = 8: 256 (five workgroups per CU). */
	static constexpr int THREADS = !ST ? 256 : (NT == 16 ? 1024 : 768);
};

/* digit t (signed, base 256) of x < 2^62: with C = 0x8080...80, x = sum_t (byte_t(x + C) - 128) 256^t and every
 * byte_t - 128 lies in [-128, 127]; x + C does not wrap. */
MODP_DEV int signed_digit(u64 x, int t)
{
	return (int)(((x + 0x8080808080808080ull) >> (8 * t)) & 0xFF) - 128;
}

/* x * 2^sh mod 2^61 - 1 for x < 2^61, sh < 61: a rotation of the 61-bit word */
MODP_DEV u64 rot61(u64 x, int sh)
{
	const u64 P = (1ull << 61) - 1;
	const u64 r = ((x << sh) & P) | (x >> (61 - sh));
	return sh == 0 ? x : (r == P ? 0 : r);
}

/* coefficient of K' word kk and tile column col for chain ch: which matrix, which entry (or none) */
template <int NT>
MODP_DEV int coef_index(int ch, int kk, int col)
{
	/* returns an index into the digit table [3][NT*NT]: 0 = c, 1 = vtAvd, 2 = winv; -1 = zero */
	if (NT == 16) {
		if (ch == 0)
			return kk < 16 ? 0 * 256 + kk * 16 + col : 1 * 256 + (kk - 16) * 16 + col;
		return 2 * 256 + kk * 16 + col;
	}
	if (col < 8)
		return kk < 8 ? 0 * 64 + kk * 8 + col : 1 * 64 + (kk - 8) * 8 + col;
	return kk < 8 ? 2 * 64 + kk * 8 + (col - 8) : -1;
}

/* coefficient of word kk and tile column col in chain ch (0 where the chain has none); the multiplier of K' = (kk, byte a)
 * is rot61(coefficient, 8 a) = coefficient * 2^(8a) mod p.  `small` is the context's [vtAv | vtAAv | winv | d | c | vtAvd]
 * (global memory, or a copy of the same six n x n panels anywhere else). */
template <int NT>
MODP_DEV u64 coefficient(const u64 *small, int ch, int kk, int col)
{
	constexpr int NN = NT * NT;
	const int ci = coef_index<NT>(ch, kk, col);
	if (ci < 0)
		return 0;
	const int mat = ci / NN, at = ci % NN;
	return small[(mat == 0 ? 4 : (mat == 1 ? 5 : 2)) * NN + at];
}

/*
 * The whole image by ONE workgroup of `nthreads` threads: a thread makes 16 consecutive bytes of B at a time (one lane's
 * fragment of one K step and digit position: two coefficients, eight rotations each, one 16-byte store) and adds their
 * sum to the accumulator-start entry of its (chain, digit, column) in `init_sh` (LDS, OG<NT>::NE ints, zeroed here);
 * `coef` = the six-panel layout of `small` (LDS copy of the coefficients).  Ends with every thread past a barrier and the
 * image complete in `img`.
 */
template <int NT>
MODP_DEV void ortho_image_build(const u64 *coef, unsigned char *__restrict__ img, int *init_sh, int tid, int nthreads)
{
	using G = OG<NT>;
	for (int e = tid; e < G::NE; e += nthreads)
		init_sh[e] = 0;
	__syncthreads();
	/* fragment order: [K step ks][digit position][lane][16 bytes]; lane = 16 (K' / 16 within the step) + column */
	for (int item = tid; item < (int)(G::B_BYTES / 16); item += nthreads) {
		const int ln = item & 63, dg = (item >> 6) & 7, ks = item >> 9;
		const int ch = ks < G::KS1 ? 0 : 1, kc = ch == 0 ? ks : ks - G::KS1;
		const int E0 = 64 * kc + 16 * (ln >> 4), col = ln & 15;
		const u64 c0 = coefficient<NT>(coef, ch, E0 >> 3, col), c1 = coefficient<NT>(coef, ch, (E0 >> 3) + 1, col);
		u32 out[4] = { 0, 0, 0, 0 };
		int sum = 0;
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int d = signed_digit(rot61(j < 8 ? c0 : c1, 8 * (j & 7)), dg);
			sum += d;
			out[j >> 2] |= (u32)(d & 0xFF) << (8 * (j & 3));
		}
		*(uint4 *)(img + (size_t)item * 16) = make_uint4(out[0], out[1], out[2], out[3]);
		atomicAdd(&init_sh[(ch * G::ND + dg) * 16 + col], sum);
	}
	__syncthreads();
	int *init = (int *)(img + G::B_BYTES);
	for (int e = tid; e < G::NE; e += nthreads)
		init[e] = 128 * init_sh[e] + (1 << 24);
}

#endif /* __HIPCC__ */
#endif
