/*
 * blz_dense_mfma.hip -- the row-local block update (orthogonalize(), sequential/lanczos_modp.c:456-492) on the matrix
 * cores, for p = 2^61 - 1 and block widths 8 and 16.
 *
 * The update is, per block row r,
 *     v'[r,:] = (d ? Av : v)[r,:] + v[r,:] * c + p[r,:] * vtAvd          p'[r,:] = (d ? 0 : p)[r,:] + v[r,:] * winv
 * i.e. three (rows x n) by (n x n) products of 61-bit residues.  On the vector ALU that is 3 n^2 64x64->128 MACs per row
 * (7 instructions each): at n = 16 the kernel is VALU-bound at a third of the HBM rate (round 1: 494 us on the GL7d19
 * shape, 0.29 of the roofline).  Here the products run as EXACT INTEGER contractions on v_mfma_i32_16x16x64_i8:
 *
 *   - a block row in HBM is already the A operand: its bytes are the base-256 digits of its words, in the order
 *     K' = (word k, byte a).  A lane takes 16 of them with one 16-byte load; XOR 0x80 turns a byte u into u - 128,
 *     which fits the instruction's signed 8-bit inputs.
 *   - the n x n coefficients are rewritten once per launch (k_ortho_mfma_prep) as SIGNED base-256 digits d_b in
 *     [-128, 127] (9 digits with the carry), so that no bias is needed on that side.  For every s = a + b (0..15) the
 *     B operand B_s[(k,a)][j] = d_{s-a}(coef[k][j]) sits in LDS in fragment order.
 *   - acc_s[r][j] = sum over K' of (u - 128) * B_s  is one chain of 2..4 MFMAs; the accumulator starts at
 *     128 * (column sum of B_s) + 2^24, which removes the bias of A and keeps the digit sum S_s non-negative (< 2^25).
 *   - the result sum_s S_s * 2^(8 s) is folded mod 2^61 - 1 on the fly: 2^(8 s) = 2^(8 s mod 61), every term is ONE
 *     v_mad_u64_u32 into one of two 64-bit sums (shifts 0..27 and 32..59), then one 128-bit fold.  The 2^24 biases add up
 *     to a constant that is subtracted mod p at the end.
 * Everything is integer and exact, so the words written are the ones the VALU kernels (and the reference) write.
 *
 * n = 16: two chains per 16-row tile, [v | p] x [c ; vtAvd] (K' = 256, 4 MFMAs per s) and v x winv (K' = 128, 2 per s).
 * n = 8:  one chain, [v | p] x [[c | winv] ; [vtAvd | 0]] (K' = 128, 2 per s): columns 0..7 of the tile are v', 8..15 p'.
 */
#include "blz_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));

#define MBLOCK 512

template <int NT>
struct OG {
	static constexpr int KS1 = NT == 16 ? 4 : 2, KS2 = NT == 16 ? 2 : 0, KS = KS1 + KS2, NCH = NT == 16 ? 2 : 1;
	static constexpr size_t B_BYTES = (size_t)KS * 16 * 64 * 16, INIT_BYTES = (size_t)NCH * 16 * 16 * 4;
	static constexpr size_t IMG_BYTES = B_BYTES + INIT_BYTES + 16;
};

/* weight of digit sum s in the folded result: 2^(8 s mod 61), applied as a 32-bit multiplier into L (shift < 32) or H */
__host__ __device__ constexpr int fold_shift(int s) { return (8 * s) % 61; }

/* signed base-256 digits of x < 2^62: x = sum d_b 256^b, d_b in [-128, 127], b = 0..8 */
__device__ static void signed_digits(u64 x, signed char *d)
{
	int carry = 0;
	for (int b = 0; b < 8; b++) {
		int t = (int)((x >> (8 * b)) & 0xFF) + carry;
		carry = t >= 128;
		d[b] = (signed char)(carry ? t - 256 : t);
	}
	d[8] = (signed char)carry;
}

/* coefficient of K' word kk and tile column col for chain ch: which matrix, which entry (or none) */
template <int NT>
__device__ static int coef_index(int ch, int kk, int col)
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

template <int NT>
__global__ void __launch_bounds__(256)
k_ortho_mfma_prep(const u64 *__restrict__ small, unsigned char *__restrict__ img, u64 p, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NN = NT * NT;
	using G = OG<NT>;
	__shared__ signed char dig[3 * NN][9];
	const int t = threadIdx.x;
	for (int e = t; e < 3 * NN; e += 256) {
		const int mat = e / NN, at = e % NN;
		const u64 x = small[(mat == 0 ? 4 : (mat == 1 ? 5 : 2)) * NN + at];
		signed_digits(x, dig[e]);
	}
	__syncthreads();
	signed char *B = (signed char *)img;
	int *init = (int *)(img + G::B_BYTES);
	for (int idx = blockIdx.x * 256 + t; idx < (int)G::B_BYTES; idx += gridDim.x * 256) {
		const int j = idx & 15, ln = (idx >> 4) & 63, s = (idx >> 10) & 15, ks = idx >> 14;
		const int ch = ks < G::KS1 ? 0 : 1, kc = ch == 0 ? ks : ks - G::KS1;
		const int E = 64 * kc + 16 * (ln >> 4) + j, kk = E >> 3, a = E & 7, b = s - a;
		const int ci = coef_index<NT>(ch, kk, ln & 15);
		B[idx] = (ci >= 0 && b >= 0 && b <= 8) ? dig[ci][b] : (signed char)0;
	}
	if (blockIdx.x != 0)
		return;
	/* accumulator start: 128 * (sum over K' of B_s[.][col]) + 2^24 */
	for (int e = t; e < G::NCH * 16 * 16; e += 256) {
		const int col = e & 15, s = (e >> 4) & 15, ch = e >> 8;
		const int words = ch == 0 ? G::KS1 * 8 : G::KS2 * 8;
		int sum = 0;
		for (int kk = 0; kk < words; kk++) {
			const int ci = coef_index<NT>(ch, kk, col);
			if (ci < 0)
				continue;
			for (int a = 0; a < 8; a++) {
				const int b = s - a;
				if (b >= 0 && b <= 8)
					sum += dig[ci][b];
			}
		}
		init[e] = 128 * sum + (1 << 24);
	}
	if (t == 0) {
		/* the 2^24 added to each of the 16 digit sums, with their fold weights: subtract it at the end */
		unsigned __int128 bias = 0;
		for (int s = 0; s < 16; s++)
			bias += (unsigned __int128)(1u << 24) << fold_shift(s);
		const u64 r = (u64)(bias % p);
		*(u64 *)(img + G::B_BYTES + G::INIT_BYTES) = r ? p - r : 0;
	}
}

/* acc += a * b (32 x 32 -> 64, 64-bit add).  Plain C on purpose: the operands come straight out of an MFMA, and the compiler
 * pads the MFMA -> VALU read hazard only for instructions it can see (inline asm read stale accumulators here).  `b` is a
 * power of two the compiler must not recognise as one (it would shift and add: three instructions instead of one
 * v_mad_u64_u32), hence the opaque `one` below. */
MODP_DEV void mad_u64(u64 &acc, u32 a, u32 b)
{
	acc += (u64)a * (u64)b;
}

MODP_DEV u64 fold61(u64 L, u64 H)
{
	/* value = L + H * 2^32 (L < 2^61, H < 2^56): fold mod 2^61 - 1 */
	const u64 P = (1ull << 61) - 1;
	const u64 lo = L + (H << 32), carry = lo < L;
	const u64 hi = (H >> 32) + carry;
	u64 x = (lo & P) + ((lo >> 61) | (hi << 3));	/* hi < 2^25: (hi << 3) < 2^28 */
	x = (x & P) + (x >> 61);
	return x >= P ? x - P : x;
}

template <int NT>
__global__ void __launch_bounds__(MBLOCK)
k_ortho_mfma(u64 *__restrict__ V, const u64 *__restrict__ AV, u64 *__restrict__ Pb, long long rows,
	     const u64 *__restrict__ small, const unsigned char *__restrict__ img, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	using G = OG<NT>;
	constexpr int NN = NT * NT;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	{
		const uint4 *src = (const uint4 *)img;
		uint4 *dst = (uint4 *)lds;
		for (int i = threadIdx.x; i < (int)((G::B_BYTES + G::INIT_BYTES + 16) / 16); i += MBLOCK)
			dst[i] = src[i];
	}
	__syncthreads();
	const v4i *Bl = (const v4i *)lds;
	const int *init = (const int *)(lds + G::B_BYTES);
	const u64 fconst = *(const u64 *)(lds + G::B_BYTES + G::INIT_BYTES);
	const u64 PR = (1ull << 61) - 1;
	u32 one;
	asm volatile("s_mov_b32 %0, 1" : "=s"(one));	/* 1, opaque to the optimiser (see mad_u64) */
	const int lane = threadIdx.x & 63, m = lane & 15, h = lane >> 4, col = m;
	const long long wave = ((long long)blockIdx.x * MBLOCK + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * MBLOCK) >> 6;
	const long long ntiles = (rows + 15) >> 4;
	/* which output this lane's tile column is, and its selector d */
	const int jout = NT == 16 ? col : (col & 7);
	const bool is_p = NT == 8 && col >= 8;
	const bool dj = small[3 * NN + jout] != 0;
	for (long long tile = wave; tile < ntiles; tile += nwaves) {
		const long long r0 = tile << 4;
		long long ra = r0 + m;
		ra = ra < rows ? ra : rows - 1;
		/* A fragments: 16 bytes of the row per K-step, biased to signed */
		v4i A[G::KS1];
		if (NT == 16) {
			A[0] = *(const v4i *)(V + (size_t)ra * 16 + 2 * h);
			A[1] = *(const v4i *)(V + (size_t)ra * 16 + 8 + 2 * h);
			A[2] = *(const v4i *)(Pb + (size_t)ra * 16 + 2 * h);
			A[3] = *(const v4i *)(Pb + (size_t)ra * 16 + 8 + 2 * h);
		} else {
			A[0] = *(const v4i *)(V + (size_t)ra * 8 + 2 * h);
			A[1] = *(const v4i *)(Pb + (size_t)ra * 8 + 2 * h);
		}
		/* the terms that are not products, in the accumulator layout: row 4 h + reg, column col */
		u64 base1[4], base2[4];
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			long long rr = r0 + 4 * h + reg;
			rr = rr < rows ? rr : rows - 1;
			const size_t at = (size_t)rr * NT + jout;
			if (NT == 16) {
				base1[reg] = dj ? AV[at] : V[at];
				base2[reg] = dj ? 0 : Pb[at];
			} else {
				base1[reg] = is_p ? (dj ? 0 : Pb[at]) : (dj ? AV[at] : V[at]);
				base2[reg] = 0;
			}
		}
#pragma unroll
		for (int q = 0; q < G::KS1; q++)
			A[q] ^= (v4i){ (int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080 };
		u64 L1[4] = { 0, 0, 0, 0 }, H1[4] = { 0, 0, 0, 0 }, L2[4] = { 0, 0, 0, 0 }, H2[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int s = 0; s < 16; s++) {
			const int i1 = init[(0 * 16 + s) * 16 + col];
			v4i acc1 = { i1, i1, i1, i1 };
#pragma unroll
			for (int ks = 0; ks < G::KS1; ks++)
				acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks], Bl[(ks * 16 + s) * 64 + lane], acc1, 0, 0, 0);
			v4i acc2 = { 0, 0, 0, 0 };
			if (NT == 16) {
				const int i2 = init[(1 * 16 + s) * 16 + col];
				acc2 = (v4i){ i2, i2, i2, i2 };
#pragma unroll
				for (int ks = 0; ks < G::KS2; ks++)
					acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks], Bl[((G::KS1 + ks) * 16 + s) * 64 + lane], acc2, 0, 0, 0);
			}
			const int sh = fold_shift(s);
			const u32 mul = one << (sh < 32 ? sh : sh - 32);
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				/* one v_mad_u64_u32 per digit sum (the compiler would shift and add: three instructions) */
				if (sh < 32) {
					mad_u64(L1[reg], (u32)acc1[reg], mul);
					if (NT == 16)
						mad_u64(L2[reg], (u32)acc2[reg], mul);
				} else {
					mad_u64(H1[reg], (u32)acc1[reg], mul);
					if (NT == 16)
						mad_u64(H2[reg], (u32)acc2[reg], mul);
				}
			}
			/* keep the digit sums sequential: without this the compiler hoists all 16 x KS fragment reads (and spills) */
			asm volatile("" ::: "memory");
		}
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const long long rr = r0 + 4 * h + reg;
			if (rr >= rows)
				continue;
			const size_t at = (size_t)rr * NT + jout;
			u64 x = fold61(L1[reg], H1[reg]);
			x = addmod(x, fconst, PR);
			x = addmod(x, base1[reg], PR);
			if (NT == 16) {
				V[at] = x;
				u64 y = fold61(L2[reg], H2[reg]);
				y = addmod(y, fconst, PR);
				y = addmod(y, base2[reg], PR);
				Pb[at] = y;
			} else if (is_p) {
				Pb[at] = x;
			} else {
				V[at] = x;
			}
		}
	}
}

size_t ortho_mfma_image_bytes(void) { return OG<16>::IMG_BYTES; }

bool ortho_mfma_supported(const KernelCfg &c)
{
	return c.mfma && c.mfma_img && c.word == 8 && c.mers == 61 && (c.n == 8 || c.n == 16);
}

hipError_t launch_orthogonalize_mfma(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows, const u64 *small,
				     const DevCtl *ctl, hipStream_t s)
{
	if (rows <= 0)
		return hipSuccess;
	const long long ntiles = (rows + 15) / 16;
	long long blocks = (ntiles + MBLOCK / 64 - 1) / (MBLOCK / 64);
	/* the B image sits in LDS: 98 KB at n = 16 (one workgroup of 8 wavefronts per CU), 33 KB at n = 8 (two: 120 VGPRs) */
	const long long cap = (long long)c.num_cu * (c.n == 16 ? 1 : 2);
	blocks = blocks > cap ? cap : blocks;
	unsigned char *img = (unsigned char *)c.mfma_img;
	if (c.n == 16) {
		static bool attr = false;
		if (!attr) {
			(void)hipFuncSetAttribute((const void *)k_ortho_mfma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OG<16>::IMG_BYTES);
			attr = true;
		}
		hipLaunchKernelGGL((k_ortho_mfma_prep<16>), dim3(32), dim3(256), 0, s, small, img, c.m.p, ctl);
		hipLaunchKernelGGL((k_ortho_mfma<16>), dim3((unsigned)blocks), dim3(MBLOCK), OG<16>::IMG_BYTES, s, (u64 *)V, (const u64 *)AV,
				   (u64 *)P, (long long)rows, small, img, ctl);
	} else {
		hipLaunchKernelGGL((k_ortho_mfma_prep<8>), dim3(16), dim3(256), 0, s, small, img, c.m.p, ctl);
		hipLaunchKernelGGL((k_ortho_mfma<8>), dim3((unsigned)blocks), dim3(MBLOCK), OG<8>::IMG_BYTES, s, (u64 *)V, (const u64 *)AV,
				   (u64 *)P, (long long)rows, small, img, ctl);
	}
	return hipGetLastError();
}
