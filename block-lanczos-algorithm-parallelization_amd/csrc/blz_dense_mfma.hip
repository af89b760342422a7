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
 *   - byte a of word k weighs 2^(8a): the n x n coefficients are rewritten once per launch (k_ortho_mfma_prep) as
 *     coef[k][j] * 2^(8a) mod p -- a rotation of the 61-bit word, p being 2^61 - 1 -- one 61-bit multiplier per K' = (k, a),
 *     and each multiplier as 8 SIGNED base-256 digits d_t in [-128, 127], so that no bias is needed on that side.  For
 *     every digit position t (0..7) the B operand B_t[(k,a)][j] = d_t(coef[k][j] 2^(8a) mod p) sits in LDS in fragment
 *     order.  (Contracting digit against digit by s = a + b instead needs 16 digit sums: twice the MFMAs, LDS reads and
 *     folding work.  That was the first version of this kernel.)
 *   - acc_t[r][j] = sum over K' of (u - 128) * B_t  is one chain of 2..4 MFMAs; the accumulator starts at
 *     128 * (column sum of B_t) + 2^24, which removes the bias of A and keeps the digit sum S_t positive (< 2^25).
 *   - the result sum_t S_t * 2^(8 t) needs no wrap (t < 8): every term is ONE v_mad_u64_u32 into one of two 64-bit sums
 *     (t < 4 and t >= 4), then one 128-bit fold mod 2^61 - 1.  The 2^24 biases add up to a constant that is subtracted
 *     mod p at the end.
 * Everything is integer and exact, so the words written are the ones the VALU kernels (and the reference) write.
 *
 * n = 16: two chains per 16-row tile, [v | p] x [c ; vtAvd] (K' = 256, 4 MFMAs per t) and v x winv (K' = 128, 2 per t).
 * n = 8:  one chain, [v | p] x [[c | winv] ; [vtAvd | 0]] (K' = 128, 2 per t): columns 0..7 of the tile are v', 8..15 p'.
 */
#include "blz_kernels.h"
#include "ortho_img.h"

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));

/* weight of digit sum s in the folded result: 2^(8 s mod 61), applied as a 32-bit multiplier into L (shift < 32) or H */
__host__ __device__ constexpr int fold_shift(int s) { return (8 * s) % 61; }

template <int NT>
__global__ void __launch_bounds__(256)
k_ortho_mfma_prep(const u64 *__restrict__ small, unsigned char *__restrict__ img, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	using G = OG<NT>;
	const int t = threadIdx.x;
	signed char *B = (signed char *)img;
	int *init = (int *)(img + G::B_BYTES);
	/* fragment order: [K step ks][digit position][lane][16 bytes]; lane = 16 (K' / 16 within the step) + column */
	for (int idx = blockIdx.x * 256 + t; idx < (int)G::B_BYTES; idx += gridDim.x * 256) {
		const int j = idx & 15, ln = (idx >> 4) & 63, dg = (idx >> 10) & 7, ks = idx >> 13;
		const int ch = ks < G::KS1 ? 0 : 1, kc = ch == 0 ? ks : ks - G::KS1;
		const int E = 64 * kc + 16 * (ln >> 4) + j;
		B[idx] = (signed char)signed_digit(rot61(coefficient<NT>(small, ch, E >> 3, ln & 15), 8 * (E & 7)), dg);
	}
	/* accumulator start: 128 * (sum over K' of B_t[.][col]) + 2^24: one wavefront per entry, its K' over the lanes (the
	 * grid has NE / 4 workgroups; a loop over the entries with workgroup-wide sums took 7 us per launch) */
	constexpr int NE = G::NCH * G::ND * 16;
	for (int e = (blockIdx.x * 256 + t) >> 6; e < NE; e += (int)gridDim.x * 4) {
		const int col = e & 15, dg = (e >> 4) & 7, ch = e >> 7;
		const int nk = (ch == 0 ? G::KS1 : G::KS2) * 64;
		int sum = 0;
		for (int kp = t & 63; kp < nk; kp += 64)
			sum += signed_digit(rot61(coefficient<NT>(small, ch, kp >> 3, col), 8 * (kp & 7)), dg);
		for (int off = 32; off; off >>= 1)
			sum += __shfl_xor(sum, off, 64);
		if ((t & 63) == 0)
			init[e] = 128 * sum + (1 << 24);
	}
}

/* the 2^24 added to each of the 8 digit sums, with their weights 2^(8 t): minus their sum mod 2^61 - 1, added at the end */
constexpr u64 ortho_bias_correction(void)
{
	const u64 P = (1ull << 61) - 1;
	unsigned __int128 bias = 0;
	for (int dg = 0; dg < 8; dg++)
		bias += (unsigned __int128)(1u << 24) << (8 * dg);
	const u64 r = (u64)(bias % P);
	return r ? P - r : 0;
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
	/* value = L + H * 2^32 (L < 2^61, H < 2^56; here both < 2^50): fold mod 2^61 - 1 */
	const u64 P = (1ull << 61) - 1;
	const u64 lo = L + (H << 32), carry = lo < L;
	const u64 hi = (H >> 32) + carry;
	u64 x = (lo & P) + ((lo >> 61) | (hi << 3));	/* hi < 2^25: (hi << 3) < 2^28 */
	x = (x & P) + (x >> 61);
	return x >= P ? x - P : x;
}

/* What a wavefront holds of one 16-row tile between the loads and their use: the rows of v and p, and the terms that are
 * not products, already in the accumulator layout (row 4 h + reg, column col).
 * Staged form (n = 16, 128-byte rows): the rows of v, p AND Av are loaded as they lie in memory (lane l has bytes 16 l ..
 * 16 l + 15 of the tile's first KB, then of the second, so every load instruction covers whole lines), pass through a
 * padded per-wavefront LDS area to become A fragments and accumulator-layout terms, and the results go back the same way
 * (16-byte stores).  Loading fragments directly makes each quarter-wavefront touch 16 lines for 16 bytes each, four times
 * per row, and that kernel got SLOWER with more wavefronts per CU (2.32 ms at 12, 1.95 ms at 8); this one gets faster
 * (1.77 ms at 8, 1.40 ms at 16 = the rate of a bare streaming kernel with the same traffic).
 * Direct form (n = 8, 64-byte rows): a fragment load covers 1 KB of consecutive rows anyway; staging gives nothing there. */
template <int NT>
struct OrthoTile {
	static constexpr int NLD = NT / 8;		/* 16-byte loads per lane and block: 16 rows x 8 NT bytes / 1 KB */
	v4i a[2 * NLD];
	v4i c[NLD];		/* staged form: the rows of Av, as they lie */
	u64 b1[4], b2[4];	/* direct form: the terms that are not products */
};

template <int NT, bool ST>
MODP_DEV void ortho_tile_load(OrthoTile<NT> &R, const u64 *V, const u64 *AV, const u64 *Pb, long long rows, long long tile,
			      int lane, int h, int jout, bool is_p, bool dj)
{
	constexpr int ROWB = NT * 8, NLD = OrthoTile<NT>::NLD;
	const long long r0 = tile << 4;
#pragma unroll
	for (int q = 0; q < NLD; q++) {
		/* staged: byte o of the tile; direct: row m = lane & 15, bytes 16 h .. of its 64-byte part q */
		const int o = ST ? 16 * lane + 1024 * q : (lane & 15) * ROWB + 64 * q + 16 * h;
		long long rr = r0 + o / ROWB;
		rr = rr < rows ? rr : rows - 1;		/* rows past the end: any valid address, the results are not stored */
		const size_t at = (size_t)rr * ROWB + (o % ROWB);
		R.a[q] = *(const v4i *)((const unsigned char *)V + at);
		R.a[NLD + q] = *(const v4i *)((const unsigned char *)Pb + at);
		if (ST)
			R.c[q] = *(const v4i *)((const unsigned char *)AV + at);
	}
	if (ST)
		return;		/* the other terms come out of the staged rows */
#pragma unroll
	for (int reg = 0; reg < 4; reg++) {
		long long rr = r0 + 4 * h + reg;
		rr = rr < rows ? rr : rows - 1;
		const size_t at = (size_t)rr * NT + jout;
		if (NT == 16) {
			R.b1[reg] = dj ? AV[at] : V[at];
			R.b2[reg] = dj ? 0 : Pb[at];
		} else {
			R.b1[reg] = is_p ? (dj ? 0 : Pb[at]) : (dj ? AV[at] : V[at]);
			R.b2[reg] = 0;
		}
	}
}

template <int NT, bool ST>
__global__ void __launch_bounds__((OG<NT, ST>::THREADS))
k_ortho_mfma(u64 *__restrict__ V, const u64 *__restrict__ AV, u64 *__restrict__ Pb, long long rows,
	     const u64 *__restrict__ small, const unsigned char *__restrict__ img, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	using G = OG<NT, ST>;
	constexpr bool PF = G::PREFETCH;
	constexpr int NN = NT * NT;
	constexpr int ROWB = NT * 8, NLD = OrthoTile<NT>::NLD;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	{
		const uint4 *src = (const uint4 *)img;
		uint4 *dst = (uint4 *)lds;
		for (int i = threadIdx.x; i < (int)(G::IMG_BYTES / 16); i += (int)blockDim.x)
			dst[i] = src[i];
	}
	__syncthreads();
	const v4i *Bl = (const v4i *)lds;
	const int *init = (const int *)(lds + G::B_BYTES);
	constexpr u64 fconst = ortho_bias_correction();
	const u64 PR = (1ull << 61) - 1;
	u32 one;
	asm volatile("s_mov_b32 %0, 1" : "=s"(one));	/* 1, opaque to the optimiser (see mad_u64) */
	const int lane = threadIdx.x & 63, m = lane & 15, h = lane >> 4, col = m;
	/* this wavefront's staging area: the tile's rows of v, then of p, each row padded by 16 bytes so that the 16 rows a
	 * fragment read touches fall on different banks */
	unsigned char *stage = lds + G::IMG_BYTES + (threadIdx.x >> 6) * G::STAGE_BYTES;
	const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
	const long long ntiles = (rows + 15) >> 4;
	/* which output this lane's tile column is, and its selector d */
	const int jout = NT == 16 ? col : (col & 7);
	const bool is_p = NT == 8 && col >= 8;
	const bool dj = small[3 * NN + jout] != 0;
	OrthoTile<NT> R;
	if (PF && wave < ntiles)
		ortho_tile_load<NT, ST>(R, V, AV, Pb, rows, wave, lane, h, jout, is_p, dj);
	for (long long tile = wave; tile < ntiles; tile += nwaves) {
		const long long r0 = tile << 4;
		if (!PF)
			ortho_tile_load<NT, ST>(R, V, AV, Pb, rows, tile, lane, h, jout, is_p, dj);
		/* A fragments: row m, 16 bytes per K-step, biased to signed */
		v4i A[G::KS1];
		if (G::STAGED) {
			/* rows -> LDS as loaded, and back in fragment order */
#pragma unroll
			for (int q = 0; q < NLD; q++) {
				const int o = 16 * lane + 1024 * q;
				const int at = (o / ROWB) * G::RSTR + (o % ROWB);
				*(v4i *)(stage + at) = R.a[q];
				*(v4i *)(stage + 16 * G::RSTR + at) = R.a[NLD + q];
				*(v4i *)(stage + 32 * G::RSTR + at) = R.c[q];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int q = 0; q < G::KS1; q++) {
				const int arr = q / NLD, part = q % NLD;		/* 0 = v, 1 = p; 64-byte part of the row */
				A[q] = *(const v4i *)(stage + arr * 16 * G::RSTR + m * G::RSTR + 64 * part + 16 * h);
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		} else {
#pragma unroll
			for (int q = 0; q < G::KS1; q++)
				A[q] = R.a[q];
		}
#pragma unroll
		for (int q = 0; q < G::KS1; q++)
			A[q] ^= (v4i){ (int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080 };
		u64 base1[4], base2[4];
		if (!G::STAGED) {
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				base1[reg] = R.b1[reg];
				base2[reg] = R.b2[reg];
			}
		}
		/* the next tile's loads fly during this tile's arithmetic */
		if (PF && tile + nwaves < ntiles)
			ortho_tile_load<NT, ST>(R, V, AV, Pb, rows, tile + nwaves, lane, h, jout, is_p, dj);
		u64 L1[4] = { 0, 0, 0, 0 }, H1[4] = { 0, 0, 0, 0 }, L2[4] = { 0, 0, 0, 0 }, H2[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int s = 0; s < G::ND; s++) {
			const int i1 = init[(0 * G::ND + s) * 16 + col];
			v4i acc1 = { i1, i1, i1, i1 };
#pragma unroll
			for (int ks = 0; ks < G::KS1; ks++)
				acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks], Bl[(ks * G::ND + s) * 64 + lane], acc1, 0, 0, 0);
			v4i acc2 = { 0, 0, 0, 0 };
			if (NT == 16) {
				const int i2 = init[(1 * G::ND + s) * 16 + col];
				acc2 = (v4i){ i2, i2, i2, i2 };
#pragma unroll
				for (int ks = 0; ks < G::KS2; ks++)
					acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks], Bl[((G::KS1 + ks) * G::ND + s) * 64 + lane], acc2, 0, 0, 0);
			}
			const u32 mul = one << (s < 4 ? 8 * s : 8 * (s - 4));
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				/* one v_mad_u64_u32 per digit sum (the compiler would shift and add: three instructions) */
				if (s < 4) {
					mad_u64(L1[reg], (u32)acc1[reg], mul);
					if (NT == 16)
						mad_u64(L2[reg], (u32)acc2[reg], mul);
				} else {
					mad_u64(H1[reg], (u32)acc1[reg], mul);
					if (NT == 16)
						mad_u64(H2[reg], (u32)acc2[reg], mul);
				}
			}
			/* keep the digit sums sequential: without this the compiler hoists all the fragment reads (and spills) */
			asm volatile("" ::: "memory");
		}
		if (G::STAGED) {
			/* the terms that are not products come out of the staged rows (they are still there; 16 registers less
			 * across the arithmetic); then results -> LDS in the accumulator layout (row 4 h + reg, column col), and out
			 * as the rows lie: 16 bytes per lane, whole lines per instruction */
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				const int at = (4 * h + reg) * G::RSTR + 8 * col;
				if (NT == 16) {
					base1[reg] = *(const u64 *)(stage + (dj ? 32 * G::RSTR : 0) + at);
					base2[reg] = dj ? 0 : *(const u64 *)(stage + 16 * G::RSTR + at);
				} else {
					/* columns 0..7 of the tile are v', 8..15 p' */
					const int at8 = (4 * h + reg) * G::RSTR + 8 * jout;
					base1[reg] = is_p ? (dj ? 0 : *(const u64 *)(stage + 16 * G::RSTR + at8))
							  : *(const u64 *)(stage + (dj ? 32 * G::RSTR : 0) + at8);
					base2[reg] = 0;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				const int at = (4 * h + reg) * G::RSTR + 8 * col;
				u64 x = fold61(L1[reg], H1[reg]);
				x = addmod(x, fconst, PR);
				x = addmod(x, base1[reg], PR);
				if (NT == 16) {
					u64 y = fold61(L2[reg], H2[reg]);
					y = addmod(y, fconst, PR);
					y = addmod(y, base2[reg], PR);
					*(u64 *)(stage + at) = x;
					*(u64 *)(stage + 16 * G::RSTR + at) = y;
				} else {
					*(u64 *)(stage + (is_p ? 16 * G::RSTR : 0) + (4 * h + reg) * G::RSTR + 8 * jout) = x;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int q = 0; q < NLD; q++) {
				const int o = 16 * lane + 1024 * q;
				const int at = (o / ROWB) * G::RSTR + (o % ROWB);
				const v4i xv = *(const v4i *)(stage + at), yv = *(const v4i *)(stage + 16 * G::RSTR + at);
				const long long rr = r0 + o / ROWB;
				if (rr < rows) {
					const size_t to = (size_t)rr * ROWB + (o % ROWB);
					*(v4i *)((unsigned char *)V + to) = xv;
					*(v4i *)((unsigned char *)Pb + to) = yv;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			continue;
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

/* threads per workgroup and workgroups per CU of k_ortho_mfma<NT, ST> (registers and LDS decide), asked once per device */
template <int NT, bool ST>
static void ortho_mfma_shape(int *threads, int *per_cu)
{
	using G = OG<NT, ST>;
	/* contexts of one process may live on several threads (loopback ranks, one thread per GPU in the CLI): the answer is the
	 * same whoever computes it, the flag (k > 0) is published after the value it guards */
	static std::atomic<int> cached_t[64], cached_k[64];
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev >= 0 && dev < 64) {
		const int k_ = cached_k[dev].load(std::memory_order_acquire);
		if (k_ > 0) {
			*threads = cached_t[dev].load(std::memory_order_relaxed);
			*per_cu = k_;
			return;
		}
	}
	int t = G::THREADS;
	if (const char *e = getenv("BLZ_MFMA_BLOCK"))	/* experiments */
		if (atoi(e) >= 64 && atoi(e) <= G::THREADS && atoi(e) % 64 == 0)
			t = atoi(e);
	const void *fn = (const void *)k_ortho_mfma<NT, ST>;
	const size_t lds = G::lds_bytes(t);
	int k = 0;
	/* more than 64 KB of dynamic LDS needs the attribute, once per DEVICE (several contexts of one process) */
	if (lds > 65536)
		(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, fn, t, lds) != hipSuccess || k < 1)
		k = 1;
	const char *e = getenv("BLZ_MFMA_PER_CU");	/* experiments */
	if (e && atoi(e) > 0 && atoi(e) < k)
		k = atoi(e);
	if (dev >= 0 && dev < 64) {
		cached_t[dev].store(t, std::memory_order_relaxed);
		cached_k[dev].store(k, std::memory_order_release);
	}
	*threads = t;
	*per_cu = k;
}

template <int NT, bool ST>
static void ortho_mfma_go(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows, const u64 *small, const DevCtl *ctl,
			  hipStream_t s, bool img_ready)
{
	int threads = 0, per_cu = 0;
	ortho_mfma_shape<NT, ST>(&threads, &per_cu);
	const long long ntiles = (rows + 15) / 16, wpb = threads / 64;
	long long blocks = (ntiles + wpb - 1) / wpb;
	/* a persistent grid of as many workgroups as fit beside the B image in LDS (49 KB at n = 16, 16 KB at n = 8) */
	const long long cap = (long long)c.num_cu * per_cu;
	blocks = blocks > cap ? cap : blocks;
	unsigned char *img = (unsigned char *)c.mfma_img;
	if (!img_ready)		/* inside the iteration the semi-inverse kernel has built the image already */
		hipLaunchKernelGGL((k_ortho_mfma_prep<NT>), dim3(OG<NT>::NCH * OG<NT>::ND * 16 / 4), dim3(256), 0, s, small, img, ctl);
	const size_t lds = OG<NT, ST>::lds_bytes(threads);
	hipLaunchKernelGGL((k_ortho_mfma<NT, ST>), dim3((unsigned)blocks), dim3(threads), lds, s, (u64 *)V, (const u64 *)AV, (u64 *)P,
			   (long long)rows, small, img, ctl);
}

hipError_t launch_orthogonalize_mfma(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows, const u64 *small,
				     const DevCtl *ctl, hipStream_t s, bool img_ready)
{
	if (rows <= 0)
		return hipSuccess;
	/* n = 8 (64-byte rows: a fragment load covers 1 KB of consecutive rows anyway) loads fragments straight from HBM:
	 * the staged form is no faster there (115.6 / 116.0 against 113.9 / 114.6 us on the GL7d19 shape); BLZ_MFMA_STAGE8=1
	 * takes it for A/B */
	if (c.n == 16)
		ortho_mfma_go<16, true>(c, V, AV, P, rows, small, ctl, s, img_ready);
	else if (c.mfma_stage8)
		ortho_mfma_go<8, true>(c, V, AV, P, rows, small, ctl, s, img_ready);
	else
		ortho_mfma_go<8, false>(c, V, AV, P, rows, small, ctl, s, img_ready);
	return hipGetLastError();
}

/* ------------------------------------------------------------------- block_dot_products on the matrix cores */

/*
 * block_dot_products(), sequential/lanczos_modp.c:443-453: vtAv = V^T Av and vtAAv = Av^T Av, two n x n sums over all
 * block rows.  The same digit contraction as above with the ROWS as the K dimension: for digit a of the left operand and
 * digit b of the right one, C_ab[i][j] = sum over 64 rows of (byte a of X[r,i] - 128) * (byte b of Y[r,j] - 128) is one
 * v_mfma_i32_16x16x64_i8, accumulated by s = a + b (15 accumulators per product).  The operands need a transposition --
 * a lane holds 16 rows of ONE column (16 coalesced 8-byte loads) and gathers byte a of each into a 16-byte fragment with
 * a 4 x 4 byte transpose of v_perm_b32 -- and both sides are biased, which the algebra turns into COLUMN SUMS:
 *     sum_r X[r,i] Y[r,j] = sum_s 2^(8s) sum_{a+b=s} C_ab[i][j] + 128 W (colsum_X[i] + colsum_Y[j]) - 16384 R W^2
 * with W = sum_{a<8} 2^(8a) and R the rows processed (rows past the end count as zero rows).  The i32 accumulators start at
 * 2^30 and are folded mod 2^61-1 into 64-bit residues every 64 tiles (4096 rows), before they can leave [2^29, 3 * 2^29].
 * n = 16: X = V, Y = Av for vtAv and X = Y = Av for vtAAv (128 MFMAs per 64 rows).  n = 8: X = Y = [V | Av] (16 columns),
 * one product whose off-diagonal and lower-right 8 x 8 blocks are vtAv and vtAAv (64 MFMAs per 64 rows).
 * Output: one partial row of 2 n^2 residues per workgroup, as k_block_dot_fast writes them (k_dot_finalize sums them).
 */
#define DBLOCK 256

/* fragments of the 8 base-256 digits of 16 words: digit a of word q goes to byte q of frag[a] (4 dwords), biased by 128 */
MODP_DEV void digit_fragments(const u64 *x, v4i *frag)
{
#pragma unroll
	for (int d = 0; d < 4; d++) {
#pragma unroll
		for (int half = 0; half < 2; half++) {
			u32 w0 = (u32)(x[4 * d + 0] >> (32 * half)), w1 = (u32)(x[4 * d + 1] >> (32 * half));
			u32 w2 = (u32)(x[4 * d + 2] >> (32 * half)), w3 = (u32)(x[4 * d + 3] >> (32 * half));
			/* 4 x 4 byte transpose: t = bytes {w0.b0 w1.b0 w0.b1 w1.b1} ..., then pairs of 16-bit halves */
			const u32 t01l = __builtin_amdgcn_perm(w1, w0, 0x05010400u);	/* w0.b0 w1.b0 w0.b1 w1.b1 */
			const u32 t01h = __builtin_amdgcn_perm(w1, w0, 0x07030602u);	/* w0.b2 w1.b2 w0.b3 w1.b3 */
			const u32 t23l = __builtin_amdgcn_perm(w3, w2, 0x05010400u);
			const u32 t23h = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
			const u32 a0 = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);	/* digit 4*half + 0 of words 4d..4d+3 */
			const u32 a1 = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
			const u32 a2 = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
			const u32 a3 = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
			frag[4 * half + 0][d] = (int)(a0 ^ 0x80808080u);
			frag[4 * half + 1][d] = (int)(a1 ^ 0x80808080u);
			frag[4 * half + 2][d] = (int)(a2 ^ 0x80808080u);
			frag[4 * half + 3][d] = (int)(a3 ^ 0x80808080u);
		}
	}
}

/* the same for two digits only: digits 2 * pr and 2 * pr + 1 (pr = 0..3) of the 16 words -> out[0], out[1].  Eight calls
 * cost the perms of one digit_fragments(); the n = 16 inner products use it to hold two fragments of the left operand at
 * a time instead of eight (24 registers less: two wavefronts per SIMD instead of one). */
template <int PR_>
MODP_DEV void digit_fragment_pair(const u64 *x, v4i *out)
{
	constexpr int half = PR_ >> 1, hi = PR_ & 1;	/* 32-bit half of the word; bytes 0,1 or 2,3 of it */
#pragma unroll
	for (int d = 0; d < 4; d++) {
		const u32 w0 = (u32)(x[4 * d + 0] >> (32 * half)), w1 = (u32)(x[4 * d + 1] >> (32 * half));
		const u32 w2 = (u32)(x[4 * d + 2] >> (32 * half)), w3 = (u32)(x[4 * d + 3] >> (32 * half));
		const u32 t01 = __builtin_amdgcn_perm(w1, w0, hi ? 0x07030602u : 0x05010400u);
		const u32 t23 = __builtin_amdgcn_perm(w3, w2, hi ? 0x07030602u : 0x05010400u);
		out[0][d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
		out[1][d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u);
	}
}

MODP_DEV u64 shfl64m(u64 x, int src)
{
	const u32 lo = (u32)__shfl((int)(u32)x, src, 64), hi = (u32)__shfl((int)(u32)(x >> 32), src, 64);
	return ((u64)hi << 32) | lo;
}

MODP_DEV u64 fold_partial61(u64 s)	/* s < 2^64 -> < 2^62, congruent mod 2^61 - 1 */
{
	return (s & ((1ull << 61) - 1)) + (s >> 61);
}

template <int NT>
__global__ void __launch_bounds__(DBLOCK, 2)	/* two wavefronts per SIMD: 256 registers (accumulators included) */
k_block_dot_mfma(const u64 *__restrict__ V, const u64 *__restrict__ AV, long long rows, u64 *__restrict__ partial,
		 const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NP = NT == 16 ? 2 : 1, WAVES = DBLOCK / 64, NN = NT * NT;
	const u64 PR = (1ull << 61) - 1;
	__shared__ u64 red[WAVES][NP][4][64];
	__shared__ u64 csum[WAVES][2][64];
	u32 one;
	asm volatile("s_mov_b32 %0, 1" : "=s"(one));
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m = lane & 15, h = lane >> 4;
	const long long wave = (long long)blockIdx.x * WAVES + wv, nwaves = (long long)gridDim.x * WAVES;
	const long long ntiles = (rows + 63) >> 6;
	v4i acc[NP][15];
	u64 res[NP][4];
#pragma unroll
	for (int p_ = 0; p_ < NP; p_++) {
#pragma unroll
		for (int s = 0; s < 15; s++)
			acc[p_][s] = (v4i){ 1 << 30, 1 << 30, 1 << 30, 1 << 30 };
#pragma unroll
		for (int reg = 0; reg < 4; reg++)
			res[p_][reg] = 0;
	}
	u64 cs0 = 0, cs1 = 0;		/* column sums (mod p, lazily folded) of this lane's column of V-side / Av-side operand */
	long long nflush = 0, ntl = 0;
	int pending = 0;
	auto flush = [&]() {
#pragma unroll
		for (int p_ = 0; p_ < NP; p_++) {
#pragma unroll
			for (int reg = 0; reg < 4; reg++) {
				u64 L = 0, H = 0;
#pragma unroll
				for (int s = 0; s < 15; s++) {
					const int sh = fold_shift(s);
					const u32 mul = one << (sh < 32 ? sh : sh - 32);
					if (sh < 32)
						mad_u64(L, (u32)acc[p_][s][reg], mul);
					else
						mad_u64(H, (u32)acc[p_][s][reg], mul);
				}
				/* L < 8 * 2^31 * 2^27 = 2^61, H likewise < 2^61: fold H first so that fold61's bounds hold */
				const u64 x = fold61(L, 0), y = fold61(H, 0);
				/* y * 2^32 mod p */
				const unsigned __int128 t = (unsigned __int128)y << 32;
				u64 z = ((u64)t & PR) + (u64)(t >> 61);
				z = (z & PR) + (z >> 61);
				z = z >= PR ? z - PR : z;
				res[p_][reg] = addmod(res[p_][reg], addmod(x, z, PR), PR);
			}
#pragma unroll
			for (int s = 0; s < 15; s++)
				acc[p_][s] = (v4i){ 1 << 30, 1 << 30, 1 << 30, 1 << 30 };
		}
		nflush++;
		pending = 0;
	};
	for (long long tile = wave; tile < ntiles; tile += nwaves) {
		const long long r0 = (tile << 6) + 16 * h;
		u64 xv[16], xa[16];
#pragma unroll
		for (int q = 0; q < 16; q++) {
			const long long r = r0 + q;
			if (NT == 16) {
				xv[q] = r < rows ? V[(size_t)r * 16 + m] : 0;
				xa[q] = r < rows ? AV[(size_t)r * 16 + m] : 0;
			} else {
				const u64 *src = m < 8 ? V : AV;
				xv[q] = r < rows ? src[(size_t)r * 8 + (m & 7)] : 0;
			}
		}
		/* column sums, folded every four additions (values < 2^61) */
#pragma unroll
		for (int q = 0; q < 16; q += 4) {
			cs0 = fold_partial61(cs0 + xv[q] + xv[q + 1] + xv[q + 2] + xv[q + 3]);
			if (NT == 16)
				cs1 = fold_partial61(cs1 + xa[q] + xa[q + 1] + xa[q + 2] + xa[q + 3]);
		}
		if (NT == 16) {
			/* Av^T Av and V^T Av share the right operand; the left one's fragments are made two digits at a time */
			v4i fa[8];
			digit_fragments(xa, fa);
#pragma unroll
			for (int a = 0; a < 8; a++) {
#pragma unroll
				for (int b = 0; b < 8; b++)
					acc[NP - 1][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fa[b], acc[NP - 1][a + b], 0, 0, 0);
			}
			auto left = [&](auto pr) {
				constexpr int PR_ = decltype(pr)::value;
				v4i fv2[2];
				digit_fragment_pair<PR_>(xv, fv2);
#pragma unroll
				for (int b = 0; b < 8; b++) {
					acc[0][2 * PR_ + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv2[0], fa[b], acc[0][2 * PR_ + b], 0, 0, 0);
					acc[0][2 * PR_ + 1 + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv2[1], fa[b], acc[0][2 * PR_ + 1 + b], 0, 0, 0);
				}
			};
			left(std::integral_constant<int, 0>{});
			left(std::integral_constant<int, 1>{});
			left(std::integral_constant<int, 2>{});
			left(std::integral_constant<int, 3>{});
		} else {
			v4i fv[8];
			digit_fragments(xv, fv);
#pragma unroll
			for (int a = 0; a < 8; a++) {
#pragma unroll
				for (int b = 0; b < 8; b++)
					acc[0][a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fv[a], fv[b], acc[0][a + b], 0, 0, 0);
			}
		}
		ntl++;
		if (++pending == 64)
			flush();
	}
	if (pending)
		flush();
	/* the 2^30 every accumulator started from, per flush: nflush * 2^30 * sum_s 2^(8 s mod 61) */
	u64 biasw = 0;
	for (int s = 0; s < 15; s++) {
		const unsigned __int128 t = (unsigned __int128)(1u << 30) << fold_shift(s);
		biasw = addmod(biasw, (u64)(t % PR), PR);
	}
	const u64 nfl = (u64)(nflush % (long long)PR);
	const u64 bias_total = mulmod<61>(biasw, nfl, ModP{ PR, 0, 61, 32 });
	/* per wave: residues in the accumulator layout (row i = 4 h + reg, column j = lane & 15) and the column sums */
#pragma unroll
	for (int p_ = 0; p_ < NP; p_++)
#pragma unroll
		for (int reg = 0; reg < 4; reg++)
			red[wv][p_][reg][lane] = addmod(res[p_][reg], bias_total ? PR - bias_total : 0, PR);
	{
		u64 c0 = fold61(cs0, 0), c1 = fold61(cs1, 0);
		csum[wv][0][lane] = c0;
		csum[wv][1][lane] = c1;
	}
	__shared__ long long tiles_sh[WAVES];
	if (lane == 0)
		tiles_sh[wv] = ntl;
	__syncthreads();
	if (wv != 0)
		return;
	/* wavefront 0: sum over the wavefronts, add the bias terms, write the partial row */
	long long tl = 0;
	for (int w = 0; w < WAVES; w++)
		tl += tiles_sh[w];
	const ModP mp = { PR, 0, 61, 32 };
	/* W = sum_{a<8} 2^(8a) mod p, K1 = 128 W, K2 = 16384 W^2, R = 64 rows per tile */
	u64 W = 0;
	for (int a = 0; a < 8; a++) {
		const unsigned __int128 t = (unsigned __int128)1 << (8 * a);
		W = addmod(W, (u64)(t % PR), PR);
	}
	const u64 K1 = mulmod<61>(W, 128, mp), K2 = mulmod<61>(mulmod<61>(W, W, mp), 16384, mp);
	const u64 Rm = (u64)((tl * 64) % (long long)PR);
	const u64 k2r = mulmod<61>(K2, Rm, mp);
	/* column sums over the 4 k-blocks (lanes m, m+16, m+32, m+48) and the wavefronts */
	u64 col0 = 0, col1 = 0;		/* of column (lane & 15) */
	for (int w = 0; w < WAVES; w++)
		for (int hh = 0; hh < 4; hh++) {
			col0 = addmod(col0, csum[w][0][m + 16 * hh], PR);
			col1 = addmod(col1, csum[w][1][m + 16 * hh], PR);
		}
	u64 *out = partial + (size_t)blockIdx.x * 2 * NN;
#pragma unroll
	for (int p_ = 0; p_ < NP; p_++) {
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 4 * h + reg, j = m;
			u64 x = 0;
			for (int w = 0; w < WAVES; w++)
				x = addmod(x, red[w][p_][reg][lane], PR);
			/* operand sides: n = 16: product 0 = V^T Av (X = V-side sums, Y = Av-side), product 1 = Av^T Av; n = 8: X = Y = [V | Av] */
			/* lane i (h = 0, m = i) holds the sums of column i: the row index of this output needs them from there */
			const u64 r0c = shfl64m(col0, i), r1c = shfl64m(col1, i);
			const u64 sx = NT == 16 ? (p_ == 0 ? r0c : r1c) : r0c;
			const u64 sy = NT == 16 ? col1 : col0;
			x = addmod(x, mulmod<61>(K1, addmod(sx, sy, PR), mp), PR);
			x = addmod(x, k2r ? PR - k2r : 0, PR);
			if (NT == 16) {
				out[p_ * NN + i * 16 + j] = x;
			} else {
				if (i < 8 && j >= 8)
					out[i * 8 + (j - 8)] = x;		/* v^T Av */
				else if (i >= 8 && j >= 8)
					out[NN + (i - 8) * 8 + (j - 8)] = x;	/* Av^T Av */
			}
		}
	}
}

bool block_dot_mfma_supported(const KernelCfg &c)
{
	return c.mfma && c.word == 8 && c.mers == 61 && (c.n == 8 || c.n == 16);
}

hipError_t launch_block_dot_mfma(const KernelCfg &c, const void *V, const void *AV, int64_t rows, u64 *partial, int max_blocks,
				 int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	const long long ntiles = (rows + 63) / 64;
	long long blocks = (ntiles + DBLOCK / 64 - 1) / (DBLOCK / 64);
	const long long cap = std::min<long long>(max_blocks, (long long)c.num_cu * 2);	/* two wavefronts per SIMD */
	blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
	*nblocks = (int)blocks;
	if (c.n == 16)
		hipLaunchKernelGGL((k_block_dot_mfma<16>), dim3((unsigned)blocks), dim3(DBLOCK), 0, s, (const u64 *)V, (const u64 *)AV,
				   (long long)rows, partial, ctl);
	else
		hipLaunchKernelGGL((k_block_dot_mfma<8>), dim3((unsigned)blocks), dim3(DBLOCK), 0, s, (const u64 *)V, (const u64 *)AV,
				   (long long)rows, partial, ctl);
	return hipGetLastError();
}
