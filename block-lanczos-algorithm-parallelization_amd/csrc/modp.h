/*
 * modp.h -- arithmetic mod p for the gfx950 kernels (and for host-side set-up).
 *
 * Every quantity the reference computes is `(a + v*b) % prime` on u64
 * (sequential/lanczos_modp.c:284,300,313).  Here sums are kept UNREDUCED in 128 bits and
 * reduced once per output word; because the final value is the canonical residue of the same
 * exact integer, the words are bit-identical to the reference's.
 *
 * Two reducers, chosen once per context:
 *   MERS = 61 / 31   p = 2^61-1 / 2^31-1: shift-and-add folding.
 *   MERS = 0         any 2 <= p < 2^62: Barrett with mu = floor(2^(63+k) / p), k = bit length of p.
 *                    Valid for T < 2^(63+k); callers bound their sums accordingly
 *                    (SpMV: T < nnz_row * 2^32 * p, dense: see dense_chunk()).
 */
#ifndef BLZ_MODP_H
#define BLZ_MODP_H

#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#if defined(__HIPCC__)
#define MODP_HD __host__ __device__ __forceinline__
#define MODP_DEV __device__ __forceinline__
#else
#define MODP_HD static inline
#endif

struct ModP {
	u64 p;
	u64 mu;		/* floor(2^(63+k)/p), saturated */
	u32 k;		/* bit length of p */
	u32 chunk;	/* how many p*p products may be summed before a reduction */
};

static inline ModP make_modp(u64 p)
{
	ModP m;
	m.p = p;
	m.k = 64 - (u32)__builtin_clzll(p);
	const unsigned __int128 num = (unsigned __int128)1 << (63 + m.k);
	const unsigned __int128 q = num / p;
	m.mu = q > (unsigned __int128)~0ull ? ~0ull : (u64)q;
	/* dense sums: residue + chunk * p^2 must stay below 2^(63+k) (Barrett) and below 2^128 */
	const int room = 63 - (int)m.k;
	u64 c = room >= 6 ? 64 : ((1ull << room) - 1);
	if (c < 1)
		c = 1;
	if (p == ((1ull << 61) - 1))
		c = 32;		/* folding takes any 128-bit value: only 2^128 bounds the sum */
	m.chunk = (u32)c;
	return m;
}

static inline int modp_mersenne(u64 p)
{
	if (p == ((1ull << 61) - 1))
		return 61;
	if (p == ((1ull << 31) - 1))
		return 31;
	return 0;
}

#if defined(__HIPCC__)

/* 128-bit accumulator in two VGPR pairs. */
struct Acc {
	u64 lo, hi;
};

MODP_DEV void acc_zero(Acc &a) { a.lo = 0; a.hi = 0; }
MODP_DEV void acc_set(Acc &a, u64 x) { a.lo = x; a.hi = 0; }

/* a += v * x, v < 2^32, x < 2^64: two v_mad_u64_u32 plus carries. */
MODP_DEV void acc_mac32(Acc &a, u32 v, u64 x)
{
	const unsigned __int128 t = ((unsigned __int128)a.hi << 64 | a.lo) + (unsigned __int128)v * x;
	a.lo = (u64)t;
	a.hi = (u64)(t >> 64);
}

/* a += x (pattern matrices: every entry is 1) */
MODP_DEV void acc_add(Acc &a, u64 x)
{
	const u64 t = a.lo + x;
	a.hi += (t < x);
	a.lo = t;
}

/* a += x * y, both < 2^64 */
MODP_DEV void acc_mac64(Acc &a, u64 x, u64 y)
{
	const unsigned __int128 t = ((unsigned __int128)a.hi << 64 | a.lo) + (unsigned __int128)x * y;
	a.lo = (u64)t;
	a.hi = (u64)(t >> 64);
}

template <int MERS>
MODP_DEV u64 reduce128(u64 hi, u64 lo, const ModP &m)
{
	if (MERS == 61) {
		const u64 P = (1ull << 61) - 1;
		/* T = a + b*2^61 + c*2^122 and 2^61 = 1 (mod p) */
		const u64 a = lo & P;
		const u64 b = ((lo >> 61) | (hi << 3)) & P;
		const u64 c = hi >> 58;
		u64 s = a + b + c;		/* < 2^63 */
		s = (s & P) + (s >> 61);
		return s >= P ? s - P : s;
	} else if (MERS == 31) {
		const u64 P = (1ull << 31) - 1;
		/* five 31-bit digits (the top one has 4 bits) */
		u64 s = (lo & P) + ((lo >> 31) & P) + (((lo >> 62) | (hi << 2)) & P) + ((hi >> 29) & P) + (hi >> 60);
		s = (s & P) + (s >> 31);
		s = (s & P) + (s >> 31);
		return s >= P ? s - P : s;
	} else {
		/* Barrett: q = floor( floor(T / 2^(k-1)) * mu / 2^64 ) is within 3 of floor(T/p) */
		const u32 sh = m.k - 1;
		const u64 th = sh ? ((hi << (64 - sh)) | (lo >> sh)) : lo;
		const u64 q = __umul64hi(th, m.mu);
		u64 r = lo - q * m.p;		/* exact: the true remainder is < 4p < 2^64 */
		while (r >= m.p)
			r -= m.p;
		return r;
	}
}

template <int MERS>
MODP_DEV u64 acc_reduce(const Acc &a, const ModP &m)
{
	return reduce128<MERS>(a.hi, a.lo, m);
}

template <int MERS>
MODP_DEV u64 mulmod(u64 x, u64 y, const ModP &m)
{
	const unsigned __int128 t = (unsigned __int128)x * y;
	return reduce128<MERS>((u64)(t >> 64), (u64)t, m);
}

/*
 * Lazy 128-bit accumulator for sums of 64x64-bit products (the dense n x n work): the four 32x32 partial
 * products go to three independent 64-bit columns, each with one v_mad_u64_u32 and no cross-column carry
 * chain; wrap-arounds of the two low columns are counted instead of propagated.
 *     value = L + M*2^32 + (H + lc)*2^64 + mc*2^96
 * 7 VALU instructions per MAC (4 mad + 3 addc) against ~13 for the compiler's 128-bit sequence, which has to
 * shuffle odd register pairs (gfx950 wants 64-bit operands in even-aligned pairs).  Callers keep the true sum
 * below 2^128 (and H below 2^64: at most 2^64 / (p>>32)^2 products) between reductions -- the same chunk rule
 * as for Acc.  The carry-out of v_mad_u64_u32 is consumed by the next instruction through VCC, exactly like
 * the add_co/addc_co pairs hipcc emits for every 64-bit add.
 */
struct AccL {
	u64 L, M, H;
	u32 lc, mc;
};

MODP_DEV void acc_zero(AccL &a) { a.L = 0; a.M = 0; a.H = 0; a.lc = 0; a.mc = 0; }
MODP_DEV void acc_set(AccL &a, u64 x) { a.L = x; a.M = 0; a.H = 0; a.lc = 0; a.mc = 0; }

MODP_DEV void acc_mac64(AccL &a, u64 x, u64 y)
{
	const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
	asm("v_mad_u64_u32 %0, vcc, %5, %7, %0\n\t"
	    "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
	    "v_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
	    "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
	    "v_mad_u64_u32 %1, vcc, %6, %7, %1\n\t"
	    "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
	    "v_mad_u64_u32 %2, vcc, %6, %8, %2"
	    : "+v"(a.L), "+v"(a.M), "+v"(a.H), "+v"(a.lc), "+v"(a.mc)
	    : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
	    : "vcc");
}

template <int MERS>
MODP_DEV u64 acc_reduce(const AccL &a, const ModP &m)
{
	const u64 t = a.M << 32;
	const u64 lo = a.L + t;
	const u64 hi = a.H + a.lc + (a.M >> 32) + ((u64)a.mc << 32) + (lo < t);
	return reduce128<MERS>(hi, lo, m);
}

/*
 * Accumulator for sums of products of 32-bit residues (p < 2^32, the reference's own domain): 96 bits, one
 * v_mad_u64_u32 and one v_addc per MAC instead of the 128-bit sequence.  Same interface as Acc / AccL; the
 * operands arrive as u64 and MUST be below 2^32.  Up to 2^32 products between reductions.
 */
struct AccS {
	u64 lo;
	u32 hi;
};

MODP_DEV void acc_zero(AccS &a) { a.lo = 0; a.hi = 0; }
MODP_DEV void acc_set(AccS &a, u64 x) { a.lo = x; a.hi = 0; }

MODP_DEV void acc_mac64(AccS &a, u64 x, u64 y)
{
	asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
	    "v_addc_co_u32 %1, vcc, 0, %1, vcc"
	    : "+v"(a.lo), "+v"(a.hi)
	    : "v"((u32)x), "v"((u32)y)
	    : "vcc");
}

template <int MERS>
MODP_DEV u64 acc_reduce(const AccS &a, const ModP &m)
{
	return reduce128<MERS>((u64)a.hi, a.lo, m);
}

MODP_DEV u64 addmod(u64 x, u64 y, u64 p)
{
	const u64 s = x + y;		/* x, y < p < 2^62 */
	return s >= p ? s - p : s;
}

#endif /* __HIPCC__ */
#endif
