/*
 * blz_kernels.hip -- hand-written gfx950 kernels of the block-Lanczos-mod-p inner iteration.
 *
 * Integer, HBM-bound work: no MFMA.  64-lane wavefronts are cut into groups of G = 2^ceil(log2 n)
 * lanes; one group owns one block row (n consecutive words = one 32/64/128-byte sector), so every
 * gather of an X row and every store of a Y row is a single coalesced segment.
 *
 * Citations are relative to /root/reference/.
 */
#include "blz_kernels.h"
#include "ortho_img.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

#define BLOCK 256

/* 64-bit cross-lane read (absolute lane of the wave) */
MODP_DEV u64 shfl64(u64 x, int src)
{
	const u32 lo = (u32)__shfl((int)(u32)x, src, 64), hi = (u32)__shfl((int)(u32)(x >> 32), src, 64);
	return ((u64)hi << 32) | lo;
}

/* the same with the byte address of the source lane (lane * 4) precomputed by the caller: one ds_bpermute per half,
 * no per-call index arithmetic */
MODP_DEV u64 bperm64(u64 x, int src_lane_x4)
{
	const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(src_lane_x4, (int)(u32)x);
	const u32 hi = (u32)__builtin_amdgcn_ds_bpermute(src_lane_x4, (int)(u32)(x >> 32));
	return ((u64)hi << 32) | lo;
}

/* residues of a 32-bit prime need one ds_bpermute, not two */
template <bool NARROW>
MODP_DEV u64 bperm_word(u64 x, int src_lane_x4)
{
	if (NARROW)
		return (u64)(u32)__builtin_amdgcn_ds_bpermute(src_lane_x4, (int)(u32)x);
	return bperm64(x, src_lane_x4);
}

/* Broadcast of lane K of every lane group to the whole group as VALU moves (DPP row_newbcast, one 16-lane row =
 * one group at NT = 16; two half-row moves at NT = 8) -- no LDS crossbar traffic, no lgkmcnt wait. */
template <int K, int NT>
MODP_DEV u32 group_bcast32(u32 x)
{
	static_assert(NT == 16 || NT == 8, "row_newbcast covers groups of 8 or 16 lanes");
	if (NT == 16)
		return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + K, 0xF, 0xF, false);
	const int lo = __builtin_amdgcn_update_dpp(0, (int)x, 0x150 + (K & 7), 0xF, 0x3, false);
	return (u32)__builtin_amdgcn_update_dpp(lo, (int)x, 0x150 + 8 + (K & 7), 0xF, 0xC, false);
}

template <int K, int NT, bool NARROW>
MODP_DEV u64 group_bcast(u64 x)
{
	const u32 lo = group_bcast32<K, NT>((u32)x);
	if (NARROW)
		return lo;
	return ((u64)group_bcast32<K, NT>((u32)(x >> 32)) << 32) | lo;
}

/* lane i of every 16-lane row receives the word of lane (i + Q) mod 16: DPP row_ror by 16 - Q, one VALU move per
 * 32 bits */
template <int Q, bool NARROW>
MODP_DEV u64 row16_rotl(u64 x)
{
	static_assert(Q >= 1 && Q <= 15, "rotation inside a row of 16");
	const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)x, 0x120 + (16 - Q), 0xF, 0xF, false);
	if (NARROW)
		return lo;
	return ((u64)(u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(x >> 32), 0x120 + (16 - Q), 0xF, 0xF, false) << 32) | lo;
}

template <int K, int N, class F>
MODP_DEV void static_for(F &&f)
{
	if constexpr (K < N) {
		f(std::integral_constant<int, K>{});
		static_for<K + 1, N>(f);
	}
}

MODP_DEV u64 shfl_xor64(u64 x, int mask)
{
	const u32 lo = (u32)__shfl_xor((int)(u32)x, mask, 64), hi = (u32)__shfl_xor((int)(u32)(x >> 32), mask, 64);
	return ((u64)hi << 32) | lo;
}

/* ----------------------------------------------------------------------------- SpMV */

/*
 * sparse_matrix_vector_product(), sequential/lanczos_modp.c:266-287.
 * The reference scatters COO triplets with a `%` per term (:284).  Here each row of the CSR slab
 * is reduced by one group of lanes: 128-bit unreduced sum of val*x per column, one reduction per
 * output word, one coalesced row store.  No atomics, deterministic.
 */

/* acc += sum over entries [k, e) of val * X[col, xl]; `stride` = words per block row.
 * TAILB: the 1-3 entries left over after the batches of four go as ONE predicated batch (a slot past the end re-reads the
 * row's last entry and is switched off by a zero multiplier) instead of one dependent index load + gather per entry.  It
 * shortens the dependent chain of a row -- relat8 shape, 15 us products: 17.0 -> 15.8 us; structured workload, gathers that
 * hit: 663 -> 647 us -- and costs the fabric-bound uniform shapes their dead slots (GL7d19 shape: 656 -> 664 us), so the
 * slab's plan asks for it only where the gathers hit or the product is a few launches' worth of latency (round 3). */
template <typename W, bool TAILB = false>
MODP_DEV void spmv_accumulate(Acc &acc, u32 k, u32 e, const int *__restrict__ ci, const u32 *__restrict__ va,
			      const u32 *spal, const W *__restrict__ X, int stride, int xl)
{
	if (spal) {
		/* packed stream: one u32 per entry = column (24 bits) | index into the value palette (8 bits, in LDS) */
		for (; k + 4 <= e; k += 4) {
			const u32 p0 = (u32)ci[k], p1 = (u32)ci[k + 1], p2 = (u32)ci[k + 2], p3 = (u32)ci[k + 3];
			const W x0 = X[(size_t)(p0 & 0xFFFFFFu) * stride + xl], x1 = X[(size_t)(p1 & 0xFFFFFFu) * stride + xl];
			const W x2 = X[(size_t)(p2 & 0xFFFFFFu) * stride + xl], x3 = X[(size_t)(p3 & 0xFFFFFFu) * stride + xl];
			acc_mac32(acc, spal[p0 >> 24], x0);
			acc_mac32(acc, spal[p1 >> 24], x1);
			acc_mac32(acc, spal[p2 >> 24], x2);
			acc_mac32(acc, spal[p3 >> 24], x3);
		}
		if (TAILB) {
			if (k < e) {
				const u32 last = e - 1;
				const u32 q1 = k + 1 < e ? k + 1 : last, q2 = k + 2 < e ? k + 2 : last;
				const u32 p0 = (u32)ci[k], p1 = (u32)ci[q1], p2 = (u32)ci[q2];
				const W x0 = X[(size_t)(p0 & 0xFFFFFFu) * stride + xl], x1 = X[(size_t)(p1 & 0xFFFFFFu) * stride + xl];
				const W x2 = X[(size_t)(p2 & 0xFFFFFFu) * stride + xl];
				acc_mac32(acc, spal[p0 >> 24], x0);
				acc_mac32(acc, k + 1 < e ? spal[p1 >> 24] : 0u, x1);
				acc_mac32(acc, k + 2 < e ? spal[p2 >> 24] : 0u, x2);
			}
		} else {
			for (; k < e; k++) {
				const u32 pk = (u32)ci[k];
				acc_mac32(acc, spal[pk >> 24], X[(size_t)(pk & 0xFFFFFFu) * stride + xl]);
			}
		}
	} else if (va) {
		for (; k + 4 <= e; k += 4) {
			const int c0 = ci[k], c1 = ci[k + 1], c2 = ci[k + 2], c3 = ci[k + 3];
			const u32 a0 = va[k], a1 = va[k + 1], a2 = va[k + 2], a3 = va[k + 3];
			const W x0 = X[(size_t)c0 * stride + xl], x1 = X[(size_t)c1 * stride + xl];
			const W x2 = X[(size_t)c2 * stride + xl], x3 = X[(size_t)c3 * stride + xl];
			acc_mac32(acc, a0, x0);
			acc_mac32(acc, a1, x1);
			acc_mac32(acc, a2, x2);
			acc_mac32(acc, a3, x3);
		}
		if (TAILB) {
			if (k < e) {
				const u32 last = e - 1;
				const u32 q1 = k + 1 < e ? k + 1 : last, q2 = k + 2 < e ? k + 2 : last;
				const int c0 = ci[k], c1 = ci[q1], c2 = ci[q2];
				const u32 a0 = va[k], a1 = va[q1], a2 = va[q2];
				const W x0 = X[(size_t)c0 * stride + xl], x1 = X[(size_t)c1 * stride + xl], x2 = X[(size_t)c2 * stride + xl];
				acc_mac32(acc, a0, x0);
				acc_mac32(acc, k + 1 < e ? a1 : 0u, x1);
				acc_mac32(acc, k + 2 < e ? a2 : 0u, x2);
			}
		} else {
			for (; k < e; k++)
				acc_mac32(acc, va[k], X[(size_t)ci[k] * stride + xl]);
		}
	} else {
		for (; k + 4 <= e; k += 4) {
			const int c0 = ci[k], c1 = ci[k + 1], c2 = ci[k + 2], c3 = ci[k + 3];
			const W x0 = X[(size_t)c0 * stride + xl], x1 = X[(size_t)c1 * stride + xl];
			const W x2 = X[(size_t)c2 * stride + xl], x3 = X[(size_t)c3 * stride + xl];
			acc_add(acc, x0);
			acc_add(acc, x1);
			acc_add(acc, x2);
			acc_add(acc, x3);
		}
		if (TAILB) {
			if (k < e) {
				const u32 last = e - 1;
				const u32 q1 = k + 1 < e ? k + 1 : last, q2 = k + 2 < e ? k + 2 : last;
				const int c0 = ci[k], c1 = ci[q1], c2 = ci[q2];
				const W x0 = X[(size_t)c0 * stride + xl], x1 = X[(size_t)c1 * stride + xl], x2 = X[(size_t)c2 * stride + xl];
				acc_add(acc, x0);
				acc_add(acc, k + 1 < e ? (u64)x1 : 0ull);
				acc_add(acc, k + 2 < e ? (u64)x2 : 0ull);
			}
		} else {
			for (; k < e; k++)
				acc_add(acc, X[(size_t)ci[k] * stride + xl]);
		}
	}
}

MODP_DEV void acc_add_acc(Acc &a, u64 olo, u64 ohi)
{
	const u64 t = a.lo + olo;
	a.hi += ohi + (t < olo);
	a.lo = t;
}

/*
 * Outlier rows.  A row much longer than the average (more than A.heavy_thr = max(64, 4 x mean) entries) would keep
 * one lane group busy long after its neighbours have finished -- measured on lognormal row lengths, that imbalance,
 * not divergence inside a wavefront, is what uneven rows cost (tools/exp_skew.py).  Such rows are listed on the host
 * when the slab is uploaded (DevCsr::heavy); the streaming kernels skip them and a second, small launch
 * (k_spmv_heavy) gives each of them a whole workgroup -- several for a row of more than HEAVY_SEG entries, whose
 * 128-bit segment sums a third launch adds: every lane group sums a slice, the slices are added through LDS.  Being a launch of its own, it spreads the outliers over the chip wherever they sit in the row order (a
 * renumbering by smallest column puts all dense rows next to each other).
 */

/* all threads of the block: sum entries [k0, e0) over BLOCK/G slices; group 0 returns the 128-bit total */
template <typename W, int G>
MODP_DEV Acc heavy_range_sum(u32 k0, u32 e0, const int *__restrict__ ci, const u32 *__restrict__ va, const u32 *spal,
			     const W *__restrict__ X, int stride, int xl, Acc (*slices)[G])
{
	constexpr int GPB = BLOCK / G;
	const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
	const u32 per = (e0 - k0 + GPB - 1) / GPB;
	const u32 lo = k0 + (u32)grp * per;
	Acc acc;
	acc_zero(acc);
	spmv_accumulate<W>(acc, lo < e0 ? lo : e0, (lo + per) < e0 ? (lo + per) : e0, ci, va, spal, X, stride, xl);
	slices[grp][lane] = acc;
	__syncthreads();
	if (grp == 0)
		for (int g = 1; g < GPB; g++)
			acc_add_acc(acc, slices[g][lane].lo, slices[g][lane].hi);
	__syncthreads();
	return acc;
}

template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_heavy(const int *__restrict__ ci, const u32 *__restrict__ va, const u32 *__restrict__ pal,
	     const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd, const HeavySeg *__restrict__ segs,
	     int nseg, u64 *__restrict__ scratch, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
	     const DevCtl *__restrict__ ctl);
template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_heavy_combine(const HeavyRow *__restrict__ mrows, int nm, const u64 *__restrict__ scratch, W *__restrict__ Y,
		     const W *__restrict__ Vd, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
		     const DevCtl *__restrict__ ctl);

template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_wave(const u32 *__restrict__ rp, const int *__restrict__ ci, const u32 *__restrict__ va,
	    const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd,
	    const int *__restrict__ list, int nlist, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
	    const DevCtl *__restrict__ ctl);

/* blocks of the outlier launches: one wavefront per medium row (4 per block, bounded), one block per segment of a
 * long row (bounded), and one lane group per split row */
static inline long long medium_blocks(const KernelCfg &c, const DevCsr &A)
{
	const long long b = ((long long)A.n_medium + 3) / 4, cap = (long long)c.num_cu * 6;
	return b < cap ? b : cap;
}

static inline long long heavy_blocks(const KernelCfg &c, const DevCsr &A, long long room)
{
	long long b = A.n_heavy < (long long)c.num_cu * 4 ? A.n_heavy : (long long)c.num_cu * 4;
	return b < room ? b : room;
}

static inline long long combine_blocks(const DevCsr &A, int G)
{
	const long long b = ((long long)A.n_multi + BLOCK / G - 1) / (BLOCK / G);
	return b < 16 ? b : 16;
}

/* The outlier launches read the same operand and write rows the streaming kernel skips: nothing orders them against it.
 * They run on a side stream forked BEFORE the streaming kernel is enqueued (heavy_fork, at the top of every dispatch) and
 * joined after, so that they fill the fabric gaps of the streaming kernel instead of adding their own time behind it
 * (structured workload: k_spmv_wave + k_spmv_heavy were 229 us of a 740 us product). */
static void heavy_fork(const KernelCfg &c, const DevCsr &A, hipStream_t s)
{
	if (c.side && (A.n_heavy || A.n_medium))
		(void)hipEventRecord(c.ev_fork, s);
}

template <typename W, int G, int MERS, bool DOT>
static void launch_heavy(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum, u64 *partial,
			 int slot0, long long hb, const DevCtl *ctl, hipStream_t main_stream)
{
	hipStream_t s = main_stream;
	if (c.side) {
		s = c.side;
		(void)hipStreamWaitEvent(s, c.ev_fork, 0);
	}
	struct Join {
		const KernelCfg &c;
		hipStream_t side, main_stream;
		~Join()
		{
			if (c.side) {
				(void)hipEventRecord(c.ev_join, side);
				(void)hipStreamWaitEvent(main_stream, c.ev_join, 0);
			}
		}
	} join{ c, s, main_stream };
	if (hb)
		hipLaunchKernelGGL((k_spmv_heavy<W, G, MERS, DOT>), dim3((unsigned)hb), dim3(BLOCK), 0, s, A.col_idx, A.val,
				   A.palette, X, Y, Vd, A.heavy, A.n_heavy, A.heavy_scratch, c.n, accum, c.m, partial, slot0, ctl);
	const long long cb = A.n_multi ? combine_blocks(A, G) : 0;
	if (cb)
		hipLaunchKernelGGL((k_spmv_heavy_combine<W, G, MERS, DOT>), dim3((unsigned)cb), dim3(BLOCK),
				   0, s, A.heavy_multi, A.n_multi, A.heavy_scratch, Y, Vd, c.n, accum, c.m, partial,
				   slot0 + (int)hb, ctl);
	if (A.n_medium)
		hipLaunchKernelGGL((k_spmv_wave<W, G, MERS, DOT>), dim3((unsigned)medium_blocks(c, A)), dim3(BLOCK), 0, s,
				   A.row_ptr, A.col_idx, A.val, A.palette, X, Y, Vd, A.medium_rows, A.n_medium, c.n, accum, c.m,
				   partial, slot0 + (int)(hb + cb), ctl);
}

template <typename W, int G, int MERS, bool TAILB>
__global__ void __launch_bounds__(BLOCK)
k_spmv(const u32 *__restrict__ rp, const int *__restrict__ ci, const u32 *__restrict__ va,
       const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, long long rows, int n, int split_log2,
       int accum, u32 heavy, ModP m, XcdRows xr, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	__shared__ u32 spal_store[BLOCK];
	const u32 *spal = pal ? spal_store : nullptr;
	if (pal)
		spal_store[threadIdx.x] = pal[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & (G - 1);
	const int xl = lane < n ? lane : 0;
	/* 2^split_log2 adjacent groups of one wavefront share a row (few, long rows: keeps every CU busy and
	 * shortens the dependent chain); their 128-bit partial sums are added across lanes before the reduction */
	const long long gid = ((long long)blockIdx.x * BLOCK + threadIdx.x) / G;
	const long long g0 = gid >> split_log2;
	const u32 part = (u32)gid & ((1u << split_log2) - 1u);
	long long ng = ((long long)gridDim.x * (BLOCK / G)) >> split_log2;
	long long r = g0;
	if (xr.begin[0] >= 0) {
		/* per-XCD row ranges (matrices whose neighbouring rows share columns; split_log2 == 0, grid a multiple of 8):
		 * blocks b and b + 8 share an XCD, so each L2 serves one contiguous eighth of the rows */
		const int xcd = blockIdx.x & 7;
		r = xr.begin[xcd] + ((long long)(blockIdx.x >> 3) * BLOCK + threadIdx.x) / G;
		rows = xr.begin[xcd + 1];
		ng = (long long)(gridDim.x >> 3) * (BLOCK / G);
	}
	for (; r < rows; r += ng) {
		u32 k = rp[r], e = rp[r + 1];
		if (e - k > heavy)	/* left to k_spmv_heavy */
			continue;
		if (split_log2) {
			const u32 len = e - k, per = (len + (1u << split_log2) - 1u) >> split_log2;
			const u32 lo = k + part * per;
			k = lo < e ? lo : e;
			e = (lo + per) < e ? (lo + per) : e;
		}
		Acc acc;
		acc_zero(acc);
		spmv_accumulate<W, TAILB>(acc, k, e, ci, va, spal, X, n, xl);
		for (int off = G; off < (G << split_log2); off <<= 1)
			acc_add_acc(acc, shfl_xor64(acc.lo, off), shfl_xor64(acc.hi, off));
		if (lane < n && part == 0) {
			if (accum)		/* column-chunked product: this launch adds to what earlier pieces left in Y */
				acc_add(acc, Y[(size_t)r * n + lane]);
			Y[(size_t)r * n + lane] = (W)acc_reduce<MERS>(acc, m);
		}
	}
}

static int spmv_split_log2(const KernelCfg &c, int64_t rows, int64_t nnz)
{
	int G = 1;
	while (G < c.n)
		G <<= 1;
	const long long groups_per_block = BLOCK / G;
	const double avg = rows ? (double)nnz / (double)rows : 0.0;
	int split_log2 = 0;
	while ((G << (split_log2 + 1)) <= 64 && (double)rows * (1 << split_log2) < 2.0 * c.num_cu * 8 * groups_per_block
	       && avg / (1 << (split_log2 + 1)) >= 4.0)
		split_log2++;
	return split_log2;
}

u32 spmv_heavy_threshold(const KernelCfg &c, int64_t rows, int64_t nnz)
{
	const double avg = rows ? (double)nnz / (double)rows : 0.0;
	u32 base = (u32)(4.0 * avg < 64.0 ? 64.0 : 4.0 * avg);
	if (const char *e = getenv("BLZ_HEAVY_THR"))	/* experiments only */
		base = (u32)atoll(e);
	return base << spmv_split_log2(c, rows, nnz);
}

/* the same product with the matrix stream staged through LDS (further down: it shares DotState with k_spmv_dot) */
template <typename W, int MERS, bool DOT>
static hipError_t staged_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum,
				  u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s);

/* ... and with the densest block rows of the operand resident in LDS (k_spmv_panel, below) */
template <typename W, int MERS, bool DOT>
static hipError_t panel_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum,
				 u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s);

template <typename W, int MERS>
static hipError_t spmv_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, int accum, const DevCtl *ctl,
				hipStream_t s)
{
	if (A.rows == 0)
		return hipSuccess;
	heavy_fork(c, A, s);
	if (A.panel_rows > 0 && c.panel)
		return panel_dispatch<W, MERS, false>(c, A, X, Y, (const W *)nullptr, accum, (u64 *)nullptr, 0, (int *)nullptr, ctl, s);
	if (A.st_ok && c.staged)
		return staged_dispatch<W, MERS, false>(c, A, X, Y, (const W *)nullptr, accum, (u64 *)nullptr, 0, (int *)nullptr, ctl, s);
	int G = 1;
	while (G < c.n)
		G <<= 1;
	const long long groups_per_block = BLOCK / G;
	/* measured on MI355X (tools/tune_spmv.py): rows of >= ~12 entries run best with 4 resident blocks per CU,
	 * short rows want 8; few long rows are split over up to 64/G groups */
	const double avg = A.kept_mean >= 0.0 ? A.kept_mean : (double)A.nnz / (double)A.rows;	/* of the rows this launch takes */
	const int split_log2 = spmv_split_log2(c, A.rows, A.nnz);
	/* (widths of 16 and more are the exception: a wavefront then holds 4 lane groups or fewer, i.e. at most 16 line
	 * fills in flight with the 4-deep batch, and 16 waves per CU cannot cover the fabric latency: 8 blocks per CU,
	 * measured 1052 -> 724 us at n = 16 on the GL7d19 shape and the same on the config-5 shape) */
	const bool few_groups_per_wave = G >= 16;
	/* (and so are uneven row lengths: lane groups that have finished their row wait for the longest one of their
	 * wavefront and issue nothing meanwhile, so more resident wavefronts are needed to keep the fabric busy --
	 * tools/exp_skew.py, 4 -> 6 blocks per CU: -9 % on lognormal lengths) */
	int per_cu = c.spmv_blocks_per_cu > 0 ? c.spmv_blocks_per_cu
					      : ((avg / (1 << split_log2) >= 12.0 && !few_groups_per_wave) ? (A.uneven ? 6 : 4) : 8);
	long long blocks = ((A.rows << split_log2) + groups_per_block - 1) / groups_per_block;
	const long long cap = (long long)c.num_cu * per_cu;
	if (blocks > cap)
		blocks = cap;
	XcdRows xr;
	xr.begin[0] = -1;
	if (A.xcd_ranges && split_log2 == 0 && blocks >= 64) {
		for (int x = 0; x < 9; x++)
			xr.begin[x] = A.xr_rows[x];
		blocks = (blocks + 7) & ~7ll;
	}
#define SPMV_CASE(GG)                                                                                             \
	case GG:                                                                                                  \
		if (A.tail_batch)                                                                                 \
			hipLaunchKernelGGL((k_spmv<W, GG, MERS, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, A.row_ptr,  \
					   A.col_idx, A.val, A.palette, X, Y, (long long)A.rows, c.n, split_log2, accum,   \
					   A.heavy_thr, c.m, xr, ctl);                                                     \
		else                                                                                              \
			hipLaunchKernelGGL((k_spmv<W, GG, MERS, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, A.row_ptr, \
					   A.col_idx, A.val, A.palette, X, Y, (long long)A.rows, c.n, split_log2, accum,   \
					   A.heavy_thr, c.m, xr, ctl);                                                     \
		if (A.n_heavy || A.n_medium)                                                                      \
			launch_heavy<W, GG, MERS, false>(c, A, X, Y, (const W *)nullptr, accum, (u64 *)nullptr, 0,   \
							 heavy_blocks(c, A, 1 << 30), ctl, s);                        \
		break;
	switch (G) {
		SPMV_CASE(1)
		SPMV_CASE(2)
		SPMV_CASE(4)
		SPMV_CASE(8)
		SPMV_CASE(16)
		SPMV_CASE(32)
		SPMV_CASE(64)
	default:
		return hipErrorInvalidValue;
	}
#undef SPMV_CASE
	return hipGetLastError();
}

hipError_t launch_spmv(const KernelCfg &c, const DevCsr &A, const void *X, void *Y, int accum, const DevCtl *ctl,
		       hipStream_t s)
{
	if (c.word == 4)
		return c.mers == 31 ? spmv_dispatch<u32, 31>(c, A, (const u32 *)X, (u32 *)Y, accum, ctl, s)
				    : spmv_dispatch<u32, 0>(c, A, (const u32 *)X, (u32 *)Y, accum, ctl, s);
	return c.mers == 61 ? spmv_dispatch<u64, 61>(c, A, (const u64 *)X, (u64 *)Y, accum, ctl, s)
			    : spmv_dispatch<u64, 0>(c, A, (const u64 *)X, (u64 *)Y, accum, ctl, s);
}

/* --------------------------------------------------------------------- block_dot_products */

/*
 * block_dot_products() / matmul_CpAtB(), sequential/lanczos_modp.c:443-453, :305-315.
 * One thread per (i,j) pair and row slice; rows are walked once for both products.
 * NP = pairs per thread (n*n may exceed the block size for n > 16).
 */
template <typename W, int MERS, int NP>
__global__ void __launch_bounds__(BLOCK)
k_block_dot(const W *__restrict__ V, const W *__restrict__ AV, long long rows, long long rows_per_block, int n,
	    ModP m, u64 *__restrict__ partial, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	__shared__ u64 red[2][BLOCK];
	const int pairs = n * n;
	const int t = threadIdx.x;
	const int slices = NP == 1 ? BLOCK / pairs : 1;
	const int slice = NP == 1 ? t / pairs : 0;
	const bool live = slice < slices;
	const long long row0 = (long long)blockIdx.x * rows_per_block;
	long long row1 = row0 + rows_per_block;
	if (row1 > rows)
		row1 = rows;
	Acc a1[NP], a2[NP];
	int pi[NP], pj[NP];
#pragma unroll
	for (int q = 0; q < NP; q++) {
		const int e = NP == 1 ? t % pairs : t + q * BLOCK;
		pi[q] = e < pairs ? e / n : -1;
		pj[q] = e < pairs ? e % n : 0;
		acc_zero(a1[q]);
		acc_zero(a2[q]);
	}
	u32 cnt = 0;
	if (live)
		for (long long r = row0 + slice; r < row1; r += slices) {
			const W *vr = V + (size_t)r * n, *ar = AV + (size_t)r * n;
#pragma unroll
			for (int q = 0; q < NP; q++)
				if (pi[q] >= 0) {
					const u64 vi = vr[pi[q]], ai = ar[pi[q]], aj = ar[pj[q]];
					acc_mac64(a1[q], vi, aj);
					acc_mac64(a2[q], ai, aj);
				}
			if (++cnt == m.chunk) {
				cnt = 0;
#pragma unroll
				for (int q = 0; q < NP; q++) {
					acc_set(a1[q], acc_reduce<MERS>(a1[q], m));
					acc_set(a2[q], acc_reduce<MERS>(a2[q], m));
				}
			}
		}
	const int nn2 = 2 * pairs;
	u64 *out = partial + (size_t)blockIdx.x * nn2;
	if (NP == 1) {
		red[0][t] = live ? acc_reduce<MERS>(a1[0], m) : 0;
		red[1][t] = live ? acc_reduce<MERS>(a2[0], m) : 0;
		__syncthreads();
		if (t < pairs) {
			u64 s1 = 0, s2 = 0;
			for (int sl = 0; sl < slices; sl++) {
				s1 = addmod(s1, red[0][sl * pairs + t], m.p);
				s2 = addmod(s2, red[1][sl * pairs + t], m.p);
			}
			out[t] = s1;
			out[pairs + t] = s2;
		}
	} else {
#pragma unroll
		for (int q = 0; q < NP; q++)
			if (pi[q] >= 0) {
				const int e = t + q * BLOCK;
				out[e] = acc_reduce<MERS>(a1[q], m);
				out[pairs + e] = acc_reduce<MERS>(a2[q], m);
			}
	}
}

/*
 * Block inner products for n = NT in {1,2,4,8,16}: a group of NT lanes owns one block row, lane i holds
 * v[r,i] and Av[r,i], and the operands Av[r,(i+t) mod NT] arrive by rotation inside the group.  Lane i
 * accumulates
 *     vtAv [i][(i+t)%NT]  for t = 0..NT-1
 *     vtAAv[i][(i+t)%NT]  for t = 0..NT/2   (Av^T Av is symmetric for ANY input, so half is enough)
 * in 128-bit registers, reduced every m.chunk rows.  Used stand-alone (k_block_dot_fast) and as the epilogue
 * of the second SpMV (k_spmv_dot), where its VALU work hides under the gather latency.
 * SYM1: inside the iteration Av = B^T (B v), so v^T Av = (B v)^T (B v) is symmetric as well -- as a TOTAL over all rows,
 * not per workgroup.  Lane i then accumulates vtAv[i][(i+t)%NT] for t = 0..NT/2 only and the partial row carries the
 * value in both positions for 0 < t < NT/2 (the totals of the two positions are equal, so the sum over the partial rows
 * is the matrix the reference computes entry by entry); t = NT/2 is computed from both sides, each lane writes its own.
 * 10 MACs per lane and row instead of 13 at n = 8, and three accumulators less in the register-bound fused kernel.
 */
template <typename A, int MERS, int NT, int BS = BLOCK, bool SYM1 = false>
struct DotState {
	static constexpr int H = NT / 2 + 1, N1 = SYM1 ? H : NT, SLOTS = N1 + H, WAVES = BS / 64;
	A a1[N1], a2[H];
	u32 cnt;

	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int q = 0; q < N1; q++)
			acc_zero(a1[q]);
#pragma unroll
		for (int q = 0; q < H; q++)
			acc_zero(a2[q]);
		cnt = 0;
	}

	/* one block row: vi = v[r,i], ai = Av[r,i]; gbase = first lane of the group within the wavefront */
	__device__ __forceinline__ void row(u64 vi, u64 ai, int i, int gbase, const ModP &m)
	{
		if constexpr (NT == 16 && std::is_same<A, AccS>::value) {
			/* a group is one DPP row: rotate with VALU moves, not through the LDS crossbar (32-bit words: -10 %; at
			 * 64 bits the kernel is VALU-bound and the extra moves cost 3 %) */
			acc_mac64(a1[0], vi, ai);
			acc_mac64(a2[0], ai, ai);
			static_for<1, N1>([&](auto qc) {
				constexpr int q = decltype(qc)::value;
				const u64 aq = row16_rotl<q, std::is_same<A, AccS>::value>(ai);
				acc_mac64(a1[q], vi, aq);
				if constexpr (q < H)
					acc_mac64(a2[q], ai, aq);
			});
		} else {
#pragma unroll
			for (int q = 0; q < N1; q++) {
				const u64 aq = q == 0 ? ai : bperm_word<std::is_same<A, AccS>::value>(ai, (gbase + ((i + q) & (NT - 1))) * 4);
				acc_mac64(a1[q], vi, aq);
				if (q < H)
					acc_mac64(a2[q], ai, aq);
			}
		}
		if (++cnt == m.chunk) {
			cnt = 0;
#pragma unroll
			for (int q = 0; q < N1; q++)
				acc_set(a1[q], acc_reduce<MERS>(a1[q], m));
#pragma unroll
			for (int q = 0; q < H; q++)
				acc_set(a2[q], acc_reduce<MERS>(a2[q], m));
		}
	}

	/* wave: sum the 64/NT groups; block: sum the waves through LDS; one partial row per block.
	 * Must be reached by every thread of the block. */
	__device__ __forceinline__ void finish(u64 (*red)[SLOTS][NT], u64 *__restrict__ partial, const ModP &m, int slot)
	{
		const int t = threadIdx.x, lane = t & 63, i = t & (NT - 1);
#pragma unroll
		for (int q = 0; q < SLOTS; q++) {
			u64 x = q < N1 ? acc_reduce<MERS>(a1[q < N1 ? q : 0], m) : acc_reduce<MERS>(a2[q < N1 ? 0 : q - N1], m);
#pragma unroll
			for (int off = NT; off < 64; off <<= 1)
				x = addmod(x, shfl_xor64(x, off), m.p);
			if (lane < NT)
				red[t >> 6][q][i] = x;
		}
		__syncthreads();
		for (int e = t; e < SLOTS * NT; e += BS) {
			const int q = e / NT, ii = e % NT;
			u64 x = 0;
#pragma unroll
			for (int w = 0; w < WAVES; w++)
				x = addmod(x, red[w][q][ii], m.p);
			u64 *out = partial + (size_t)slot * 2 * NT * NT;
			if (q < N1) {
				const int jj = (ii + q) & (NT - 1);
				out[ii * NT + jj] = x;
				if (SYM1 && q > 0 && 2 * q < NT)	/* (jj, ii) is computed by no lane: same total */
					out[jj * NT + ii] = x;
			} else {
				const int jj = (ii + (q - N1)) & (NT - 1);
				out[NT * NT + ii * NT + jj] = x;
				out[NT * NT + jj * NT + ii] = x;
			}
		}
	}
};

template <typename W, int MERS, int NT>
__global__ void __launch_bounds__(BLOCK)
k_block_dot_fast(const W *__restrict__ V, const W *__restrict__ AV, long long rows, ModP m,
		 u64 *__restrict__ partial, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	/* the 8-register lazy accumulator while the register file allows (13 of them at NT = 8) */
	using DotAcc = typename std::conditional<sizeof(W) == 4, AccS, typename std::conditional<(NT <= 8), AccL, Acc>::type>::type;
	using DS = DotState<DotAcc, MERS, NT>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	constexpr int GPB = BLOCK / NT;
	const int t = threadIdx.x, lane = t & 63, i = t & (NT - 1), gbase = lane - i;
	const long long g0 = (long long)blockIdx.x * GPB + t / NT, ng = (long long)gridDim.x * GPB;
	DS ds;
	ds.init();
	W nv = 0, na = 0;		/* next row's operands are in flight during this row's MACs */
	if (g0 < rows) {
		nv = V[(size_t)g0 * NT + i];
		na = AV[(size_t)g0 * NT + i];
	}
	for (long long r = g0; r < rows; r += ng) {
		const u64 vv = nv, aa = na;
		if (r + ng < rows) {
			nv = V[(size_t)(r + ng) * NT + i];
			na = AV[(size_t)(r + ng) * NT + i];
		}
		ds.row(vv, aa, i, gbase, m);
	}
	ds.finish(red, partial, m, (int)blockIdx.x);
}

/*
 * n = 64: a block row is one wavefront wide, lane i holds v[r,i] and Av[r,i].  The 64 + 33 rotations do not fit the
 * register file at once, so the rows are walked four times, 16 rotations (and, below 33, 16 symmetric-product terms)
 * per pass; V and AV come out of L2 / the Infinity Cache on the later passes.  Every wavefront writes its own partial
 * row, so there is no cross-wave reduction and no LDS.
 */
template <typename W, int MERS>
__global__ void __launch_bounds__(BLOCK)
k_block_dot_64(const W *__restrict__ V, const W *__restrict__ AV, long long rows, ModP m, u64 *__restrict__ partial,
	       const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = 64, QN = 16, H = NT / 2 + 1;
	constexpr bool NARROW = sizeof(W) == 4;
	using A = typename std::conditional<NARROW, AccS, Acc>::type;
	const int i = threadIdx.x & 63;
	const long long w0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((long long)gridDim.x * BLOCK) >> 6;
	u64 *out = partial + (size_t)w0 * 2 * NT * NT;
	for (int qlo = 0; qlo < NT; qlo += QN) {
		A a1[QN], a2[QN];
#pragma unroll
		for (int q = 0; q < QN; q++) {
			acc_zero(a1[q]);
			acc_zero(a2[q]);
		}
		u32 cnt = 0;
		for (long long r = w0; r < rows; r += nw) {
			const u64 vi = V[(size_t)r * NT + i], ai = AV[(size_t)r * NT + i];
#pragma unroll
			for (int q = 0; q < QN; q++) {
				const int src = ((i + qlo + q) & (NT - 1)) * 4;
				const u64 aq = bperm_word<NARROW>(ai, src);
				acc_mac64(a1[q], vi, aq);
				if (qlo + q < H)		/* uniform: whole passes or the single term q = 32 */
					acc_mac64(a2[q], ai, aq);
			}
			if (++cnt == m.chunk) {
				cnt = 0;
#pragma unroll
				for (int q = 0; q < QN; q++) {
					acc_set(a1[q], acc_reduce<MERS>(a1[q], m));
					acc_set(a2[q], acc_reduce<MERS>(a2[q], m));
				}
			}
		}
#pragma unroll
		for (int q = 0; q < QN; q++) {
			const int jj = (i + qlo + q) & (NT - 1);
			out[i * NT + jj] = acc_reduce<MERS>(a1[q], m);
			if (qlo + q < H) {
				const u64 x = acc_reduce<MERS>(a2[q], m);
				out[NT * NT + i * NT + jj] = x;
				out[NT * NT + jj * NT + i] = x;
			}
		}
	}
}

/*
 * Second SpMV of an iteration (Av = M tmp, sequential/lanczos_modp.c:636) with block_dot_products (:640) as its
 * epilogue: the lane that has just produced Av[r,i] loads v[r,i] and feeds both products.  The SpMV is bound by
 * the gather request rate and leaves the VALU idle ~90 % of the time, so the n x n work is free here and the
 * separate pass over v and Av (2*N*n*w bytes) disappears.  n = NT = G in {1,2,4,8}: at n = 16 the 25 accumulators
 * per lane cost more occupancy than the saved pass is worth (measured: 15.8 ms fused against 12.4 + 2.3 ms apart on
 * the config-5 shape), so that width runs k_spmv and k_block_dot_fast.
 */
template <typename W, int MERS, int NT, bool TAILB>
__global__ void __launch_bounds__(BLOCK)
k_spmv_dot(const u32 *__restrict__ rp, const int *__restrict__ ci, const u32 *__restrict__ va,
	   const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd,
	   long long rows, int accum, u32 heavy, ModP m, u64 *__restrict__ partial, XcdRows xr,
	   const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, BLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	__shared__ u32 spal_store[BLOCK];
	const u32 *spal = pal ? spal_store : nullptr;
	if (pal)
		spal_store[threadIdx.x] = pal[threadIdx.x];
	__syncthreads();
	const int t = threadIdx.x, lane = t & (NT - 1), gbase = (t & 63) - lane;
	long long r = ((long long)blockIdx.x * BLOCK + t) / NT;
	long long ng = (long long)gridDim.x * (BLOCK / NT);
	if (xr.begin[0] >= 0) {		/* per-XCD row ranges, as in k_spmv */
		const int xcd = blockIdx.x & 7;
		r = xr.begin[xcd] + ((long long)(blockIdx.x >> 3) * BLOCK + t) / NT;
		rows = xr.begin[xcd + 1];
		ng = (long long)(gridDim.x >> 3) * (BLOCK / NT);
	}
	DS ds;
	ds.init();
	for (; r < rows; r += ng) {
		const u32 k = rp[r], e = rp[r + 1];
		if (e - k > heavy)	/* row and its share of the inner products: k_spmv_heavy */
			continue;
		const u64 vi = Vd[(size_t)r * NT + lane];
		Acc acc;
		acc_zero(acc);
		spmv_accumulate<W, TAILB>(acc, k, e, ci, va, spal, X, NT, lane);
		if (accum)
			acc_add(acc, Y[(size_t)r * NT + lane]);
		const u64 y = acc_reduce<MERS>(acc, m);
		Y[(size_t)r * NT + lane] = (W)y;
		ds.row(vi, y, lane, gbase, m);
	}
	ds.finish(red, partial, m, (int)blockIdx.x);
}

/* The outlier rows of a slab (see "Outlier rows" above), one workgroup per segment of at most HEAVY_SEG entries.
 * A row that is one segment is finished here; the 128-bit sums of a split row go to `scratch` and
 * k_spmv_heavy_combine adds them.  DOT: the launches follow k_spmv_dot and add these rows' share of v^T Av and
 * Av^T Av as partial rows slot0 + blockIdx.x (n = G). */
template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_heavy(const int *__restrict__ ci, const u32 *__restrict__ va, const u32 *__restrict__ pal,
	     const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd, const HeavySeg *__restrict__ segs,
	     int nseg, u64 *__restrict__ scratch, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
	     const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = DOT ? G : 1;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, BLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	__shared__ Acc slices[BLOCK / G][G];
	__shared__ u32 spal_store[BLOCK];
	const u32 *spal = pal ? spal_store : nullptr;
	if (pal)
		spal_store[threadIdx.x] = pal[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & (G - 1);
	const int xl = lane < n ? lane : 0;
	DS ds;
	if (DOT)
		ds.init();
	for (int h = blockIdx.x; h < nseg; h += gridDim.x) {
		const HeavySeg sg = segs[h];
		Acc acc = heavy_range_sum<W, G>(sg.k0, sg.k1, ci, va, spal, X, n, xl, slices);
		if (threadIdx.x < G && lane < n) {
			if (!sg.whole_row) {
				scratch[((size_t)h * G + lane) * 2] = acc.lo;
				scratch[((size_t)h * G + lane) * 2 + 1] = acc.hi;
				continue;
			}
			const long long r = sg.row;
			if (accum)
				acc_add(acc, Y[(size_t)r * n + lane]);
			const u64 y = acc_reduce<MERS>(acc, m);
			Y[(size_t)r * n + lane] = (W)y;
			if (DOT)
				ds.row(Vd[(size_t)r * n + lane], y, lane, 0, m);
		}
	}
	if (DOT)
		ds.finish(red, partial, m, slot0 + (int)blockIdx.x);
}

/* one lane group per split row: add its segments' sums, reduce, store (and feed the inner products) */
template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_heavy_combine(const HeavyRow *__restrict__ mrows, int nm, const u64 *__restrict__ scratch, W *__restrict__ Y,
		     const W *__restrict__ Vd, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
		     const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = DOT ? G : 1;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, BLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	const int lane = threadIdx.x & (G - 1), gbase = (threadIdx.x & 63) - lane;
	const int g0 = (blockIdx.x * BLOCK + threadIdx.x) / G, ng = gridDim.x * (BLOCK / G);
	DS ds;
	if (DOT)
		ds.init();
	for (int q = g0; q < nm; q += ng) {
		const HeavyRow hr = mrows[q];
		Acc acc;
		acc_zero(acc);
		if (lane < n) {
			for (int sidx = hr.first; sidx < hr.first + hr.count; sidx++)
				acc_add_acc(acc, scratch[((size_t)sidx * G + lane) * 2], scratch[((size_t)sidx * G + lane) * 2 + 1]);
			const long long r = hr.row;
			if (accum)
				acc_add(acc, Y[(size_t)r * n + lane]);
			const u64 y = acc_reduce<MERS>(acc, m);
			Y[(size_t)r * n + lane] = (W)y;
			if (DOT)
				ds.row(Vd[(size_t)r * n + lane], y, lane, gbase, m);
		}
	}
	if (DOT)
		ds.finish(red, partial, m, slot0 + (int)blockIdx.x);
}

/* Medium rows (longer than the outlier threshold, at most 256 entries per lane group of a wavefront): one wavefront
 * per row -- its 64/G lane groups each sum a slice, the slices are added across lanes, group 0 finishes the row.  A
 * workgroup per row (k_spmv_heavy) spends most of its time in barriers on rows of a few hundred entries. */
template <typename W, int G, int MERS, bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_wave(const u32 *__restrict__ rp, const int *__restrict__ ci, const u32 *__restrict__ va,
	    const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd,
	    const int *__restrict__ list, int nlist, int n, int accum, ModP m, u64 *__restrict__ partial, int slot0,
	    const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = DOT ? G : 1, GPW = 64 / G;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, BLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	__shared__ u32 spal_store[BLOCK];
	const u32 *spal = pal ? spal_store : nullptr;
	if (pal)
		spal_store[threadIdx.x] = pal[threadIdx.x];
	__syncthreads();
	const int lane = threadIdx.x & (G - 1), wl = threadIdx.x & 63, grp = wl / G;
	const int xl = lane < n ? lane : 0;
	const int wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLOCK) >> 6;
	DS ds;
	if (DOT)
		ds.init();
	for (int h = wave; h < nlist; h += nwaves) {
		const long long r = list[h];
		const u32 k0 = rp[r], e0 = rp[r + 1];
		const u32 per = (e0 - k0 + GPW - 1) / GPW, lo = k0 + (u32)grp * per;
		Acc acc;
		acc_zero(acc);
		spmv_accumulate<W>(acc, lo < e0 ? lo : e0, (lo + per) < e0 ? (lo + per) : e0, ci, va, spal, X, n, xl);
#pragma unroll
		for (int off = G; off < 64; off <<= 1)
			acc_add_acc(acc, shfl_xor64(acc.lo, off), shfl_xor64(acc.hi, off));
		if (grp == 0 && lane < n) {
			if (accum)
				acc_add(acc, Y[(size_t)r * n + lane]);
			const u64 y = acc_reduce<MERS>(acc, m);
			Y[(size_t)r * n + lane] = (W)y;
			if (DOT)
				ds.row(Vd[(size_t)r * n + lane], y, lane, 0, m);
		}
	}
	if (DOT)
		ds.finish(red, partial, m, slot0 + (int)blockIdx.x);
}

template <typename W, int MERS>
static hipError_t spmv_dot_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum,
				    u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	heavy_fork(c, A, s);
	if (A.panel_rows > 0 && c.panel)
		return panel_dispatch<W, MERS, true>(c, A, X, Y, Vd, accum, partial, max_blocks, nblocks, ctl, s);
	if (A.st_ok && c.staged && !A.st_dyn)	/* (a slab planned for dynamic rows has tiles the lockstep form cannot walk) */
		return staged_dispatch<W, MERS, true>(c, A, X, Y, Vd, accum, partial, max_blocks, nblocks, ctl, s);
	const long long gpb = BLOCK / c.n;
	long long blocks = (A.rows + gpb - 1) / gpb;
	/* the accumulators cost registers: 4 resident blocks per CU at n = 8, so size the grid for that */
	const long long per_cu = c.n >= 8 ? 4 : 6;
	/* partial rows: one per block of the streaming kernel, then one per block of the outlier-row launch */
	const long long hb = heavy_blocks(c, A, max_blocks / 4), cb = A.n_multi ? combine_blocks(A, c.n) : 0;
	const long long mb = A.n_medium ? medium_blocks(c, A) : 0;
	const long long room = max_blocks - hb - cb - mb;
	const long long cap = (long long)c.num_cu * per_cu < room ? (long long)c.num_cu * per_cu : room;
	blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
	XcdRows xr;
	xr.begin[0] = -1;
	if (A.xcd_ranges && blocks >= 64 && (blocks & ~7ll) >= 64) {
		for (int x = 0; x < 9; x++)
			xr.begin[x] = A.xr_rows[x];
		blocks &= ~7ll;		/* whole rounds of the XCDs, still within the room for partial rows */
	}
	*nblocks = (int)(blocks + hb + cb + mb);
#define SPMV_DOT(NN)                                                                                                \
	case NN:                                                                                                    \
		if (A.tail_batch)                                                                                   \
			hipLaunchKernelGGL((k_spmv_dot<W, MERS, NN, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, A.row_ptr, A.col_idx, \
					   A.val, A.palette, X, Y, Vd, (long long)A.rows, accum, A.heavy_thr, c.m, partial, xr, ctl); \
		else                                                                                                \
			hipLaunchKernelGGL((k_spmv_dot<W, MERS, NN, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, A.row_ptr, A.col_idx, \
					   A.val, A.palette, X, Y, Vd, (long long)A.rows, accum, A.heavy_thr, c.m, partial, xr, ctl); \
		if (hb || mb)                                                                                       \
			launch_heavy<W, NN, MERS, true>(c, A, X, Y, Vd, accum, partial, (int)blocks, hb, ctl, s);      \
		break;
	switch (c.n) {
		SPMV_DOT(1)
		SPMV_DOT(2)
		SPMV_DOT(4)
		SPMV_DOT(8)
	default:
		return hipErrorInvalidValue;
	}
#undef SPMV_DOT
	return hipGetLastError();
}

bool spmv_dot_supported(const KernelCfg &c)
{
	return c.n == 1 || c.n == 2 || c.n == 4 || c.n == 8;
}

hipError_t launch_spmv_dot(const KernelCfg &c, const DevCsr &A, const void *X, void *Y, const void *Vd, int accum,
			   u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	if (c.word == 4)
		return c.mers == 31 ? spmv_dot_dispatch<u32, 31>(c, A, (const u32 *)X, (u32 *)Y, (const u32 *)Vd, accum, partial, max_blocks, nblocks, ctl, s)
				    : spmv_dot_dispatch<u32, 0>(c, A, (const u32 *)X, (u32 *)Y, (const u32 *)Vd, accum, partial, max_blocks, nblocks, ctl, s);
	return c.mers == 61 ? spmv_dot_dispatch<u64, 61>(c, A, (const u64 *)X, (u64 *)Y, (const u64 *)Vd, accum, partial, max_blocks, nblocks, ctl, s)
			    : spmv_dot_dispatch<u64, 0>(c, A, (const u64 *)X, (u64 *)Y, (const u64 *)Vd, accum, partial, max_blocks, nblocks, ctl, s);
}

/* ------------------------------------------------------- SpMV, matrix stream staged through LDS (round 2) */

/*
 * Same product as k_spmv / k_spmv_dot (sequential/lanczos_modp.c:266-287), three things changed:
 *
 * 1. The matrix stream goes through LDS.  k_spmv reads row_ptr / col_idx with 4-byte loads that are uniform inside
 *    a lane group: every 128-byte line of the stream is requested from L2 about six times (the gathers evict it from
 *    the 32 KB L1 between uses; profiles/r01_v6_gl7d19_pmc_l2_fabric.txt: 7.5 M stream requests per launch where
 *    1.2 M lines would do).  Here a wavefront owns a TILE of TR = (64/G) * rpg consecutive rows; the tile's entries
 *    are one contiguous piece of col_idx (and val), copied into the wavefront's LDS buffer with 16-byte-per-lane
 *    LDS-DMA loads (global_load_lds_dwordx4: 1 KB per instruction, no VGPRs), one tile ahead of the gathers, so each
 *    line of the stream crosses L2 -> CU once.  The row pointers of a tile are one coalesced load (lane i holds
 *    rp[r0 + i]) fetched two tiles ahead; lanes read them from each other.
 * 2. Tiles are dealt to XCDs in contiguous, nnz-balanced ranges (XcdTiles): blocks b and b + 8 share an XCD
 *    (observed round-robin placement, MI355X_MICROARCH.md -- speed only, any placement computes every tile once), so
 *    an XCD's 4 MB L2 sees one eighth of the rows instead of an interleaved sample of all of them; on matrices whose
 *    rows have local column supports each L2 then caches a different part of X.
 * 3. Row tails are predicated batches, not one dependent gather per left-over entry.
 *
 * A row whose entries do not all fit the staged window (tiles of unusually long rows) takes the k_spmv path from
 * global memory; rows above the outlier threshold are left to k_spmv_wave / k_spmv_heavy as before.
 */
enum { V_ONES = 0, V_PACKED = 1, V_ARRAY = 2 };

struct XcdTiles {
	long long begin[9];
	int interleave;		/* 1: tiles dealt round-robin over all wavefronts of the grid (no XCD ranges) */
};

template <typename W, int VALS, int U>
MODP_DEV void staged_accumulate(Acc &acc, u32 i, u32 i1, const u32 *sci, const u32 *sva, const u32 *spal,
				const W *__restrict__ X, int stride, int xl)
{
	/* every slot of a batch reads LDS and gathers unconditionally (a left-over slot re-reads the row's last entry:
	 * same LDS word, same line of X) and is switched off by a zero multiplier: no branches inside the batch */
	const u32 last = i1 - 1;
	for (; i < i1; i += U) {
		u32 c[U];
		u32 a[U];
		W x[U];
#pragma unroll
		for (int j = 0; j < U; j++) {
			const u32 at = i + j < i1 ? i + j : last;
			c[j] = sci[at];
			if (VALS == V_ARRAY)
				a[j] = sva[at];
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			x[j] = X[(size_t)(VALS == V_PACKED ? (c[j] & 0xFFFFFFu) : c[j]) * stride + xl];
			if (VALS == V_PACKED)
				a[j] = spal[c[j] >> 24];
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			const bool live = i + j < i1;
			if (VALS == V_ONES)
				acc_add(acc, live ? (u64)x[j] : 0ull);
			else
				acc_mac32(acc, live ? a[j] : 0u, x[j]);
		}
	}
}

/*
 * DYN (round 3): the lane groups of a wavefront take the rows of a tile from a shared counter instead of in lockstep.
 * With one row per group per step, every step lasts as long as its longest row, and a group that has finished issues
 * no gathers meanwhile: on CSR(M^T) of the config-5 shape (Poisson row lengths around 40, four groups of 16 lanes) a
 * step of four rows takes 6.3 batches of eight gathers on average where 5 would do, and the product ran 11 % slower
 * than the one over the constant-length rows of CSR(M).  Here every group keeps its own position; at the top of a round
 * the groups whose row is exhausted (one ballot) store it and take the next unassigned rows of the tile, the others go
 * on gathering: a tile of R rows takes about (sum of the rows' batches) / groups rounds instead of the sum of the
 * steps' maxima.  Rows stay whole and owned by one group -- same sums, same words, no cross-group reduction.
 * TR is then any row count up to 64 (the row pointers of a tile are one register across the wavefront), chosen so that
 * the tile's entries fill the staging window.
 */
/* The same batch with TWO adjacent words per lane (16-byte gathers: a 128-byte block row of n = 16 u64 words is 8 lanes,
 * a wavefront holds 8 rows instead of 4 and issues half the vector-memory instructions per row).  tools/ubench5: the
 * gather + output-row loop runs 5-9 % faster with 16 bytes per lane than with 8 (profiles/r03_ubench5_*.txt). */
template <typename W, int VALS, int U>
MODP_DEV void staged_accumulate2(Acc &a0, Acc &a1, u32 i, u32 i1, const u32 *sci, const u32 *sva, const u32 *spal,
				 const W *__restrict__ X, int stride, int xw)
{
	typedef W W2 __attribute__((ext_vector_type(2)));
	const u32 last = i1 - 1;
	for (; i < i1; i += U) {
		u32 c[U];
		u32 a[U];
		W2 x[U];
#pragma unroll
		for (int j = 0; j < U; j++) {
			const u32 at = i + j < i1 ? i + j : last;
			c[j] = sci[at];
			if (VALS == V_ARRAY)
				a[j] = sva[at];
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			x[j] = *(const W2 *)(X + (size_t)(VALS == V_PACKED ? (c[j] & 0xFFFFFFu) : c[j]) * stride + xw);
			if (VALS == V_PACKED)
				a[j] = spal[c[j] >> 24];
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			const bool live = i + j < i1;
			if (VALS == V_ONES) {
				acc_add(a0, live ? (u64)x[j].x : 0ull);
				acc_add(a1, live ? (u64)x[j].y : 0ull);
			} else {
				acc_mac32(a0, live ? a[j] : 0u, x[j].x);
				acc_mac32(a1, live ? a[j] : 0u, x[j].y);
			}
		}
	}
}

/* WPL = words per lane: 1 (a group of G lanes owns a block row of up to G words) or 2 (the row is 2 G words, 16-byte
 * gathers and stores; n = 2 G exactly, no fused inner products, lockstep rows) */
template <typename W, int G, int MERS, bool DOT, int VALS, int U, bool DYN = false, int WPL = 1>
__global__ void __launch_bounds__(BLOCK, (DOT && G >= 8) ? 4 : 1)	/* the fused form at n = 8 must keep 4 workgroups per CU (<= 128 VGPRs) */
k_spmv_staged(const u32 *__restrict__ rp, const u32 *__restrict__ ci, const u32 *__restrict__ va,
	      const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd,
	      long long rows, int n, int tile_rows, int capw, int accum, u32 heavy, ModP m, u64 *__restrict__ partial,
	      XcdTiles xt, const DevCtl *__restrict__ ctl)
{
	static_assert(!(DYN && DOT), "the dynamic form has no fused inner products (yet)");
	static_assert(WPL == 1 || (WPL == 2 && !DOT && !DYN && sizeof(W) == 8), "two words per lane: plain lockstep form, 64-bit words");
	if (ctl->stop)
		return;
	constexpr int GPW = 64 / G, NS = VALS == V_ARRAY ? 2 : 1, WAVES = BLOCK / 64, NT = DOT ? G : 1;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, BLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	__shared__ u32 spal_store[BLOCK];
	extern __shared__ __attribute__((aligned(16))) u32 stage[];	/* [wave][2 buffers][NS streams][capw] */
	if (VALS == V_PACKED)
		spal_store[threadIdx.x] = pal[threadIdx.x];
	__syncthreads();
	const int t = threadIdx.x, wl = t & 63, wave = t >> 6, lane = wl & (G - 1), grp = wl / G;
	const int xl = lane < n ? lane : 0, gbase = wl - lane;
	u32 *const mine = stage + (size_t)wave * 2 * NS * capw;
	const int TR = tile_rows, rpg = TR / GPW;	/* lockstep form: TR = GPW * rows per group */
	const int xcd = blockIdx.x & 7;
	const long long nwv = xt.interleave ? (long long)gridDim.x * WAVES : (long long)(gridDim.x >> 3) * WAVES;
	long long tile = xt.interleave ? (long long)blockIdx.x * WAVES + wave
				       : xt.begin[xcd] + (long long)(blockIdx.x >> 3) * WAVES + wave;
	const long long tile_end = xt.interleave ? xt.begin[8] : xt.begin[xcd + 1];
	DS ds;
	if (DOT)
		ds.init();

	/* row pointers of a tile: lane i holds rp[r0 + i] (clamped), `re` = rp[r0 + TR] */
	auto load_rp = [&](long long tl, u32 &rv, u32 &re) {
		const long long r0 = tl * TR;
		long long a = r0 + wl, b = r0 + TR;
		a = a > rows ? rows : a;
		b = b > rows ? rows : b;
		rv = rp[a];
		re = rp[b];
	};
	/* copy the tile's piece of the stream(s) into buffer `buf`: entries [K0, K0 + capw) with K0 = rp[r0] & ~3 */
	auto stage_tile = [&](u32 rv, u32 re, int buf) {
		const u32 klo = (u32)__shfl((int)rv, 0, 64), K0 = klo & ~3u;
		u32 cnt = re - K0;
		cnt = cnt > (u32)capw ? (u32)capw : cnt;
		const u32 nch = (cnt + 3u) >> 2;
		u32 *dst = mine + (size_t)buf * NS * capw;
		for (u32 c = 0; c < nch; c += 64) {
			if (c + (u32)wl < nch) {
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ci + (size_t)K0 + 4 * (size_t)(c + wl)),
								 (__attribute__((address_space(3))) void *)(dst + 4 * c), 16, 0, 0);
				if (VALS == V_ARRAY)
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(va + (size_t)K0 + 4 * (size_t)(c + wl)),
									 (__attribute__((address_space(3))) void *)(dst + capw + 4 * c), 16, 0, 0);
			}
		}
	};

	u32 rv0 = 0, re0 = 0, rv1 = 0, re1 = 0;
	if (tile < tile_end) {
		load_rp(tile, rv0, re0);
		stage_tile(rv0, re0, 0);
		if (tile + nwv < tile_end)
			load_rp(tile + nwv, rv1, re1);
	}
	/* The store of a finished row is issued one row late (after the next row's sums are complete), so that at the top
	 * of a tile the youngest vector-memory operations of the wavefront are gathers it has already consumed: memory
	 * operations retire in issue order, and the wait for the tile's stream then costs nothing -- with the store issued
	 * last it exposed the full store latency once per tile (measured: +3..5 % on the GL7d19 shape). */
	W *pend_at = nullptr;
	u64 pend_y = 0, pend_y1 = 0;
	int buf = 0;
	for (; tile < tile_end; tile += nwv, buf ^= 1) {
		/* everything this wavefront has issued so far has landed: this tile's stream and the next tile's row pointers */
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		u32 rv2 = 0, re2 = 0;
		if (tile + nwv < tile_end) {
			stage_tile(rv1, re1, buf ^ 1);
			if (tile + 2 * nwv < tile_end)
				load_rp(tile + 2 * nwv, rv2, re2);
		}
		asm volatile("" ::: "memory");
		const long long r0 = tile * TR;
		const u32 K0 = (u32)__shfl((int)rv0, 0, 64) & ~3u;
		const u32 *sci = mine + (size_t)buf * NS * capw, *sva = sci + capw;
		if constexpr (DYN) {
			constexpr unsigned long long LEAD = ~0ull / ((G < 64 ? (1ull << (G & 63)) : 0ull) - 1ull);	/* bit 0 of every group */
			const unsigned long long below = (1ull << gbase) - 1ull;
			const int tr = (int)(rows - r0 < (long long)TR ? rows - r0 : (long long)TR);
			u32 gi = 0, ge = 0;		/* this group's position in its row, and the row's end (entries from K0) */
			int grow = -1;			/* its row of the tile, or -1 */
			int next = 0;			/* rows of the tile handed out so far (the same in every lane) */
			Acc acc;
			acc_zero(acc);
			for (;;) {
				const bool want = gi >= ge;
				const unsigned long long wm = __ballot(want) & LEAD;
				if (wm) {
					/* the bounds of the row each asking group would get, fetched with every lane active (a lane that is
					 * masked off supplies 0 to a cross-lane read) */
					const int q = next + __popcll(wm & below), qq = q < tr ? q : 0;
					const u32 k = (u32)__shfl((int)rv0, qq, 64);
					const u32 en = (u32)__shfl((int)rv0, (qq + 1) & 63, 64);
					const u32 e2 = qq + 1 < 64 ? en : re0;
					if (want) {
						if (grow >= 0 && lane < n) {
							const long long r = r0 + grow;
							if (accum)
								acc_add(acc, Y[(size_t)r * n + lane]);
							const u64 y = acc_reduce<MERS>(acc, m);
							if (pend_at)
								*pend_at = (W)pend_y;
							pend_at = Y + (size_t)r * n + lane;
							pend_y = y;
						}
						acc_zero(acc);
						if (q < tr && e2 - k <= heavy) {	/* (rows above the outlier threshold belong to k_spmv_wave / _heavy) */
							grow = q;
							gi = k - K0;
							ge = e2 - K0;
						} else {
							grow = -1;
							gi = ge = 0;
						}
					}
					next += __popcll(wm);
				}
				if (next >= tr && __ballot(grow >= 0) == 0)
					break;
				if (grow >= 0 && gi < ge) {
					const u32 stop_at = gi + U < ge ? gi + U : ge;
					if (ge <= (u32)capw)
						staged_accumulate<W, VALS, U>(acc, gi, stop_at, sci, sva, spal_store, X, n, xl);
					else		/* the row runs past the staged window: its entries come from global memory */
						spmv_accumulate<W>(acc, K0 + gi, K0 + stop_at, (const int *)ci, VALS == V_ARRAY ? va : nullptr,
								   VALS == V_PACKED ? spal_store : nullptr, X, n, xl);
					gi = stop_at;
				}
			}
		} else
		for (int j = 0; j < rpg; j++) {
			const int q = j * GPW + grp;
			const long long r = r0 + q;
			const u32 k = (u32)__shfl((int)rv0, q, 64);
			const u32 enext = (u32)__shfl((int)rv0, (q + 1) & 63, 64);
			const u32 e = q + 1 < 64 ? enext : re0;
			if constexpr (WPL == 2) {
				if (r < rows && e - k <= heavy) {
					typedef W W2 __attribute__((ext_vector_type(2)));
					Acc acc, acc1;
					acc_zero(acc);
					acc_zero(acc1);
					if (e - K0 <= (u32)capw) {
						staged_accumulate2<W, VALS, U>(acc, acc1, k - K0, e - K0, sci, sva, spal_store, X, n, 2 * lane);
					} else {	/* the row runs past the staged window: its entries come from global memory */
						for (u32 q = k; q < e; q++) {
							const u32 cw = ci[q];
							const W2 xv = *(const W2 *)(X + (size_t)(VALS == V_PACKED ? (cw & 0xFFFFFFu) : cw) * n + 2 * lane);
							if (VALS == V_ONES) {
								acc_add(acc, (u64)xv.x);
								acc_add(acc1, (u64)xv.y);
							} else {
								const u32 av = VALS == V_PACKED ? spal_store[cw >> 24] : va[q];
								acc_mac32(acc, av, xv.x);
								acc_mac32(acc1, av, xv.y);
							}
						}
					}
					W *const yat = Y + (size_t)r * n + 2 * lane;
					if (accum) {
						const W2 old = *(const W2 *)yat;
						acc_add(acc, (u64)old.x);
						acc_add(acc1, (u64)old.y);
					}
					const u64 y0 = acc_reduce<MERS>(acc, m), y1 = acc_reduce<MERS>(acc1, m);
					if (pend_at) {
						const W2 pv = { (W)pend_y, (W)pend_y1 };
						*(W2 *)pend_at = pv;
					}
					pend_at = yat;
					pend_y = y0;
					pend_y1 = y1;
				}
			} else
			if (r < rows && e - k <= heavy) {
				Acc acc;
				acc_zero(acc);
				u64 vi = 0;
				if (DOT)
					vi = Vd[(size_t)r * NT + lane];
				if (e - K0 <= (u32)capw)
					staged_accumulate<W, VALS, U>(acc, k - K0, e - K0, sci, sva, spal_store, X, n, xl);
				else
					spmv_accumulate<W>(acc, k, e, (const int *)ci, VALS == V_ARRAY ? va : nullptr,
							   VALS == V_PACKED ? spal_store : nullptr, X, n, xl);
				if (lane < n) {
					if (accum)
						acc_add(acc, Y[(size_t)r * n + lane]);
					const u64 y = acc_reduce<MERS>(acc, m);
					if (pend_at)
						*pend_at = (W)pend_y;
					pend_at = Y + (size_t)r * n + lane;
					pend_y = y;
					if (DOT)
						ds.row(vi, y, lane, gbase, m);
				}
			}
		}
		rv0 = rv1;
		re0 = re1;
		rv1 = rv2;
		re1 = re2;
	}
	if (pend_at) {
		if constexpr (WPL == 2) {
			typedef W W2 __attribute__((ext_vector_type(2)));
			const W2 pv = { (W)pend_y, (W)pend_y1 };
			*(W2 *)pend_at = pv;
		} else {
			*pend_at = (W)pend_y;
		}
	}
	if (DOT)
		ds.finish(red, partial, m, (int)blockIdx.x);
}

/* two words per lane where the block row is 16 or 8 u64 words (see staged_accumulate2); BLZ_NO_PAIR=1 keeps one word per lane.
 * Measured (profiles/r03_synth5q_pair_lanes.txt, r03_pair_lanes_n8.txt): config-5 quarter shape, n = 16: 12.46 / 11.67 ->
 * 11.42 / 11.21 ms per product; relat9 shape, n = 8, first product (the second carries the inner products and keeps one
 * word per lane): 624 -> 617 us. */
static inline bool spmv_pair_lanes(const KernelCfg &c)
{
	return c.pair && c.word == 8 && (c.n == 16 || c.n == 8);
}

/* Host side of the plan, made once per slab at upload (blz_api.hip): tile height, staging window, grid density and the
 * nnz-balanced tile ranges of the eight XCDs. */
void spmv_plan_staged(const KernelCfg &c, const u32 *row_ptr, DevCsr &D, bool allow_dyn)
{
	D.st_ok = false;
	D.st_dyn = false;
	if (D.rows <= 0 || D.nnz <= 0)
		return;
	int G = 1;
	while (G < c.n)
		G <<= 1;
	if (spmv_split_log2(c, D.rows, D.nnz) != 0)	/* few long rows shared by adjacent groups: k_spmv */
		return;
	const double avg = D.kept_mean >= 0.0 ? D.kept_mean : (double)D.nnz / (double)D.rows;
	/* gathers that mostly hit (see below) */
	const bool local = allow_dyn && D.locality < 0.3 && !D.uneven && D.outlier_share < 0.02;	/* (nfs: 17 % of the entries in outlier rows) */
	/* lanes per row: G, or G / 2 lanes of two words (16-byte gathers and stores); the heuristics below go by the row width G.
	 * A slab that carries the inner products keeps one word per lane.  At n = 8 the form pays on rows of a few entries (the
	 * output rows are then a large part of the traffic: relat9 shape from the right, rows of 3: 622 -> 616 us) and on
	 * gathers that hit (band matrix: 411 -> 326 us), and LOSES on long rows gathered from a large operand (relat9 shape from
	 * the left, rows of 71 out of 791 MB: 735 -> 780 us): profiles/r03_exp_pair_rows.txt */
	const bool pair = spmv_pair_lanes(c) && allow_dyn && c.stage_dyn != 1 && (G == 16 || avg < 12.0 || local);
	D.st_pair = pair;
	{
		/* Measured on MI355X (gpurun_out r2a..r2g, profiles/r02_staged_*): staging pays where the stream is a large
		 * part of the row's work or rows span several lines of it -- relat9 shape, rows of 3 entries: 711 -> 668 us,
		 * its transpose, rows of 71: 751 -> 727 us -- and costs 2-4 % on rows of ~20 entries (GL7d19 shape: 654 ->
		 * 670 us at every tile height, window, grid density and batch depth tried), where k_spmv's per-row loads
		 * already hit L1.  BLZ_STAGE_ALWAYS=1 stages every slab (PMC comparisons). */
		const double a = D.kept_mean >= 0.0 ? D.kept_mean : (double)D.nnz / (double)D.rows;
		const char *e = getenv("BLZ_STAGE_ALWAYS");
		/* (the same holds at 128-byte block rows: GL7d19 shape at n = 16, rows of 19.5: 722 -> 779 us staged, while the
		 * config-5 quarter shape, rows of 40, gains: 13.1 -> 12.4 ms -- there 130 M stream requests per launch become
		 * 16 M lines, profiles/r02_synth5q_pmc.txt.  The boundary is put at 32 entries.) */
		/* Round 3: ... unless the gathers mostly HIT.  With the renumbering's sampled footprint below 0.3 lines per entry
		 * (a banded or otherwise local matrix) the product is paced by the kernel, not by the fabric, and there the staged
		 * form (with 16-byte lanes) is the quicker one whatever the row length -- band matrix, 2 M x 2 M, rows of 20, first
		 * product: 535 us as k_spmv, 332 us staged (profiles/r03_exp_band_staged_always.txt; round 2 saw 516 -> 369 on the
		 * GL7d19 shape with all entries in a band) -- except on heavy-tailed row lengths, where it loses (`nfs` workload: 853
		 * against 664 us).  The product that carries the inner products gains nothing from it (406 against 404 us). */
		if (!(e && e[0] == '1') && a >= 12.0 && a < 32.0 && !local)
			return;
	}
	const int GPW = pair ? 128 / G : 64 / G, NS = (D.val && !D.palette) ? 2 : 1;
	/* Rows of a few entries (relat9 shape, 3.15 per row, profiles/r02_exp_staged_sweep_relat9.txt): a batch of 8 gathers
	 * per lane is mostly switched-off slots, and 16 wavefronts per CU with the larger window beat 32 with the smaller one
	 * (672 -> 622 us in the iteration; 4 per CU x U = 4: 623 us, 8 x 4: 691, 4 x 8: 772, 8 x 8: 739 in the sweep). */
	const bool few = avg < 6.0 && G < 16;
	D.st_deep = !few;
	int per_cu = c.spmv_blocks_per_cu > 0 ? c.spmv_blocks_per_cu : (((avg >= 12.0 || few) && G < 16) ? (D.uneven && !few ? 6 : 4) : 8);
	/* two words per lane: 8 rows per wavefront keep the fabric busy with 16 wavefronts per CU and the larger window (config-5
	 * quarter shape, per product: 11.85 / 11.17 ms at 8 workgroups per CU, 11.42 / 11.21 at 4, 14.08 / 11.99 at 6) */
	if (pair && G == 16 && c.spmv_blocks_per_cu <= 0)
		per_cu = 4;
	/* LDS: 4 wavefronts x 2 buffers x capw entries x 4 B per block (x NS streams); 160 KB per CU */
	int capw = (per_cu <= 4 ? 1024 : 512) / NS;
	if (const char *e = getenv("BLZ_STAGE_CAPW"))
		if (atoi(e) >= 64 && atoi(e) <= 4096)
			capw = atoi(e) & ~3;
	while ((long long)per_cu * (BLOCK / 64) * 2 * NS * capw * 4 > 150 * 1024 && per_cu > 1)
		per_cu--;
	int rpg = (int)(0.6 * capw / (GPW * (avg > 1.0 ? avg : 1.0)));
	if (const char *e = getenv("BLZ_STAGE_RPG"))
		if (atoi(e) >= 1)
			rpg = atoi(e);
	if (rpg < 1) {
		if (GPW * avg * 1.25 > capw)	/* a tile of one row per group does not fit the window: nothing to stage */
			return;
		rpg = 1;
	}
	if (rpg > 64 / GPW)
		rpg = 64 / GPW;
	long long TR = (long long)GPW * rpg;
	/* Dynamic rows (k_spmv_staged<..., DYN>): where a wavefront holds few lane groups and the rows take several batches
	 * each, so that a step of the lockstep form lasts as long as its longest row.  The tile is then as many rows as fill
	 * ~85 % of the window on average (a tile that overflows it reads the overflowing rows from global memory).
	 * BLZ_STAGE_DYN=0 / 1 forces it off / on wherever the kernel has the form (A/B). */
	/* Measured on the config-5 quarter shape (profiles/r03_synth5q_dynamic_rows.txt): SLOWER than the lockstep form --
	 * 12.67 / 12.43 ms against 12.45 / 11.68 per product -- although every lane group gathers in every round: the gathers
	 * are bound by the fabric, not by idle groups, and the bookkeeping of a round costs issue slots.  Off unless asked for. */
	bool dyn = false;
	if (c.stage_dyn >= 0)
		dyn = allow_dyn && c.stage_dyn == 1 && G >= 8 && G < 64;	/* (the launcher has the form for 8, 16 and 32 lanes per group) */
	if (dyn) {
		long long t = (long long)(0.85 * capw / (avg > 1.0 ? avg : 1.0));
		if (const char *e = getenv("BLZ_STAGE_TR"))
			if (atoi(e) >= 1)
				t = atoi(e);
		TR = t < GPW ? GPW : (t > 64 ? 64 : t);
		D.st_dyn = true;
	}
	const long long nt = (D.rows + TR - 1) / TR;
	D.st_tr = (int)TR;
	D.st_rpg = rpg;
	D.st_capw = capw;
	D.st_per_cu = per_cu;
	D.st_ns = NS;
	D.st_tiles[0] = 0;
	for (int x = 1; x < 8; x++) {
		const double target = (double)D.nnz * x / 8.0;
		long long lo = D.st_tiles[x - 1], hi = nt;
		while (lo < hi) {
			const long long mid = (lo + hi) / 2, r = mid * TR < D.rows ? mid * TR : D.rows;
			if ((double)row_ptr[r] < target)
				lo = mid + 1;
			else
				hi = mid;
		}
		D.st_tiles[x] = lo;
	}
	D.st_tiles[8] = nt;
	{
		const char *e = getenv("BLZ_STAGE_INTERLEAVE");	/* experiments: 1 = no XCD ranges */
		D.st_interleave = e && e[0] == '1';
	}
	D.st_ok = true;
}

template <typename W, int MERS, bool DOT, int G, int VALS>
static void staged_launch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum, u64 *partial,
			  long long blocks, const DevCtl *ctl, hipStream_t s)
{
	XcdTiles xt;
	for (int x = 0; x < 9; x++)
		xt.begin[x] = A.st_tiles[x];
	xt.interleave = A.st_interleave;
	const size_t lds = (size_t)(BLOCK / 64) * 2 * A.st_ns * A.st_capw * sizeof(u32);
	/* gathers in flight per lane: 8 where a wavefront holds few lane groups (G >= 16) or the registers allow (the
	 * plain form at G = 8), else 4; BLZ_STAGE_U overrides between the two where both exist */
	bool deep = G >= 16 || (G == 8 && !DOT && A.st_deep);
	if (c.stage_u > 0)
		deep = c.stage_u >= 8;
#define STAGED_GO(UU, DD)                                                                                               \
	hipLaunchKernelGGL((k_spmv_staged<W, G, MERS, DOT, VALS, UU, DD>), dim3((unsigned)blocks), dim3(BLOCK), lds, s, A.row_ptr, \
			   (const u32 *)A.col_idx, A.val, A.palette, X, Y, Vd, (long long)A.rows, c.n, A.st_tr, A.st_capw, accum, \
			   A.heavy_thr, c.m, partial, xt, ctl)
	if constexpr ((G == 16 || G == 8) && !DOT && sizeof(W) == 8) {
		if (A.st_pair) {	/* G / 2 lanes of two words per row */
#define STAGED_PAIR(UU)                                                                                                 \
	hipLaunchKernelGGL((k_spmv_staged<W, G / 2, MERS, false, VALS, UU, false, 2>), dim3((unsigned)blocks), dim3(BLOCK), lds, s, \
			   A.row_ptr, (const u32 *)A.col_idx, A.val, A.palette, X, Y, Vd, (long long)A.rows, c.n, A.st_tr, A.st_capw, \
			   accum, A.heavy_thr, c.m, partial, xt, ctl)
			if (G == 16 || A.st_deep)
				STAGED_PAIR(8);
			else
				STAGED_PAIR(4);
#undef STAGED_PAIR
			return;
		}
	}
	if constexpr (G >= 8 && !DOT) {
		if (A.st_dyn) {
			if constexpr (G < 64) {
				if (deep)
					STAGED_GO(8, true);
				else
					STAGED_GO(4, true);
			}
		} else if (deep)
			STAGED_GO(8, false);
		else
			STAGED_GO(4, false);
	} else {
		STAGED_GO(4, false);
	}
#undef STAGED_GO
}

template <typename W, int MERS, bool DOT>
static hipError_t staged_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum,
				  u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	int G = 1;
	while (G < c.n)
		G <<= 1;
	const long long TR = A.st_tr, nt = (A.rows + TR - 1) / TR;
	long long blocks = (nt + BLOCK / 64 - 1) / (BLOCK / 64);
	long long cap = (long long)c.num_cu * A.st_per_cu;
	long long hb = heavy_blocks(c, A, 1 << 30), cb = 0, mb = 0;
	if (DOT) {	/* one partial row per block of every launch that feeds the inner products */
		hb = heavy_blocks(c, A, max_blocks / 4);
		cb = A.n_multi ? combine_blocks(A, c.n) : 0;
		mb = A.n_medium ? medium_blocks(c, A) : 0;
		const long long room = ((long long)max_blocks - hb - cb - mb) & ~7ll;
		cap = cap < room ? cap : room;
	}
	blocks = blocks > cap ? cap : blocks;
	blocks = (blocks + 7) & ~7ll;		/* whole rounds of the eight XCDs */
	if (blocks == ((cap + 7) & ~7ll) && blocks >= 64) {
		/* A wavefront walks its XCD's tiles with a fixed stride, so the launch takes ceil(T / waves) rounds of tiles
		 * and the last round is partly empty (GL7d19 shape: 9954 tiles per XCD over 512 wavefronts = 19.4 rounds: the
		 * twentieth runs with 43 % of the chip, measured +3 % on the launch).  Give up to 15 % of the wavefronts away
		 * so that the rounds come out (nearly) full: the fabric, not the number of resident waves, bounds the kernel. */
		long long T = 0;
		for (int x = 0; x < 8; x++)
			T = std::max(T, A.st_tiles[x + 1] - A.st_tiles[x]);
		const long long W4 = BLOCK / 64, bmax = blocks / 8, bmin = std::max<long long>(1, bmax * 85 / 100);
		long long best = bmax;
		double best_fill = 0.0;
		for (long long b = bmax; b >= bmin; b--) {
			const long long rounds = (T + b * W4 - 1) / (b * W4);
			const double fill = (double)T / (double)(rounds * b * W4);
			if (fill > best_fill + 1e-9) {
				best_fill = fill;
				best = b;
			}
		}
		blocks = best * 8;
	}
	if (DOT)
		*nblocks = (int)(blocks + hb + cb + mb);
	const int vals = A.palette ? V_PACKED : (A.val ? V_ARRAY : V_ONES);
#define STAGED_G(GG)                                                                                              \
	case GG:                                                                                                  \
		if (vals == V_PACKED)                                                                             \
			staged_launch<W, MERS, DOT, GG, V_PACKED>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s); \
		else if (vals == V_ARRAY)                                                                         \
			staged_launch<W, MERS, DOT, GG, V_ARRAY>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s);  \
		else                                                                                              \
			staged_launch<W, MERS, DOT, GG, V_ONES>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s);   \
		if (A.n_heavy || A.n_medium)                                                                      \
			launch_heavy<W, GG, MERS, DOT>(c, A, X, Y, Vd, accum, partial, (int)blocks, hb, ctl, s);      \
		break;
	if constexpr (DOT) {
		switch (G) {
			STAGED_G(1)
			STAGED_G(2)
			STAGED_G(4)
			STAGED_G(8)
		default:
			return hipErrorInvalidValue;
		}
	} else {
		switch (G) {
			STAGED_G(1)
			STAGED_G(2)
			STAGED_G(4)
			STAGED_G(8)
			STAGED_G(16)
			STAGED_G(32)
			STAGED_G(64)
		default:
			return hipErrorInvalidValue;
		}
	}
#undef STAGED_G
	return hipGetLastError();
}

/* ------------------------------------------ SpMV with a panel of dense block rows resident in LDS (round 2) */

/*
 * Matrices with heavy-tailed column degrees (sieve relation matrices: the columns of the small primes hold a large
 * share of the entries) gather the same few block rows of X over and over.  The host numbers the densest rows and
 * columns first (blz_reorder_hot) and sorts every row's entries by column, so the first `H` block rows of the operand
 * are the hot ones and a row's hot entries come first.  One workgroup of 1024 threads per CU copies X[0..H) into LDS
 * (up to 128 KB) and keeps it for the whole launch: an entry with column < H reads its block row from LDS, the others
 * are gathered as in k_spmv.  The gather ceiling (~55 G line fills per second, whatever the row size) then applies to
 * the cold entries only (tools/ubench2 q3: half of the entries in the panel = 1.9 x the rows per second).
 * Rows are dealt to XCDs in contiguous nnz-balanced ranges (blocks b and b + 8 share an XCD), so neighbouring rows with
 * overlapping column supports meet in one 4 MB L2 instead of being fetched into all eight.
 * DOT: block_dot_products as the epilogue, as in k_spmv_dot (n = G in {1,2,4,8}).
 */
#define PBLOCK 1024
#define PANEL_BYTES (128 * 1024)

template <typename W, int G, int MERS, bool DOT, int VALS>
__global__ void __launch_bounds__(PBLOCK)
k_spmv_panel(const u32 *__restrict__ rp, const u32 *__restrict__ ci, const u32 *__restrict__ va,
	     const u32 *__restrict__ pal, const W *__restrict__ X, W *__restrict__ Y, const W *__restrict__ Vd, int accum,
	     u32 heavy, ModP m, u64 *__restrict__ partial, XcdRows xr, u32 H, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = DOT ? G : 1;
	using DS = DotState<typename std::conditional<sizeof(W) == 4, AccS, Acc>::type, MERS, NT, PBLOCK, true>;
	__shared__ u64 red[DS::WAVES][DS::SLOTS][NT];
	__shared__ u32 spal[256];
	extern __shared__ __attribute__((aligned(16))) unsigned char panel_raw[];
	W *panel = (W *)panel_raw;
	const int t = threadIdx.x, lane = t & (G - 1), gbase = (t & 63) - lane;
	if (VALS == V_PACKED && t < 256)
		spal[t] = pal[t];
	for (u32 i = (u32)t; i < H * (u32)G; i += PBLOCK)
		panel[i] = X[i];
	__syncthreads();
	const int xcd = blockIdx.x & 7;
	const long long ng = (long long)(gridDim.x >> 3) * (PBLOCK / G);
	const long long g0 = xr.begin[xcd] + (long long)(blockIdx.x >> 3) * (PBLOCK / G) + t / G;
	const long long rend = xr.begin[xcd + 1];
	const u32 Hm1 = H - 1;
	DS ds;
	if (DOT)
		ds.init();
	for (long long r = g0; r < rend; r += ng) {
		u32 k = rp[r];
		const u32 e = rp[r + 1];
		if (e - k > heavy)
			continue;
		u64 vi = 0;
		if (DOT)
			vi = Vd[(size_t)r * G + lane];
		Acc acc;
		acc_zero(acc);
		for (; k < e; k += 4) {
			u32 c[4], a[4];
			W x[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const u32 at = k + j < e ? k + j : e - 1;
				const u32 pk = ci[at];
				c[j] = VALS == V_PACKED ? (pk & 0xFFFFFFu) : pk;
				if (VALS == V_PACKED)
					a[j] = spal[pk >> 24];
				else if (VALS == V_ARRAY)
					a[j] = va[at];
				else
					a[j] = 1u;
				if (k + j >= e)
					a[j] = 0u;
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const W hot = panel[(size_t)(c[j] < H ? c[j] : Hm1) * G + lane];
				W cold = 0;
				if (c[j] >= H)
					cold = X[(size_t)c[j] * G + lane];
				x[j] = c[j] < H ? hot : cold;
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				if (VALS == V_ONES)
					acc_add(acc, a[j] ? (u64)x[j] : 0ull);
				else
					acc_mac32(acc, a[j], x[j]);
			}
		}
		if (accum)
			acc_add(acc, Y[(size_t)r * G + lane]);
		const u64 y = acc_reduce<MERS>(acc, m);
		Y[(size_t)r * G + lane] = (W)y;
		if (DOT)
			ds.row(vi, y, lane, gbase, m);
	}
	if (DOT)
		ds.finish(red, partial, m, (int)blockIdx.x);
}

/* rows of the panel an operand of width G words of `word` bytes can have */
int64_t spmv_panel_capacity(const KernelCfg &c)
{
	int G = 1;
	while (G < c.n)
		G <<= 1;
	if (G != c.n)		/* exact-width blocks (BLZ_NO_PAD): no panel */
		return 0;
	return PANEL_BYTES / ((int64_t)G * c.word);
}

void spmv_plan_panel(const KernelCfg &c, const u32 *row_ptr, DevCsr &D, int64_t hot_rows)
{
	D.panel_rows = 0;
	if (D.rows <= 0 || D.nnz <= 0)
		return;
	if (hot_rows > 0 && hot_rows <= spmv_panel_capacity(c) && hot_rows <= D.cols)
		D.panel_rows = (int)hot_rows;
	/* nnz-balanced row ranges of the eight XCDs (used by k_spmv_panel always, by k_spmv / k_spmv_dot when the slab's
	 * rows have local column supports: DevCsr::xcd_ranges) */
	D.xr_rows[0] = 0;
	{
		/* balanced by the entries the streaming kernel keeps: rows above the outlier threshold go to other launches, and
		 * on a heavy-tailed matrix (renumbered: densest rows first) they would otherwise fill the first XCD's share */
		double kept = 0.0;
		for (int64_t r = 0; r < D.rows; r++) {
			const u32 len = row_ptr[r + 1] - row_ptr[r];
			kept += len <= D.heavy_thr ? (double)len : 0.0;
		}
		double run = 0.0;
		int x = 1;
		for (int64_t r = 0; r < D.rows && x < 8; r++) {
			const u32 len = row_ptr[r + 1] - row_ptr[r];
			run += len <= D.heavy_thr ? (double)len : 0.0;
			while (x < 8 && run >= kept * x / 8.0)
				D.xr_rows[x++] = r + 1;
		}
		for (; x < 8; x++)
			D.xr_rows[x] = D.rows;
	}
	D.xr_rows[8] = D.rows;
	if (const char *e = getenv("BLZ_PANEL_STRIPES"))	/* experiments: 1 = eight equal stripes by row count */
		if (e[0] == '1')
			D.xr_rows[0] = -1;
}

template <typename W, int MERS, bool DOT, int G, int VALS>
static void panel_launch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum, u64 *partial,
			 long long blocks, const DevCtl *ctl, hipStream_t s)
{
	XcdRows xr;
	for (int x = 0; x < 9; x++)
		xr.begin[x] = A.xr_rows[0] < 0 ? A.rows * x / 8 : A.xr_rows[x];
	const size_t lds = (size_t)A.panel_rows * G * sizeof(W);
	auto kern = k_spmv_panel<W, G, MERS, DOT, VALS>;
	static std::atomic<bool> attr_set[64];	/* per instantiation and per device; several threads may ask (setting it twice is harmless) */
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
		(void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_BYTES);
		if (dev >= 0 && dev < 64)
			attr_set[dev].store(true, std::memory_order_release);
	}
	hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(PBLOCK), lds, s, A.row_ptr, (const u32 *)A.col_idx, A.val,
			   A.palette, X, Y, Vd, accum, A.heavy_thr, c.m, partial, xr, (u32)A.panel_rows, ctl);
}

template <typename W, int MERS, bool DOT>
static hipError_t panel_dispatch(const KernelCfg &c, const DevCsr &A, const W *X, W *Y, const W *Vd, int accum,
				 u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	const int G = c.n;	/* spmv_panel_capacity: a power of two */
	long long blocks = ((long long)c.num_cu + 7) & ~7ll;	/* one workgroup of 1024 per CU, whole rounds of the XCDs */
	long long hb = heavy_blocks(c, A, 1 << 30), cb = 0, mb = 0;
	if (DOT) {
		hb = heavy_blocks(c, A, max_blocks / 4);
		cb = A.n_multi ? combine_blocks(A, c.n) : 0;
		mb = A.n_medium ? medium_blocks(c, A) : 0;
		if (blocks + hb + cb + mb > max_blocks)
			return hipErrorInvalidValue;
		*nblocks = (int)(blocks + hb + cb + mb);
	}
	const int vals = A.palette ? V_PACKED : (A.val ? V_ARRAY : V_ONES);
#define PANEL_G(GG)                                                                                              \
	case GG:                                                                                                 \
		if (vals == V_PACKED)                                                                            \
			panel_launch<W, MERS, DOT, GG, V_PACKED>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s); \
		else if (vals == V_ARRAY)                                                                        \
			panel_launch<W, MERS, DOT, GG, V_ARRAY>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s);  \
		else                                                                                             \
			panel_launch<W, MERS, DOT, GG, V_ONES>(c, A, X, Y, Vd, accum, partial, blocks, ctl, s);   \
		if (A.n_heavy || A.n_medium)                                                                     \
			launch_heavy<W, GG, MERS, DOT>(c, A, X, Y, Vd, accum, partial, (int)blocks, hb, ctl, s);     \
		break;
	if constexpr (DOT) {
		switch (G) {
			PANEL_G(1)
			PANEL_G(2)
			PANEL_G(4)
			PANEL_G(8)
		default:
			return hipErrorInvalidValue;
		}
	} else {
		switch (G) {
			PANEL_G(1)
			PANEL_G(2)
			PANEL_G(4)
			PANEL_G(8)
			PANEL_G(16)
			PANEL_G(32)
			PANEL_G(64)
		default:
			return hipErrorInvalidValue;
		}
	}
#undef PANEL_G
	return hipGetLastError();
}

template <typename W, int MERS>
static hipError_t dot_dispatch(const KernelCfg &c, const W *V, const W *AV, int64_t rows, u64 *partial, int max_blocks,
			       int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	const int n = c.n, pairs = n * n;
	if (n == 64) {	/* one partial row per WAVEFRONT: at most max_blocks of them */
		long long blocks = (rows + 4 * 16 - 1) / (4 * 16);	/* >= 16 rows per wavefront */
		const long long cap = std::min<long long>(max_blocks / 4, (long long)c.num_cu * 2);
		blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
		*nblocks = (int)blocks * 4;
		hipLaunchKernelGGL((k_block_dot_64<W, MERS>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, (long long)rows,
				   c.m, partial, ctl);
		return hipGetLastError();
	}
	if (n == 1 || n == 2 || n == 4 || n == 8 || n == 16 || n == 32) {
		const long long gpb = BLOCK / n;
		long long blocks = (rows + gpb * 8 - 1) / (gpb * 8);	/* >= 8 rows per group */
		blocks = blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks);
		*nblocks = (int)blocks;
#define DOT_FAST(NN)                                                                                              \
	case NN:                                                                                                  \
		hipLaunchKernelGGL((k_block_dot_fast<W, MERS, NN>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, \
				   (long long)rows, c.m, partial, ctl);                                            \
		break;
		switch (n) {
			DOT_FAST(1)
			DOT_FAST(2)
			DOT_FAST(4)
			DOT_FAST(8)
			DOT_FAST(16)
			DOT_FAST(32)
		}
#undef DOT_FAST
		return hipGetLastError();
	}
	long long blocks = (rows + 255) / 256;
	blocks = blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks);
	const long long rpb = (rows + blocks - 1) / blocks;
	*nblocks = (int)blocks;
	if (pairs <= BLOCK)
		hipLaunchKernelGGL((k_block_dot<W, MERS, 1>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, (long long)rows,
				   rpb, n, c.m, partial, ctl);
	else if (pairs <= 4 * BLOCK)
		hipLaunchKernelGGL((k_block_dot<W, MERS, 4>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, (long long)rows,
				   rpb, n, c.m, partial, ctl);
	else
		hipLaunchKernelGGL((k_block_dot<W, MERS, 16>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, (long long)rows,
				   rpb, n, c.m, partial, ctl);
	return hipGetLastError();
}

hipError_t launch_block_dot(const KernelCfg &c, const void *V, const void *AV, int64_t rows, u64 *partial,
			    int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s)
{
	if (block_dot_mfma_supported(c) && rows >= 4096)	/* below that one VALU workgroup is quicker than the fold set-up */
		return launch_block_dot_mfma(c, V, AV, rows, partial, max_blocks, nblocks, ctl, s);
	if (c.word == 4)
		return c.mers == 31 ? dot_dispatch<u32, 31>(c, (const u32 *)V, (const u32 *)AV, rows, partial, max_blocks, nblocks, ctl, s)
				    : dot_dispatch<u32, 0>(c, (const u32 *)V, (const u32 *)AV, rows, partial, max_blocks, nblocks, ctl, s);
	return c.mers == 61 ? dot_dispatch<u64, 61>(c, (const u64 *)V, (const u64 *)AV, rows, partial, max_blocks, nblocks, ctl, s)
			    : dot_dispatch<u64, 0>(c, (const u64 *)V, (const u64 *)AV, rows, partial, max_blocks, nblocks, ctl, s);
}

/*
 * out[e] = sum_b partial[b][e] mod p.  A workgroup of 16 wavefronts owns CW = min(8, words) adjacent output words; a lane is
 * (row lane, word): one load instruction of a wavefront covers 64 / CW partial rows x CW words (64-byte pieces of the
 * rows instead of one word out of 64 different lines), the 16 wavefronts take interleaved sets of rows, so for the ~1000
 * partial rows of the fused SpMV every lane has 8 loads, all in flight at once.  Row lanes are added across the
 * wavefront, wavefronts through LDS.  (Round 2: one wavefront per word, lanes over rows: 6.2 us on the GL7d19 shape.)
 */
#define FIN_THREADS 1024
__global__ void __launch_bounds__(FIN_THREADS)
k_dot_finalize(const u64 *__restrict__ partial, int nblocks, int words, int cw_log2, u64 p, u64 *__restrict__ out,
	       const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	__shared__ u64 part[FIN_THREADS / 64][8];
	const int CW = 1 << cw_log2, RL = 64 >> cw_log2;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wl = lane & (CW - 1), rl = lane >> cw_log2;
	const int e = blockIdx.x * CW + wl, step = (FIN_THREADS / 64) * RL;
	u64 s = 0;
	int b = wave * RL + rl;
	for (; b + 7 * step < nblocks; b += 8 * step) {	/* eight independent loads in flight, then the adds */
		u64 x[8];
#pragma unroll
		for (int q = 0; q < 8; q++)
			x[q] = partial[(size_t)(b + q * step) * words + e];
#pragma unroll
		for (int q = 0; q < 8; q++)
			s = addmod(s, x[q], p);
	}
	for (; b < nblocks; b += step)
		s = addmod(s, partial[(size_t)b * words + e], p);
	for (int off = 32; off >= CW; off >>= 1)
		s = addmod(s, shfl_xor64(s, off), p);
	if (lane < CW)
		part[wave][lane] = s;
	__syncthreads();
	if (threadIdx.x < CW) {
		u64 t = 0;
#pragma unroll
		for (int w = 0; w < FIN_THREADS / 64; w++)
			t = addmod(t, part[w][threadIdx.x], p);
		out[e] = t;
	}
}

hipError_t launch_dot_finalize(const KernelCfg &c, const u64 *partial, int nblocks, u64 *out, const DevCtl *ctl,
			       hipStream_t s)
{
	const int words = 2 * c.n * c.n;	/* a power of two times 2: 2, 8, 18 (n = 3, unpadded) ...: CW must divide it */
	int cw_log2 = 3;
	while ((words & ((1 << cw_log2) - 1)) != 0)
		cw_log2--;
	hipLaunchKernelGGL(k_dot_finalize, dim3(words >> cw_log2), dim3(FIN_THREADS), 0, s, partial, nblocks, words, cw_log2, c.m.p, out,
			   ctl);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------- semi_inverse */

/* invmod(), sequential/lanczos_modp.c:318-336 (extended Euclid; |t| <= p < 2^62 throughout). */
__device__ static u64 dev_invmod(u64 a, u64 p)
{
	long long t = 0, nt = 1, r = (long long)p, nr = (long long)(a % p);
	while (nr != 0) {
		const long long q = r / nr;
		long long s = nt;
		nt = t - q * nt;
		t = s;
		s = nr;
		nr = r - q * nr;
		r = s;
	}
	if (t < 0)
		t += (long long)p;
	return (u64)t;
}

/* The same inverse for p = 2^61 - 1 (a prime) as a^(p-2): p - 2 = 2^61 - 3 is 59 ones, a zero and a one in binary, so
 * 60 squarings and 10 products along the chain 1, 2, 3, 6, 12, 24, 48, 54, 57, 59 (exponents 2^k - 1) do it.  The
 * inverse mod a prime is unique, so the word is the one the extended Euclid returns; a serial chain of Mersenne
 * multiplications has no 64-bit division in it (about 36 of them in the Euclid, each some hundred instructions here). */
/* x^2 mod 2^61 - 1 for x < 2^61 from three 32 x 32 products: x = x1 2^32 + x0 (x1 < 2^29), x^2 = x0^2 + x0 x1 2^33 + x1^2 2^64,
 * and 2^61 = 1: x1^2 2^64 = 8 x1^2 (< 2^61); x0 x1 2^33 with x0 x1 = mh 2^28 + ml is mh + ml 2^33 (< 2^61); x0^2 folds once.
 * About half the instructions of the general 128-bit product and fold: the 60 squarings of the inversion below were 4.1 of the
 * semi-inverse kernel's 18 us (in-kernel clock, round 3). */
MODP_DEV u64 sqrmod61(u64 x)
{
	const u64 P = (1ull << 61) - 1;
	const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
	const u64 lo = (u64)x0 * x0, mid = (u64)x0 * x1, hi = (u64)x1 * x1;
	const u64 t = (lo & P) + (lo >> 61) + (mid >> 28) + ((mid & ((1ull << 28) - 1)) << 33) + (hi << 3);	/* < 4 * 2^61 */
	const u64 r = (t & P) + (t >> 61);
	return r >= P ? r - P : r;
}

/* x y mod 2^61 - 1 for RESIDUES x, y < 2^61, from four 32 x 32 products in the same way (x1, y1 < 2^29: x1 y1 2^64 = 8 x1 y1;
 * the middle terms x0 y1 + x1 y0 = mh 2^29 + ml times 2^32 are mh + ml 2^32) */
MODP_DEV u64 mulmod61_res(u64 x, u64 y)
{
	const u64 P = (1ull << 61) - 1;
	const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
	const u64 lo = (u64)x0 * y0, mid = (u64)x0 * y1 + (u64)x1 * y0, hi = (u64)x1 * y1;	/* mid < 2^62 */
	const u64 t = (lo & P) + (lo >> 61) + (mid >> 29) + ((mid & ((1ull << 29) - 1)) << 32) + (hi << 3);	/* < 4 * 2^61 */
	const u64 r = (t & P) + (t >> 61);
	return r >= P ? r - P : r;
}

__device__ static u64 dev_invmod_mers61(u64 a, const ModP &m)
{
	(void)m;
	auto sqn = [&](u64 x, int k) {
		for (int i = 0; i < k; i++)
			x = sqrmod61(x);
		return x;
	};
	const u64 x1 = a;
	const u64 x2 = mulmod61_res(sqn(x1, 1), x1);
	const u64 x3 = mulmod61_res(sqn(x2, 1), x1);
	const u64 x6 = mulmod61_res(sqn(x3, 3), x3);
	const u64 x12 = mulmod61_res(sqn(x6, 6), x6);
	const u64 x24 = mulmod61_res(sqn(x12, 12), x12);
	const u64 x48 = mulmod61_res(sqn(x24, 24), x24);
	const u64 x54 = mulmod61_res(sqn(x48, 6), x6);
	const u64 x57 = mulmod61_res(sqn(x54, 3), x3);
	const u64 x59 = mulmod61_res(sqn(x57, 2), x2);
	return mulmod61_res(sqn(x59, 2), x1);
}

/* product of two residues with the lean form where the prime has one */
template <int MERS>
MODP_DEV u64 mulres(u64 x, u64 y, const ModP &m)
{
	if (MERS == 61)
		return mulmod61_res(x, y);
	return mulmod<MERS>(x, y, m);
}

/* ... and for p = 2^31 - 1: p - 2 = 2^31 - 3 is 29 ones, a zero and a one: 30 squarings and 8 products along
 * 1, 2, 3, 6, 12, 24, 27, 29.  The extended Euclid costs ~20 steps with a 64-bit division each (a hundred instructions
 * apiece on this chip): 5 of the 9.5 us of the semi-inverse kernel on the relat8 shape (round 3). */
__device__ static u64 dev_invmod_mers31(u64 a, const ModP &m)
{
	auto sqn = [&](u64 x, int k) {
		for (int i = 0; i < k; i++)
			x = mulmod<31>(x, x, m);
		return x;
	};
	const u64 x1 = a;
	const u64 x2 = mulmod<31>(sqn(x1, 1), x1, m);
	const u64 x3 = mulmod<31>(sqn(x2, 1), x1, m);
	const u64 x6 = mulmod<31>(sqn(x3, 3), x3, m);
	const u64 x12 = mulmod<31>(sqn(x6, 6), x6, m);
	const u64 x24 = mulmod<31>(sqn(x12, 12), x12, m);
	const u64 x27 = mulmod<31>(sqn(x24, 3), x3, m);
	const u64 x29 = mulmod<31>(sqn(x27, 2), x2, m);
	return mulmod<31>(sqn(x29, 2), x1, m);
}

template <int MERS>
__device__ static u64 dev_invmod_any(u64 a, const ModP &m)
{
	if (MERS == 61)
		return dev_invmod_mers61(a, m);
	if (MERS == 31)
		return dev_invmod_mers31(a, m);
	return dev_invmod(a, m.p);
}

/*
 * One Gauss-Jordan sweep with the pivot rule of sequential/lanczos_modp.c:351-381 / :393-436: for column
 * j take the first non-zero entry in rows j..n-1; a column without one is skipped (row j is then never
 * used as a pivot row); swap the pivot row into row j; eliminate column j everywhere else.
 *
 * The reference scales each pivot row by the inverse of its pivot (one invmod per pivot, and the
 * extended Euclid is by far the slowest thing in this kernel).  Here the sweep is FRACTION-FREE: every
 * row is carried as s_i * (the reference's row) with a non-zero scalar s_i,
 *     pivot row:   s_j  <- its entry in column j              (instead of dividing the row by it)
 *     other rows:  row_i <- s_j*row_i - row_i[j]*row_j ,  s_i <- s_i*s_j     (only when row_i[j] != 0)
 * Zero patterns are unchanged (p prime: no zero divisors), so the pivot choices -- hence d -- are the
 * reference's, and dividing row i by s_i at the end gives exactly the reference's rows.  The n scalars
 * are inverted together with ONE invmod (prefix products).  lanes = (row-in-pass, column).
 */
template <int MERS>
__device__ static int ff_sweep(u64 *A, u64 *Wm, u64 *S, int n, int G, const ModP &m, u64 *mask)
{
	/* T = blockDim.x threads (a multiple of 64, at least G*... one wavefront for small n): thread = (row of the pass,
	 * column); the pivot search is a ballot in wavefront 0 (n <= 64 rows), published through LDS */
	const int lane = threadIdx.x, li = lane / G, k = lane % G, RP = (int)blockDim.x / G;
	__shared__ unsigned long long cand_sh;
	int found = 0;
	u64 bits = 0;
	for (int j = 0; j < n; j++) {
		__syncthreads();
		const u64 probe = (lane < n && lane >= j) ? A[lane * n + j] : 0;
		const unsigned long long mine = __ballot(probe != 0);
		if (lane == 0)
			cand_sh = mine;
		__syncthreads();
		const unsigned long long cand = cand_sh;
		if (cand == 0)
			continue;
		const int piv = __ffsll(cand) - 1;
		bits |= 1ull << j;
		found++;
		__syncthreads();
		if (piv != j) {
			if (lane < n) {
				u64 x = A[piv * n + lane];
				A[piv * n + lane] = A[j * n + lane];
				A[j * n + lane] = x;
				if (Wm) {
					x = Wm[piv * n + lane];
					Wm[piv * n + lane] = Wm[j * n + lane];
					Wm[j * n + lane] = x;
				}
			}
			if (S && lane == 0)
				S[piv] = S[j];
		}
		__syncthreads();
		const u64 pv = A[j * n + j];
		if (S && lane == 0)
			S[j] = pv;
		for (int ib = 0; ib < n; ib += RP) {
			const int i = ib + li;
			const bool act = i < n && i != j && k < n;
			u64 mult = 0, x = 0, y = 0, wx = 0, wy = 0;
			if (act) {
				mult = A[i * n + j];
				x = A[i * n + k];
				y = A[j * n + k];
				if (Wm) {
					wx = Wm[i * n + k];
					wy = Wm[j * n + k];
				}
			}
			__syncthreads();
			if (act && mult != 0) {
				const u64 neg = m.p - mult;
				Acc acc;
				acc_zero(acc);
				acc_mac64(acc, pv, x);
				acc_set(acc, acc_reduce<MERS>(acc, m));
				acc_mac64(acc, neg, y);
				A[i * n + k] = acc_reduce<MERS>(acc, m);
				if (Wm) {
					acc_zero(acc);
					acc_mac64(acc, pv, wx);
					acc_set(acc, acc_reduce<MERS>(acc, m));
					acc_mac64(acc, neg, wy);
					Wm[i * n + k] = acc_reduce<MERS>(acc, m);
				}
				if (S && k == 0)
					S[i] = mulmod<MERS>(S[i], pv, m);
			}
		}
	}
	__syncthreads();
	*mask = bits;
	return found;
}

/*
 * semi_inverse(), sequential/lanczos_modp.c:342-438, plus the two n x n coefficient matrices that
 * orthogonalize() derives from it (:460-475), so that the row kernel only streams.
 * small = [vtAv | vtAAv | winv | d | c | vtAvd].  One workgroup.
 * `sums` = where vtAv | vtAAv come from: `small` itself, or the landing place of the all-reduce over the ranks (sums of
 * residues); `small` receives their residues, and only if the stop flag is down -- the iterations a batch enqueues past
 * the stop run the all-reduce again, and `small` must stay what the last real iteration left (round 3).
 * IMGN = 16: the kernel also builds the coefficient image of the matrix-core block update (ortho_img.h) from the
 * coefficients it has just computed: one launch where round 2 had two (k_ortho_mfma_prep).
 */
template <int MERS, int IMGN>
__global__ void __launch_bounds__(1024)
k_semi_inverse(const u64 *__restrict__ sums, u64 *__restrict__ small, DevCtl *__restrict__ ctl, int n, int G, ModP m, int in_loop,
	       unsigned char *__restrict__ img)
{
	/* in_loop = 0: stand-alone call (blz_semi_inverse): neither obeys nor sets the sticky stop flag */
	if (in_loop && ctl->stop)
		return;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const int nn = n * n;
	u64 *A = (u64 *)smem_raw, *Wm = A + nn, *S = Wm + nn, *Pre = S + n;
	u64 *coef = Pre + n;			/* IMGN: the six-panel layout of `small`, panels winv, c, vtAvd filled */
	const int lane = threadIdx.x, T = (int)blockDim.x;
	u64 *vtAv = small, *vtAAv = small + nn, *winv = small + 2 * nn, *dvec = small + 3 * nn;
	u64 *cmat = small + 4 * nn, *vtAvd = small + 5 * nn;

	/* inputs may be sums of per-rank residues: bring them back into [0,p) */
	for (int e = lane; e < nn; e += T) {
		const u64 x = reduce128<MERS>(0, sums[e], m), y = reduce128<MERS>(0, sums[nn + e], m);
		vtAv[e] = x;
		vtAAv[e] = y;
		A[e] = x;
	}
	u64 sel = 0, dbits = 0;
	/* one sweep on the unmasked matrix while every column pivots (see k_semi_inverse_reg); the reference's two otherwise */
	for (int e = lane; e < nn; e += T)
		Wm[e] = (e / n == e % n) ? 1 : 0;
	if (lane < n)
		S[lane] = 1;
	__syncthreads();
	int npiv = ff_sweep<MERS>(A, Wm, S, n, G, m, &dbits);
	if (npiv != n) {
		for (int e = lane; e < nn; e += T)
			A[e] = vtAv[e];
		__syncthreads();
		ff_sweep<MERS>(A, nullptr, nullptr, n, G, m, &sel);		/* phase 1, :349-382: which columns */
		for (int e = lane; e < nn; e += T) {				/* :384-388 */
			const int i = e / n, j = e % n;
			const bool both = ((sel >> i) & 1) && ((sel >> j) & 1);
			A[e] = both ? vtAv[e] : 0;
			Wm[e] = (i == j && ((sel >> i) & 1)) ? 1 : 0;
		}
		if (lane < n)
			S[lane] = 1;
		__syncthreads();
		npiv = ff_sweep<MERS>(A, Wm, S, n, G, m, &dbits);	/* phase 2, :389-436 */
	}
	/* 1/s_i for all rows from one inversion: Pre[i] = s_0..s_i, then walk back */
	if (lane == 0) {
		u64 run = 1;
		for (int i = 0; i < n; i++) {
			run = mulmod<MERS>(run, S[i], m);
			Pre[i] = run;
		}
		u64 inv = dev_invmod_any<MERS>(run, m);
		for (int i = n - 1; i >= 0; i--) {
			const u64 si = S[i];
			S[i] = i ? mulmod<MERS>(inv, Pre[i - 1], m) : inv;
			inv = mulmod<MERS>(inv, si, m);
		}
	}
	__syncthreads();
	for (int e = lane; e < nn; e += T) {
		const u64 w = mulmod<MERS>(Wm[e], S[e / n], m);
		Wm[e] = w;
		winv[e] = w;
		if (IMGN)
			coef[2 * nn + e] = w;
	}
	if (lane < n)
		dvec[lane] = (dbits >> lane) & 1;
	__syncthreads();
	/* c = -(winv * spliced), vtAvd = -vtAv on the selected columns (:462-475), canonical */
	for (int e = lane; e < nn; e += T) {
		const int i = e / n, j = e % n;
		const bool dj = (dbits >> j) & 1;
		const u64 *sp = dj ? vtAAv : vtAv;
		Acc acc;
		acc_zero(acc);
		u32 cnt = 0;
		for (int k = 0; k < n; k++) {
			acc_mac64(acc, Wm[i * n + k], sp[k * n + j]);
			if (++cnt == m.chunk) {
				cnt = 0;
				acc_set(acc, acc_reduce<MERS>(acc, m));
			}
		}
		const u64 r = acc_reduce<MERS>(acc, m);
		const u64 ce = r ? m.p - r : 0;
		cmat[e] = ce;
		const u64 x = vtAv[e];
		const u64 ve = (dj && x) ? m.p - x : 0;
		vtAvd[e] = ve;
		if (IMGN) {
			coef[4 * nn + e] = ce;
			coef[5 * nn + e] = ve;
		}
	}
	if (lane == 0) {
		ctl->npiv = npiv;
		if (in_loop) {
			if (npiv == 0)
				ctl->stop = 1;
			else
				ctl->iterations += 1;
		}
	}
	if constexpr (IMGN == 16) {
		/* (the image of an iteration that stops is never read: every later kernel is a no-op) */
		__syncthreads();
		ortho_image_build<16>(coef, img, (int *)(coef + 6 * nn), lane, T);
	}
}

/* 64-bit read of a lane that is the same for the whole wavefront: two v_readlane_b32, no LDS crossbar round trip */
MODP_DEV u64 readlane64(u64 x, int lane)
{
	const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x, lane), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), lane);
	return ((u64)hi << 32) | lo;
}

/* lane (i, k) of the (rows x 2^LG) lane grid reads lane (i, K): a VALU move with a DPP modifier (row_newbcast inside the
 * 16-lane DPP row for groups of 8, quad_perm for groups of 4 and 2), no LDS crossbar */
template <int K, int LG>
MODP_DEV u32 col_bcast32(u32 x)
{
	if constexpr (LG == 3)
		return group_bcast32<K & 7, 8>(x);
	else if constexpr (LG == 2)
		return (u32)__builtin_amdgcn_update_dpp(0, (int)x, (K & 3) * 0x55, 0xF, 0xF, false);
	else if constexpr (LG == 1)
		return (u32)__builtin_amdgcn_update_dpp(0, (int)x, (K & 1) * 0x05 + (2 + (K & 1)) * 0x50, 0xF, 0xF, false);
	else
		return x;
}

template <int K, int LG>
MODP_DEV u64 col_bcast(u64 x)
{
	return ((u64)col_bcast32<K, LG>((u32)(x >> 32)) << 32) | col_bcast32<K, LG>((u32)x);
}

/* the same with the column picked at run time (uniform over the wavefront: a scalar branch) */
template <int LG>
MODP_DEV u64 col_bcast_at(u64 x, int j)
{
	constexpr int G = 1 << LG;
	u64 r = x;
	static_for<0, G>([&](auto kc) {
		constexpr int K = decltype(kc)::value;
		if (j == K)
			r = col_bcast<K, LG>(x);
	});
	return r;
}

/*
 * Register-resident form for n*n <= 64 (n <= 8): lane (i,k) = (lane >> LG, lane & (G-1)) of the single wavefront
 * holds A[i][k], W[i][k] and the scalar of row i; pivot search is one ballot, row swaps / broadcasts are cross-lane
 * reads.  No LDS round trips: the serial chain per column is one crossbar read and two MACs (round 3: the pivot comes
 * by v_readlane, the multiplier of a row by a DPP move, and rows are only exchanged when the pivot is not on the
 * diagonal -- almost never mod a large prime).  Same fraction-free sweep, same pivot rule, same single inversion as
 * k_semi_inverse.
 */
template <int MERS, int LG>
__device__ static int ff_sweep_reg(u64 &a, u64 *w, u64 *s, int n, const ModP &m, u64 *mask)
{
	constexpr int G = 1 << LG;
	const int lane = threadIdx.x, i = lane >> LG, k = lane & (G - 1);
	const bool valid = i < n && k < n;
	int found = 0;
	u64 bits = 0;
	for (int j = 0; j < n; j++) {
		const unsigned long long cand = __ballot(valid && k == j && i >= j && a != 0);
		if (cand == 0)
			continue;
		const int piv = (__ffsll(cand) - 1) >> LG;
		bits |= 1ull << j;
		found++;
		if (piv != j) {		/* swap rows piv <-> j */
			const int from = ((i == j ? piv : (i == piv ? j : i)) << LG) + k;
			a = shfl64(a, from);
			if (w)
				*w = shfl64(*w, from);
			if (s)
				*s = shfl64(*s, from);
		}
		const u64 pv = readlane64(a, (j << LG) + j);
		if (s && i == j)
			*s = pv;
		const u64 mult = col_bcast_at<LG>(a, j), y = shfl64(a, (j << LG) + k);
		const u64 wy = w ? shfl64(*w, (j << LG) + k) : 0;
		if (valid && i != j && mult != 0) {
			const u64 neg = m.p - mult;
			/* pv * a + neg * y is one 128-bit sum when two products fit between reductions (every prime below 2^61, and
			 * 2^61 - 1 itself): one reduction per update instead of two */
			const bool two = m.chunk >= 2;
			Acc acc;
			acc_zero(acc);
			acc_mac64(acc, pv, a);
			if (!two)
				acc_set(acc, acc_reduce<MERS>(acc, m));
			acc_mac64(acc, neg, y);
			a = acc_reduce<MERS>(acc, m);
			if (w) {
				acc_zero(acc);
				acc_mac64(acc, pv, *w);
				if (!two)
					acc_set(acc, acc_reduce<MERS>(acc, m));
				acc_mac64(acc, neg, wy);
				*w = acc_reduce<MERS>(acc, m);
			}
			if (s)
				*s = mulres<MERS>(*s, pv, m);
		}
	}
	*mask = bits;
	return found;
}

/* IMG: n = 8, p = 2^61 - 1: fifteen more wavefronts wait at a barrier while wavefront 0 runs the sweep, then all sixteen
 * build the coefficient image of the matrix-core block update from the coefficients wavefront 0 left in LDS.
 * (Tried in round 3 and not kept: the partial rows of the inner products summed by this kernel itself, 1024 threads, where
 * they are few -- relat8 shape, 256 rows of 32 words: 12.4 us against 4.4 + 8.5 for k_dot_finalize and this kernel apart,
 * 57.4 against 56.8 us per iteration.) */
template <int MERS, int LG, bool IMG>
__global__ void __launch_bounds__(IMG ? 1024 : 64)
k_semi_inverse_reg(const u64 *__restrict__ sums, u64 *__restrict__ small, DevCtl *__restrict__ ctl, int n, ModP m, int in_loop,
		   unsigned char *__restrict__ img)
{
	constexpr int G = 1 << LG, T = IMG ? 1024 : 64;
	const int nn = n * n, lane = threadIdx.x, i = (lane & 63) >> LG, k = lane & (G - 1);
	const bool valid = lane < 64 && i < n && k < n;
	const int e = valid ? i * n + k : 0;
	/* the three loads go out together; the flag is looked at when they are back */
	const int stop = in_loop ? ctl->stop : 0;
	const u64 xr = sums[e], yr = sums[nn + e];
	if (stop)
		return;
	__shared__ u64 coef[IMG ? 6 * 64 : 1];
	__shared__ int init_sh[IMG ? OG<8>::NE : 1];
	if (lane < 64) {
		u64 *vtAv = small, *vtAAv = small + nn, *winv = small + 2 * nn, *dvec = small + 3 * nn;
		u64 *cmat = small + 4 * nn, *vtAvd = small + 5 * nn;
		/* inputs may be sums of per-rank residues: bring them back into [0,p) */
		const u64 x0 = valid ? reduce128<MERS>(0, xr, m) : 0, y0 = valid ? reduce128<MERS>(0, yr, m) : 0;
		if (valid) {
			vtAv[e] = x0;
			vtAAv[e] = y0;
		}
		/* The reference sweeps twice (:349-382 finds the set of columns that pivot, :389-436 eliminates again on the matrix
		 * masked to that set, carrying the identity).  While vtAv is non-singular -- every iteration but the last few -- the
		 * set is every column and the mask changes nothing, so the second sweep on the UNMASKED matrix is both at once: it
		 * performs the first sweep's operations on `a`, and finds n pivots exactly when the first sweep would.  Only when
		 * it finds fewer are the reference's two sweeps run (round 3: 2.9 of the kernel's 18 us). */
		u64 sel = 0, dbits = 0;
		u64 a = x0, w = (valid && i == k) ? 1 : 0, s = 1;
		int npiv = ff_sweep_reg<MERS, LG>(a, &w, &s, n, m, &dbits);
		if (npiv != n) {
			a = x0;
			ff_sweep_reg<MERS, LG>(a, nullptr, nullptr, n, m, &sel);		/* phase 1, :349-382 */
			const bool both = valid && ((sel >> i) & 1) && ((sel >> k) & 1);
			a = both ? x0 : 0;							/* :384-388 */
			w = (valid && i == k && ((sel >> i) & 1)) ? 1 : 0;
			s = 1;
			npiv = ff_sweep_reg<MERS, LG>(a, &w, &s, n, m, &dbits);		/* phase 2, :389-436 */
		}
		/* one inversion for all row scalars: the lanes of row i multiply the OTHER rows' scalars (q), q * s_i is the product
		 * of all of them -- the same in every lane --, and 1 / s_i = q / (that product): n + 2 products around the
		 * inversion where the prefix walk had 3 n */
		u64 q = 1;
#pragma unroll
		for (int r = 0; r < G; r++) {
			const u64 sr = readlane64(s, r << LG);
			if (r < n) {
				const u64 t = mulres<MERS>(q, sr, m);
				q = r == i ? q : t;
			}
		}
		const u64 all = readlane64(mulres<MERS>(q, s, m), 0);
		const u64 mine = mulres<MERS>(dev_invmod_any<MERS>(all, m), q, m);
		const u64 wn = valid ? mulres<MERS>(w, mine, m) : 0;
		if (valid)
			winv[e] = wn;
		if (lane < n)
			dvec[lane] = (dbits >> lane) & 1;
		/* c = -(winv * spliced), vtAvd = -vtAv on the selected columns (:462-475), canonical; lane (i,k) owns c[i][k] */
		const bool dk = (dbits >> k) & 1;
		const u64 sp = dk ? y0 : x0;						/* spliced[i][k] */
		Acc acc;
		acc_zero(acc);
		u32 cnt = 0;
		static_for<0, G>([&](auto qc) {
			constexpr int Q = decltype(qc)::value;
			const u64 wiq = col_bcast<Q, LG>(wn), sqk = shfl64(sp, (Q << LG) + k);
			if (Q < n) {
				acc_mac64(acc, wiq, sqk);
				if (++cnt == m.chunk) {
					cnt = 0;
					acc_set(acc, acc_reduce<MERS>(acc, m));
				}
			}
		});
		u64 ce = 0, ve = 0;
		if (valid) {
			const u64 r = acc_reduce<MERS>(acc, m);
			ce = r ? m.p - r : 0;
			ve = (dk && x0) ? m.p - x0 : 0;
			cmat[e] = ce;
			vtAvd[e] = ve;
		}
		if (IMG) {		/* n = 8 = G: e = lane */
			coef[2 * 64 + lane] = wn;
			coef[4 * 64 + lane] = ce;
			coef[5 * 64 + lane] = ve;
		}
		if (lane == 0) {
			ctl->npiv = npiv;
			if (in_loop) {
				if (npiv == 0)
					ctl->stop = 1;
				else
					ctl->iterations += 1;
			}
		}
	}
	if constexpr (IMG) {
		__syncthreads();
		ortho_image_build<8>(coef, img, init_sh, lane, T);
	}
}

hipError_t launch_semi_inverse(const KernelCfg &c, const u64 *sums, u64 *small, DevCtl *ctl, int in_loop, int build_img,
			       hipStream_t s)
{
	int G = 1, LG = 0;
	while (G < c.n) {
		G <<= 1;
		LG++;
	}
	unsigned char *img = (unsigned char *)c.mfma_img;
	if (build_img && !(ortho_mfma_supported(c) && img))
		return hipErrorInvalidValue;
	if (c.n <= 8) {		/* the whole n x n problem fits one wavefront's registers */
#define SEMI_REG(MM, LL)                                                                                          \
	hipLaunchKernelGGL((k_semi_inverse_reg<MM, LL, false>), dim3(1), dim3(64), 0, s, sums, small, ctl, c.n, c.m, in_loop, img)
#define SEMI_REG_LG(MM)                                                                                           \
	do {                                                                                                      \
		if (LG == 0) SEMI_REG(MM, 0);                                                                     \
		else if (LG == 1) SEMI_REG(MM, 1);                                                                \
		else if (LG == 2) SEMI_REG(MM, 2);                                                                \
		else SEMI_REG(MM, 3);                                                                             \
	} while (0)
		if (build_img)		/* n = 8, p = 2^61 - 1 (ortho_mfma_supported): 16 wavefronts, one 16-byte piece of the image each */
			hipLaunchKernelGGL((k_semi_inverse_reg<61, 3, true>), dim3(1), dim3(1024), 0, s, sums, small, ctl, c.n, c.m, in_loop,
					   img);
		else if (c.mers == 61)
			SEMI_REG_LG(61);
		else if (c.mers == 31)
			SEMI_REG_LG(31);
		else
			SEMI_REG_LG(0);
#undef SEMI_REG_LG
#undef SEMI_REG
		return hipGetLastError();
	}
	size_t lds = ((size_t)2 * c.n * c.n + 2 * c.n) * sizeof(u64);
	/* one thread per matrix entry up to a full workgroup: every elimination step is then one pass (n = 64: four) */
	int threads = G * G > 1024 ? 1024 : (G * G < 64 ? 64 : G * G);
#define SEMI(MM, II)                                                                                               \
	do {                                                                                                       \
		if (lds > 48 * 1024)                                                                               \
			hipFuncSetAttribute((const void *)k_semi_inverse<MM, II>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
					    (int)lds);                                                             \
		hipLaunchKernelGGL((k_semi_inverse<MM, II>), dim3(1), dim3(threads), lds, s, sums, small, ctl, c.n, G, c.m, in_loop, img); \
	} while (0)
	if (build_img) {	/* n = 16, p = 2^61 - 1: all 16 wavefronts for the 49 KB image */
		lds += (size_t)6 * c.n * c.n * sizeof(u64) + OG<16>::NE * sizeof(int);
		threads = 1024;
		SEMI(61, 16);
	} else if (c.mers == 61)
		SEMI(61, 0);
	else if (c.mers == 31)
		SEMI(31, 0);
	else
		SEMI(0, 0);
#undef SEMI
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ orthogonalize */

/*
 * orthogonalize() rows, sequential/lanczos_modp.c:478-491, with the copy v <- tmp (:655-656)
 * folded in: the update is row-local, so v' and p' overwrite v and p in place.
 *   v'[r,j] = (d[j] ? Av[r,j] : v[r,j]) + sum_k v[r,k] c[k,j] + sum_k p[r,k] vtAvd[k,j]
 *   p'[r,j] = (d[j] ? 0 : p[r,j])       + sum_k v[r,k] winv[k,j]
 * One group of G lanes per row; the three n x n matrices and the staged rows live in LDS.
 */
template <typename W, int MERS>
__global__ void __launch_bounds__(BLOCK)
k_orthogonalize(W *__restrict__ V, const W *__restrict__ AV, W *__restrict__ P, long long rows, int n, int G,
		ModP m, const u64 *__restrict__ small, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const int nn = n * n;
	u64 *sc = (u64 *)smem_raw, *svd = sc + nn, *sw = svd + nn;
	u64 *sv = sw + nn;			/* [groups][n] */
	const int gpb = BLOCK / G;
	u64 *sp = sv + gpb * n;
	const int t = threadIdx.x, g = t / G, lane = t % G;
	for (int e = t; e < nn; e += BLOCK) {
		sw[e] = small[2 * nn + e];
		sc[e] = small[4 * nn + e];
		svd[e] = small[5 * nn + e];
	}
	const bool dj = lane < n ? small[3 * nn + lane] != 0 : false;
	for (long long base = (long long)blockIdx.x * gpb; base < rows; base += (long long)gridDim.x * gpb) {
		const long long r = base + g;
		const bool ok = r < rows && lane < n;
		u64 vv = 0, aa = 0, pp = 0;
		if (ok) {
			vv = V[(size_t)r * n + lane];
			aa = AV[(size_t)r * n + lane];
			pp = P[(size_t)r * n + lane];
			sv[g * n + lane] = vv;
			sp[g * n + lane] = pp;
		}
		__syncthreads();
		if (ok) {
			Acc av, ap;
			acc_set(av, dj ? aa : vv);
			acc_set(ap, dj ? 0 : pp);
			u32 cv = 0, cp = 0;
			for (int k = 0; k < n; k++) {
				const u64 vk = sv[g * n + k], pk = sp[g * n + k];
				acc_mac64(av, vk, sc[k * n + lane]);
				if (++cv == m.chunk) {
					cv = 0;
					acc_set(av, acc_reduce<MERS>(av, m));
				}
				acc_mac64(av, pk, svd[k * n + lane]);
				if (++cv == m.chunk) {
					cv = 0;
					acc_set(av, acc_reduce<MERS>(av, m));
				}
				acc_mac64(ap, vk, sw[k * n + lane]);
				if (++cp == m.chunk) {
					cp = 0;
					acc_set(ap, acc_reduce<MERS>(ap, m));
				}
			}
			V[(size_t)r * n + lane] = (W)acc_reduce<MERS>(av, m);
			P[(size_t)r * n + lane] = (W)acc_reduce<MERS>(ap, m);
		}
		__syncthreads();
	}
}

/*
 * Fast path for n = NT in {1,2,4,8,16} (needs m.chunk >= 2*NT so that a row's sums fit one reduction):
 * lane j of a group keeps column j of c, vtAvd and winv in registers (3*NT words), loads v[r,j], Av[r,j],
 * p[r,j] with three coalesced loads, and receives v[r,k], p[r,k] by broadcast inside the group.
 * No LDS, no barrier: pure streaming with 3*NT 64x64-bit MACs per output pair.
 */
template <typename W, int MERS, int NT>
__global__ void __launch_bounds__(BLOCK)
k_orthogonalize_fast(W *__restrict__ V, const W *__restrict__ AV, W *__restrict__ P, long long rows, ModP m,
		     const u64 *__restrict__ small, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int GPB = BLOCK / NT, NN = NT * NT;
	const int t = threadIdx.x, lane = t & 63, j = t & (NT - 1), gbase = lane - j;
	using CW = typename std::conditional<sizeof(W) == 4, u32, u64>::type;	/* coefficients are residues too */
	CW cc[NT], vd[NT], ww[NT];
#pragma unroll
	for (int k = 0; k < NT; k++) {
		ww[k] = (CW)small[2 * NN + k * NT + j];
		cc[k] = (CW)small[4 * NN + k * NT + j];
		vd[k] = (CW)small[5 * NN + k * NT + j];
	}
	const bool dj = small[3 * NN + j] != 0;
	const long long g0 = (long long)blockIdx.x * GPB + t / NT, ng = (long long)gridDim.x * GPB;
	const int src0 = gbase * 4;			/* byte address of lane 0 of the group for ds_bpermute */
	for (long long r = g0; r < rows; r += ng) {
		const size_t at = (size_t)r * NT + j;
		const u64 vv = V[at], aa = AV[at], pp = P[at];
		typename std::conditional<sizeof(W) == 8, AccL, AccS>::type av, ap;
		acc_set(av, dj ? aa : vv);
		acc_set(ap, dj ? 0 : pp);
		if constexpr (NT == 16) {	/* one DPP row = one group; at NT = 8 the two half-row moves cost more than they save */
			static_for<0, NT>([&](auto kc) {
				constexpr int k = decltype(kc)::value;
				const u64 vk = group_bcast<k, NT, sizeof(W) == 4>(vv), pk = group_bcast<k, NT, sizeof(W) == 4>(pp);
				acc_mac64(av, vk, cc[k]);
				acc_mac64(av, pk, vd[k]);
				acc_mac64(ap, vk, ww[k]);
			});
		} else {
#pragma unroll
			for (int k = 0; k < NT; k++) {
				const u64 vk = NT == 1 ? vv : bperm_word<sizeof(W) == 4>(vv, src0 + 4 * k);
				const u64 pk = NT == 1 ? pp : bperm_word<sizeof(W) == 4>(pp, src0 + 4 * k);
				acc_mac64(av, vk, cc[k]);
				acc_mac64(av, pk, vd[k]);
				acc_mac64(ap, vk, ww[k]);
			}
		}
		V[at] = (W)acc_reduce<MERS>(av, m);
		P[at] = (W)acc_reduce<MERS>(ap, m);
	}
}

/*
 * n = 32: one wavefront per block row, lane = (h, j): output column j, and the half h of the sum over k (k in
 * [16h, 16h + 16)), so that a lane keeps 3 * 16 coefficient words like the n = 16 kernel instead of 3 * 32.  Each
 * 16-lane DPP row loads the 16 words v[r, 16h ..], p[r, 16h ..] it has to broadcast (row_newbcast); the two halves
 * are reduced separately and added mod p across lane L and L ^ 32.
 */
template <typename W, int MERS>
__global__ void __launch_bounds__(BLOCK)
k_orthogonalize_32(W *__restrict__ V, const W *__restrict__ AV, W *__restrict__ P, long long rows, ModP m,
		   const u64 *__restrict__ small, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = 32, NN = NT * NT, HK = 16;
	constexpr bool NARROW = sizeof(W) == 4;
	const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, k0 = HK * h;
	using CW = typename std::conditional<NARROW, u32, u64>::type;
	CW cc[HK], vd[HK], ww[HK];
#pragma unroll
	for (int kk = 0; kk < HK; kk++) {
		ww[kk] = (CW)small[2 * NN + (k0 + kk) * NT + j];
		cc[kk] = (CW)small[4 * NN + (k0 + kk) * NT + j];
		vd[kk] = (CW)small[5 * NN + (k0 + kk) * NT + j];
	}
	const bool dj = small[3 * NN + j] != 0;
	const long long w0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((long long)gridDim.x * BLOCK) >> 6;
	for (long long r = w0; r < rows; r += nw) {
		const size_t row = (size_t)r * NT;
		const u64 bv = V[row + k0 + (lane & 15)], bp = P[row + k0 + (lane & 15)];	/* to broadcast */
		u64 vj = 0, aj = 0, pj = 0;							/* own column: half 0 adds it */
		if (h == 0) {
			vj = V[row + j];
			aj = AV[row + j];
			pj = P[row + j];
		}
		typename std::conditional<NARROW, AccS, AccL>::type av, ap;
		acc_set(av, h ? 0 : (dj ? aj : vj));
		acc_set(ap, (h || dj) ? 0 : pj);
		static_for<0, HK>([&](auto kc) {
			constexpr int kk = decltype(kc)::value;
			const u64 vk = group_bcast<kk, 16, NARROW>(bv), pk = group_bcast<kk, 16, NARROW>(bp);
			acc_mac64(av, vk, cc[kk]);
			acc_mac64(av, pk, vd[kk]);
			acc_mac64(ap, vk, ww[kk]);
		});
		u64 rv = acc_reduce<MERS>(av, m), rp = acc_reduce<MERS>(ap, m);
		rv = addmod(rv, shfl_xor64(rv, 32), m.p);
		rp = addmod(rp, shfl_xor64(rp, 32), m.p);
		if (h == 0) {
			V[row + j] = (W)rv;
			P[row + j] = (W)rp;
		}
	}
}

/*
 * n = 64: one workgroup per block row; wavefront q takes the quarter k in [16q, 16q + 16) of the sum for all 64 output
 * columns (3 * 16 coefficient words per lane, operands by row_newbcast as at n = 32), the four partial residues are
 * added through LDS and wavefront 0 stores the row.
 */
template <typename W, int MERS>
__global__ void __launch_bounds__(BLOCK)
k_orthogonalize_64(W *__restrict__ V, const W *__restrict__ AV, W *__restrict__ P, long long rows, ModP m,
		   const u64 *__restrict__ small, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	constexpr int NT = 64, NN = NT * NT, HK = 16;
	constexpr bool NARROW = sizeof(W) == 4;
	__shared__ u64 red[2][2][4][NT];
	const int j = threadIdx.x & 63, q4 = threadIdx.x >> 6, k0 = HK * q4;
	using CW = typename std::conditional<NARROW, u32, u64>::type;
	CW cc[HK], vd[HK], ww[HK];
#pragma unroll
	for (int kk = 0; kk < HK; kk++) {
		ww[kk] = (CW)small[2 * NN + (k0 + kk) * NT + j];
		cc[kk] = (CW)small[4 * NN + (k0 + kk) * NT + j];
		vd[kk] = (CW)small[5 * NN + (k0 + kk) * NT + j];
	}
	const bool dj = small[3 * NN + j] != 0;
	int par = 0;
	for (long long r = blockIdx.x; r < rows; r += gridDim.x, par ^= 1) {
		const size_t row = (size_t)r * NT;
		const u64 bv = V[row + k0 + (j & 15)], bp = P[row + k0 + (j & 15)];
		u64 vj = 0, aj = 0, pj = 0;
		if (q4 == 0) {
			vj = V[row + j];
			aj = AV[row + j];
			pj = P[row + j];
		}
		typename std::conditional<NARROW, AccS, AccL>::type av, ap;
		acc_set(av, q4 ? 0 : (dj ? aj : vj));
		acc_set(ap, (q4 || dj) ? 0 : pj);
		static_for<0, HK>([&](auto kc) {
			constexpr int kk = decltype(kc)::value;
			const u64 vk = group_bcast<kk, 16, NARROW>(bv), pk = group_bcast<kk, 16, NARROW>(bp);
			acc_mac64(av, vk, cc[kk]);
			acc_mac64(av, pk, vd[kk]);
			acc_mac64(ap, vk, ww[kk]);
		});
		red[par][0][q4][j] = acc_reduce<MERS>(av, m);
		red[par][1][q4][j] = acc_reduce<MERS>(ap, m);
		__syncthreads();	/* also: every wavefront has read row r before wavefront 0 overwrites it */
		if (q4 == 0) {
			u64 rv = red[par][0][0][j], rp = red[par][1][0][j];
#pragma unroll
			for (int w = 1; w < 4; w++) {
				rv = addmod(rv, red[par][0][w][j], m.p);
				rp = addmod(rp, red[par][1][w][j], m.p);
			}
			V[row + j] = (W)rv;
			P[row + j] = (W)rp;
		}
	}
}

template <typename W, int MERS>
static hipError_t ortho_dispatch(const KernelCfg &c, W *V, const W *AV, W *P, int64_t rows, const u64 *small,
				 const DevCtl *ctl, hipStream_t s)
{
	if (rows == 0)
		return hipSuccess;
	const int n = c.n;
	const long long cap = (long long)c.num_cu * 8;
	if (n == 64 && c.m.chunk >= 32u) {
		long long blocks = rows;
		const long long fit = (long long)c.num_cu * (sizeof(W) == 4 ? 6 : 3);
		if (blocks > fit)
			blocks = fit;
		hipLaunchKernelGGL((k_orthogonalize_64<W, MERS>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, P,
				   (long long)rows, c.m, small, ctl);
		return hipGetLastError();
	}
	if (n == 32 && c.m.chunk >= 32u) {	/* 2 * 16 products per half-sum */
		long long blocks = (rows + 3) / 4;
		const long long fit = (long long)c.num_cu * (sizeof(W) == 4 ? 6 : 3);	/* what the registers let be resident */
		if (blocks > fit)
			blocks = fit;
		hipLaunchKernelGGL((k_orthogonalize_32<W, MERS>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, P,
				   (long long)rows, c.m, small, ctl);
		return hipGetLastError();
	}
	if ((n == 1 || n == 2 || n == 4 || n == 8 || n == 16) && c.m.chunk >= 2u * (unsigned)n) {
		const long long gpb = BLOCK / n;
		long long blocks = (rows + gpb - 1) / gpb;
		/* persistent grid = what is resident at once (registers allow 5 / 3 blocks of 256 per CU at n = 8 / 16):
		 * a grid of 8 per CU would run as 1.6 "rounds" and end on a half-empty chip */
		const long long fit = (long long)c.num_cu * (n >= 16 ? 3 : (n >= 8 ? 5 : 8));
		if (blocks > fit)
			blocks = fit;
		if (blocks > cap)
			blocks = cap;
#define ORTHO_FAST(NN)                                                                                               \
	case NN:                                                                                                     \
		hipLaunchKernelGGL((k_orthogonalize_fast<W, MERS, NN>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, V, AV, P, \
				   (long long)rows, c.m, small, ctl);                                                 \
		break;
		switch (n) {
			ORTHO_FAST(1)
			ORTHO_FAST(2)
			ORTHO_FAST(4)
			ORTHO_FAST(8)
			ORTHO_FAST(16)
		}
#undef ORTHO_FAST
		return hipGetLastError();
	}
	int G = 1;
	while (G < n)
		G <<= 1;
	const int gpb = BLOCK / G;
	const size_t lds = ((size_t)3 * n * n + (size_t)2 * gpb * n) * sizeof(u64);
	if (lds > 48 * 1024)
		hipFuncSetAttribute((const void *)k_orthogonalize<W, MERS>, hipFuncAttributeMaxDynamicSharedMemorySize,
				    (int)lds);
	long long blocks = (rows + gpb - 1) / gpb;
	if (blocks > cap)
		blocks = cap;
	hipLaunchKernelGGL((k_orthogonalize<W, MERS>), dim3((unsigned)blocks), dim3(BLOCK), lds, s, V, AV, P,
			   (long long)rows, n, G, c.m, small, ctl);
	return hipGetLastError();
}

bool ortho_uses_mfma(const KernelCfg &c, int64_t rows)
{
	return ortho_mfma_supported(c) && rows >= c.mfma_min_rows && rows > 0;
}

hipError_t launch_orthogonalize(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows,
				const u64 *small, const DevCtl *ctl, hipStream_t s, bool img_ready)
{
	if (ortho_uses_mfma(c, rows))
		return launch_orthogonalize_mfma(c, V, AV, P, rows, small, ctl, s, img_ready);
	if (c.word == 4)
		return c.mers == 31 ? ortho_dispatch<u32, 31>(c, (u32 *)V, (const u32 *)AV, (u32 *)P, rows, small, ctl, s)
				    : ortho_dispatch<u32, 0>(c, (u32 *)V, (const u32 *)AV, (u32 *)P, rows, small, ctl, s);
	return c.mers == 61 ? ortho_dispatch<u64, 61>(c, (u64 *)V, (const u64 *)AV, (u64 *)P, rows, small, ctl, s)
			    : ortho_dispatch<u64, 0>(c, (u64 *)V, (const u64 *)AV, (u64 *)P, rows, small, ctl, s);
}

/* ---------------------------------------------------------------------------- utilities */

template <typename W>
__global__ void __launch_bounds__(BLOCK) k_any_nonzero(const W *__restrict__ X, long long words, int *flag)
{
	bool any = false;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < words; k += (long long)gridDim.x * BLOCK)
		any |= (X[k] != 0);
	if (__any(any) && (threadIdx.x & 63) == 0)
		atomicOr(flag, 1);
}

/* the control words, stored to host-mapped pinned memory by the GPU itself: a hipMemcpyAsync of these 24 bytes queues
 * behind whatever the copy engines are busy with (a checkpoint's 3 GB device-to-host transfer held blz_iterate's return
 * back by ~100 ms) */
__global__ void k_publish_ctl(const DevCtl *__restrict__ ctl, DevCtl *__restrict__ host_mapped)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
		*host_mapped = *ctl;
}

hipError_t launch_publish_ctl(const DevCtl *ctl, DevCtl *host_mapped, hipStream_t s)
{
	hipLaunchKernelGGL(k_publish_ctl, dim3(1), dim3(64), 0, s, ctl, host_mapped);
	return hipGetLastError();
}

/* dst <- src, 16 bytes per lane (the snapshot of v and p: hipMemcpyAsync device-to-device ran at ~60 GB/s here) */
__global__ void __launch_bounds__(BLOCK)
k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long long n16)
{
	const long long step = (long long)gridDim.x * BLOCK;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += step)
		dst[i] = src[i];
}

hipError_t launch_copy(const KernelCfg &c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
	const long long n16 = (long long)(bytes / 16);
	if (n16 > 0) {
		long long blocks = (n16 + BLOCK * 8 - 1) / (BLOCK * 8);
		blocks = blocks > (long long)c.num_cu * 8 ? (long long)c.num_cu * 8 : blocks;
		hipLaunchKernelGGL(k_copy16, dim3((unsigned)blocks), dim3(BLOCK), 0, s, (uint4 *)dst, (const uint4 *)src, n16);
	}
	if (bytes % 16)		/* slabs are whole block rows of 4- or 8-byte words: a tail of 4, 8 or 12 bytes at most */
		return hipMemcpyAsync((char *)dst + n16 * 16, (const char *)src + n16 * 16, bytes % 16, hipMemcpyDeviceToDevice, s);
	return hipGetLastError();
}

/* x <- x mod p, words that are sums of a few residues (the reduce-scatter of partial products) */
/* Out of place, and a no-op after the stop: the collective that fills `src` is enqueued by the host whatever the flag
 * says, and past the stop it sums stale partial products -- dst must keep the last real iteration's residues (a batch
 * always runs past the stop: blz_final_check reads TMP). */
template <int MERS>
__global__ void __launch_bounds__(BLOCK)
k_reduce_modp(u64 *__restrict__ dst, const u64 *__restrict__ src, long long words, ModP m, const DevCtl *__restrict__ ctl)
{
	if (ctl->stop)
		return;
	const long long step = (long long)gridDim.x * BLOCK;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < words; i += step)
		dst[i] = reduce128<MERS>(0, src[i], m);
}

hipError_t launch_reduce_modp(const KernelCfg &c, void *X, const void *S, int64_t words, const DevCtl *ctl, hipStream_t s)
{
	if (words <= 0)		/* (64-bit words whatever the residue width: the callers' buffers hold u64 sums) */
		return hipSuccess;
	long long blocks = (words + BLOCK * 4 - 1) / (BLOCK * 4);
	blocks = blocks < 1 ? 1 : (blocks > (long long)c.num_cu * 8 ? (long long)c.num_cu * 8 : blocks);
	if (c.mers == 61)
		hipLaunchKernelGGL((k_reduce_modp<61>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (u64 *)X, (const u64 *)S, (long long)words, c.m, ctl);
	else
		hipLaunchKernelGGL((k_reduce_modp<0>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (u64 *)X, (const u64 *)S, (long long)words, c.m, ctl);
	return hipGetLastError();
}

struct SumSources {
	const void *p[BLZ_LOOP_MAX_RANKS];
};

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_sum_buffers(SumSources src, int nsrc, T *__restrict__ dst, long long words)
{
	const long long step = (long long)gridDim.x * BLOCK;
	for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < words; i += step) {
		T s = 0;
		for (int q = 0; q < nsrc; q++)
			s += ((const T *)src.p[q])[i];
		dst[i] = s;
	}
}

hipError_t launch_sum_buffers(const void *const *src, int nsrc, void *dst, long long words, int word_bytes, hipStream_t s)
{
	if (nsrc < 1 || nsrc > BLZ_LOOP_MAX_RANKS || words < 0)
		return hipErrorInvalidValue;
	if (words == 0)
		return hipSuccess;
	SumSources ss;
	for (int q = 0; q < BLZ_LOOP_MAX_RANKS; q++)
		ss.p[q] = q < nsrc ? src[q] : nullptr;
	long long blocks = (words + BLOCK - 1) / BLOCK;
	blocks = blocks > 2048 ? 2048 : blocks;
	if (word_bytes == 8)
		hipLaunchKernelGGL((k_sum_buffers<u64>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, ss, nsrc, (u64 *)dst, words);
	else
		hipLaunchKernelGGL((k_sum_buffers<u32>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, ss, nsrc, (u32 *)dst, words);
	return hipGetLastError();
}

hipError_t launch_any_nonzero(const KernelCfg &c, const void *X, int64_t words, int *flag, hipStream_t s)
{
	if (words == 0)
		return hipSuccess;
	long long blocks = (words + BLOCK - 1) / BLOCK;
	if (blocks > (long long)c.num_cu * 8)
		blocks = (long long)c.num_cu * 8;
	if (c.word == 4)
		hipLaunchKernelGGL((k_any_nonzero<u32>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (const u32 *)X,
				   (long long)words, flag);
	else
		hipLaunchKernelGGL((k_any_nonzero<u64>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, (const u64 *)X,
				   (long long)words, flag);
	return hipGetLastError();
}
