/* blz_kernels.h -- launchers of the gfx950 kernels (C++ side only; not part of the C ABI). */
#ifndef BLZ_KERNELS_H
#define BLZ_KERNELS_H

#include <hip/hip_runtime.h>
#include "modp.h"

/* Outlier rows of a slab (more than heavy_thr entries), cut into segments of at most HEAVY_SEG entries. */
#define HEAVY_SEG 4096u
struct HeavySeg {
	int row;
	u32 k0, k1;		/* entries [k0, k1) of col_idx / val */
	int whole_row;		/* the row is this one segment: k_spmv_heavy finishes it */
};
struct HeavyRow {
	int row, first, count;	/* a split row: segments [first, first + count) of the list */
};

/* rows [begin[x], begin[x+1]) belong to XCD x (blocks b and b + 8 share an XCD); begin[0] < 0: no ranges */
struct XcdRows {
	long long begin[9];
};

/* One CSR slab resident in HBM. */
struct DevCsr {
	int64_t rows = 0, cols = 0, nnz = 0;
	u32 *row_ptr = nullptr;
	int *col_idx = nullptr;
	u32 *val = nullptr;	/* nullptr: all ones, or packed (palette != nullptr) */
	u32 *palette = nullptr;	/* 256 values: col_idx then holds  column | (palette index << 24)  */
	HeavySeg *heavy = nullptr;	/* segments of the rows longer than heavy_thr: handled by k_spmv_heavy */
	int n_heavy = 0;
	HeavyRow *heavy_multi = nullptr;	/* the rows among them that span several segments */
	int n_multi = 0;
	u64 *heavy_scratch = nullptr;	/* 128-bit partial sums: [segment][G lanes][lo, hi] */
	int *medium_rows = nullptr;	/* rows above heavy_thr that one wavefront can take (k_spmv_wave) */
	int n_medium = 0;
	double kept_mean = -1.0;	/* mean length of the rows the streaming kernel keeps (< 0: nnz / rows) */
	u32 heavy_thr = 0xFFFFFFFFu;
	bool uneven = false;		/* row lengths vary a lot (std > mean/2): the SpMV wants more resident waves */
	bool tail_batch = false;	/* k_spmv / k_spmv_dot take a row's 1-3 left-over entries as one predicated batch (gathers that hit, or a
					 * product that is a few launches' worth of latency) */
	double outlier_share = 0.0;	/* share of the entries in rows above heavy_thr (the outlier launches take them) */
	double locality = 1.0;		/* lines of the operand per gathered entry that the renumbering's sample found for this product (1: no reuse) */
	/* plan of the LDS-staged streaming kernel (k_spmv_staged), made at upload by spmv_plan_staged() */
	bool st_ok = false;		/* false: the slab runs k_spmv / k_spmv_dot */
	int st_rpg = 0;			/* lockstep form: rows per lane group per tile; a tile is st_tr = (64/G) * st_rpg consecutive rows */
	int st_tr = 0;			/* rows per tile */
	bool st_pair = false;		/* n = 16, 64-bit words: 8 lanes of two words per row (16-byte gathers), 8 rows per wavefront */
	bool st_dyn = false;		/* the lane groups take the tile's rows from a shared counter (k_spmv_staged<..., DYN>) */
	int st_capw = 0;		/* staging window per wavefront and buffer, entries */
	int st_per_cu = 0;		/* resident workgroups per CU the grid is sized for */
	int st_ns = 1;			/* streams staged: col_idx (+ val when it is a separate array) */
	/* LDS-resident panel of the operand's first panel_rows block rows (k_spmv_panel), set by spmv_plan_panel() */
	int panel_rows = 0;
	bool xcd_ranges = false;	/* k_spmv / k_spmv_dot walk per-XCD row ranges (set when the renumbering found locality) */
	long long xr_rows[9] = { 0 };	/* XCD x takes rows [xr_rows[x], xr_rows[x+1]); xr_rows[0] < 0: equal stripes */
	int st_deep = 1;		/* 8 gathers in flight per lane where the kernel has that form (else 4): off for rows of a few entries */
	int st_interleave = 0;		/* 1: tiles round-robin over the whole grid instead of per-XCD ranges */
	long long st_tiles[9] = { 0 };	/* XCD x takes tiles [st_tiles[x], st_tiles[x+1]): contiguous, nnz-balanced */
};

/* col_idx / val of a slab are allocated with this many spare entries: the staged kernel copies whole 16-byte chunks */
#define BLZ_STREAM_PAD 16

/* Control words shared by all kernels of a context (device memory). */
struct DevCtl {
	int stop;		/* set by semi_inverse when npiv == 0: every later kernel is a no-op */
	int npiv;
	long long iterations;	/* the reference's n_iterations */
	int flag_v_nonzero, flag_t_nonzero;
};

struct KernelCfg {
	int n;			/* block width */
	int word;		/* 4 or 8 */
	int mers;		/* 0, 31, 61 */
	ModP m;
	int num_cu;
	int spmv_blocks_per_cu;	/* grid of the persistent SpMV = num_cu * this (BLZ_SPMV_BLOCKS_PER_CU overrides) */
	hipStream_t side;	/* stream of the outlier-row launches (nullptr: same stream as the streaming kernel; BLZ_NO_SIDE=1) */
	hipEvent_t ev_fork, ev_join;
	int mfma;		/* 1: the dense row kernels use the matrix cores where they can (p = 2^61-1, n = 8 / 16); BLZ_NO_MFMA=1 */
	int mfma_stage8;		/* 1: the n = 8 block update stages its rows through LDS like the n = 16 one (A/B; BLZ_MFMA_STAGE8=1) */
	long long mfma_min_rows;	/* block update: below this many rows the vector-ALU kernel is quicker (launch + image set-up); BLZ_MFMA_MIN_ROWS */
	void *mfma_img;		/* device scratch for the coefficient digits in MFMA fragment order (ortho_mfma_image_bytes()) */
	int panel;		/* 1: slabs whose operand has hot block rows run k_spmv_panel; BLZ_NO_PANEL=1 turns it off */
	int staged;		/* 1: slabs with a plan run k_spmv_staged; BLZ_NO_STAGE=1 keeps the round-1 kernels (A/B) */
	int pair;		/* 1: two words per lane (16-byte gathers) in the staged SpMV at n = 16 and n = 8, 64-bit words (BLZ_NO_PAIR=1: off) */
	int stage_dyn;		/* -1: dynamic rows in the staged SpMV by plan; 0 / 1: forced off / on (BLZ_STAGE_DYN, read once) */
	int stage_u;		/* 0: gathers in flight per lane of the staged SpMV chosen by the slab's plan; 4 / 8: forced (BLZ_STAGE_U, read once) */
};

/* fills the st_* fields of D from the host copy of its row pointers (D.rows, D.nnz, D.val, D.palette, D.kept_mean,
 * D.uneven must be set) */
void spmv_plan_staged(const KernelCfg &c, const u32 *row_ptr, DevCsr &D, bool allow_dyn);	/* allow_dyn: the slab never runs the fused (DOT) form */

/* how many block rows the LDS panel of k_spmv_panel can hold for this context (0: the form is not available) */
int64_t spmv_panel_capacity(const KernelCfg &c);
/* the slab's operand has its `hot_rows` densest block rows numbered first: plan k_spmv_panel (D.rows, D.nnz, D.cols set) */
void spmv_plan_panel(const KernelCfg &c, const u32 *row_ptr, DevCsr &D, int64_t hot_rows);

/* rows of a slab with more entries than this get a workgroup each (DevCsr::heavy) */
u32 spmv_heavy_threshold(const KernelCfg &c, int64_t rows, int64_t nnz);

/* Y[rows x n] = A * X, X addressed through A.col_idx (row-major, n words per row).
 * sequential/lanczos_modp.c:266-287 */
hipError_t launch_spmv(const KernelCfg &c, const DevCsr &A, const void *X, void *Y, int accum, const DevCtl *ctl,
		       hipStream_t s);	/* accum != 0: Y = (Y + A*X) mod p (one piece of a column-chunked product) */

/* Y = A*X with block_dot_products(V_slab, Y) as the epilogue (n in {1,2,4,8} only): one partial row per block. */
bool spmv_dot_supported(const KernelCfg &c);
hipError_t launch_spmv_dot(const KernelCfg &c, const DevCsr &A, const void *X, void *Y, const void *Vd, int accum,
			   u64 *partial, int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s);

/* partial[b][0..n*n) = sum over block b's rows of v^T Av, partial[b][n*n..2n*n) = Av^T Av.
 * sequential/lanczos_modp.c:443-453.  Returns the number of partial rows written via *nblocks. */
hipError_t launch_block_dot(const KernelCfg &c, const void *V, const void *AV, int64_t rows, u64 *partial,
			    int max_blocks, int *nblocks, const DevCtl *ctl, hipStream_t s);
/* out[0..2*n*n) = sum_b partial[b][.] mod p */
hipError_t launch_dot_finalize(const KernelCfg &c, const u64 *partial, int nblocks, u64 *out, const DevCtl *ctl,
			       hipStream_t s);

/* small: [vtAv | vtAAv | winv | d | c | vtAvd], n*n words each (d: n words, padded to n*n).
 * Reads vtAv/vtAAv (reduced mod p first: they may be sums over ranks), writes the rest and
 * updates ctl.  sequential/lanczos_modp.c:342-438 and :460-475. */
hipError_t launch_semi_inverse(const KernelCfg &c, const u64 *sums, u64 *small, DevCtl *ctl, int in_loop, int build_img,
			       hipStream_t s);
/* `sums` = the 2 n^2 words vtAv | vtAAv the kernel starts from: `small` itself on one rank, the landing place of the
 * all-reduce on several (sums of the ranks' residues; the kernel writes their residues to `small` unless the stop flag
 * is up, so `small` never holds anything but residues).  build_img != 0: the kernel also writes the coefficient image
 * of the matrix-core block update (c.mfma_img; ortho_mfma_supported(c) must hold): one launch instead of two. */

/* Row-local update in place: V <- v', P <- p'.  sequential/lanczos_modp.c:478-491, :655-656 */
hipError_t launch_orthogonalize(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows,
				const u64 *small, const DevCtl *ctl, hipStream_t s, bool img_ready = false);
/* true when launch_orthogonalize(rows) runs on the matrix cores, i.e. reads the coefficient image */
bool ortho_uses_mfma(const KernelCfg &c, int64_t rows);

/* *host_mapped <- *ctl, written by the GPU into host-mapped pinned memory (no copy engine involved) */
hipError_t launch_publish_ctl(const DevCtl *ctl, DevCtl *host_mapped, hipStream_t s);

/* device-to-device copy by a streaming kernel (HBM speed) */
hipError_t launch_copy(const KernelCfg &c, void *dst, const void *src, size_t bytes, hipStream_t s);

/* dst[i] <- src[i] mod p for 64-bit words that hold sums of a few residues (the landing place of a reduce-scatter);
 * a no-op once the stop flag is up, so dst keeps the residues of the last real iteration */
hipError_t launch_reduce_modp(const KernelCfg &c, void *dst, const void *src, int64_t words, const DevCtl *ctl, hipStream_t s);

/* the same update on v_mfma_i32_16x16x64_i8 (blz_dense_mfma.hip): exact integer contractions of base-256 digits */
size_t ortho_mfma_image_bytes(void);
bool ortho_mfma_supported(const KernelCfg &c);
hipError_t launch_orthogonalize_mfma(const KernelCfg &c, void *V, const void *AV, void *P, int64_t rows, const u64 *small,
				     const DevCtl *ctl, hipStream_t s, bool img_ready);

/* block_dot_products on the matrix cores (p = 2^61-1, n = 8 / 16): same partial rows as launch_block_dot */
bool block_dot_mfma_supported(const KernelCfg &c);
hipError_t launch_block_dot_mfma(const KernelCfg &c, const void *V, const void *AV, int64_t rows, u64 *partial, int max_blocks,
				 int *nblocks, const DevCtl *ctl, hipStream_t s);

/* dst[i] = sum over the nsrc buffers of src[q][i] (plain wrap-around sums of 8- or 4-byte words): the data movement of the
 * loopback communicator's all-reduce / reduce-scatter (blz_api.hip) */
#define BLZ_LOOP_MAX_RANKS 16
hipError_t launch_sum_buffers(const void *const *src, int nsrc, void *dst, long long words, int word_bytes, hipStream_t s);

/* flag |= any(X != 0) over `words` words */
hipError_t launch_any_nonzero(const KernelCfg &c, const void *X, int64_t words, int *flag, hipStream_t s);

static inline size_t small_words(int n) { return (size_t)6 * n * n; }

#endif
