/*
 * blz_api.hip -- the device half of the C ABI declared in include/blz.h: contexts, HBM residency,
 * kernel sequencing, HIP-event timing and the RCCL exchange steps.
 *
 * HBM layout per context (one per GPU).  "side 0" = rows of v/Av/p, "side 1" = rows of tmp.
 *   slab[V], slab[AV], slab[P] : this rank's rows of side 0, slab[TMP] : its rows of side 1, row-major
 *   rows x n, padded to `stride` rows.  With one rank these ARE the reference's N x n arrays.
 *   gath[side] (only with nranks > 1): the all-gathered operand of a product, K = ag_chunks pieces:
 *   [piece k][rank g][stride/K rows][n] -- ncclAllGather number k moves piece k of every slab and lands
 *   contiguously, so the product on csr[t][k] (the entries whose columns lie in piece k) can run on the compute
 *   stream while piece k+1 is still in flight on the exchange stream.
 *   csr[0][k] / csr[1][k] = this rank's rows of M / M^T restricted to the columns of piece k.
 *   small  = [vtAv | vtAAv | winv | d | c | vtAvd], 6*n*n words;  ctl = DevCtl.
 */
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "blz_internal.h"
#include "blz_kernels.h"

#define HIPCHK(expr)                                                                                     \
	do {                                                                                             \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess)                                                                    \
			return blz_fail(BLZ_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

namespace {

struct Rccl {
	void *handle = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
	ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
	ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

/* RCCL is resolved at run time and only when a communicator is asked for: a single-GPU solve has
 * no dependency on it. */
int rccl_load()
{
	/* the CLI's --gpus N has one thread per GPU, each asking for its communicator: one of them loads, the others wait */
	static std::mutex load_mu;
	std::lock_guard<std::mutex> lk(load_mu);
	if (g_rccl.handle)
		return BLZ_OK;
	/* Prefer the RCCL that belongs to the ROCm installation this library was built against (the HIP runtime the
	 * process uses is /opt/rocm's when libblz_hip.so is loaded first); BLZ_RCCL_PATH overrides; a copy already
	 * mapped under the plain soname (e.g. torch's) is the last resort. */
	const char *names[] = { getenv("BLZ_RCCL_PATH"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so" };
	void *h = nullptr;
	for (const char *nm : names)
		if (nm && nm[0] && (h = dlopen(nm, RTLD_NOW | RTLD_LOCAL)))
			break;
	if (!h)
		return blz_fail(BLZ_ECOMM, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                                   \
	*(void **)(&g_rccl.field) = dlsym(h, name);                                        \
	if (!g_rccl.field)                                                                 \
		return blz_fail(BLZ_ECOMM, "librccl lacks %s", name);
	SYM(GetUniqueId, "ncclGetUniqueId")
	SYM(CommInitRank, "ncclCommInitRank")
	SYM(CommDestroy, "ncclCommDestroy")
	SYM(AllGather, "ncclAllGather")
	SYM(AllReduce, "ncclAllReduce")
	SYM(ReduceScatter, "ncclReduceScatter")
	SYM(CommCount, "ncclCommCount")
	SYM(CommUserRank, "ncclCommUserRank")
	SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
	g_rccl.handle = h;
	return BLZ_OK;
}

#define NCCLCHK(expr)                                                                                     \
	do {                                                                                              \
		ncclResult_t r_ = (expr);                                                                 \
		if (r_ != ncclSuccess)                                                                    \
			return blz_fail(BLZ_ECOMM, "%s: %s", #expr, g_rccl.GetErrorString(r_));           \
	} while (0)

}  // namespace

enum { PK_SPMV1 = 0, PK_SPMV2, PK_DOT, PK_SEMI, PK_ORTHO, PK_AG_V, PK_AG_T, PK_AR, PK_RS, PK_COUNT };

/*
 * Loopback communicator (round 3): several contexts of ONE process on ONE device, one host thread each, exchange through
 * this object instead of RCCL (which refuses two ranks on one GPU).  Everything else of the multi-rank path -- the slabs,
 * the gathered layouts, the piece pipeline on two streams with its events, the landing buffers, what a batch does past
 * the stop -- is the production code, so the one-GPU boxes this was developed on can run blz_iterate with 2, 3 or 8 real
 * ranks' worth of sums against the oracle.  A collective is: post my buffers and an event on my stream; meet the other
 * threads; make my stream wait for their events and enqueue my copies / sums from THEIR send buffers; record a second
 * event; meet again; make my stream wait for everybody's second event (nobody overwrites a buffer somebody still reads).
 * Not a transport: the data never leaves the device.
 */
struct blz_loop_group {
	int nranks = 0;
	std::mutex mu;
	std::condition_variable cv;
	int arrived = 0;
	unsigned long long generation = 0;
	bool broken = false;
	int timeout_s = 120;		/* BLZ_LOOP_TIMEOUT_S at creation (tests of the time-out itself) */
	std::vector<const void *> send;
	/* one pair of events per rank, OWNED BY THE GROUP and destroyed with it: a rank that is done may destroy its context
	 * while a peer's stream still holds a wait on its event that has not executed yet */
	std::vector<hipEvent_t> ready, done;
	/* all ranks meet; false after the time-out or once any rank has given up */
	bool meet()
	{
		std::unique_lock<std::mutex> lk(mu);
		if (broken)
			return false;
		const unsigned long long gen = generation;
		if (++arrived == nranks) {
			arrived = 0;
			generation++;
			cv.notify_all();
			return true;
		}
		if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != gen || broken; }))
			broken = true;
		if (broken) {
			cv.notify_all();
			return false;
		}
		return true;
	}
};

struct ProfSpan {
	int cls;
	hipEvent_t a, b;
};

struct blz_ctx {
	bool profiling = false;
	std::vector<ProfSpan> spans;
	std::vector<hipEvent_t> pool;
	int device = 0;
	KernelCfg cfg{};
	u64 prime = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool have_matrix = false;
	int right = 0, rank = 0, nranks = 1;
	int64_t glob_rows[2] = { 0, 0 };		/* side 0: N, side 1: C */
	int64_t first[2] = { 0, 0 }, count[2] = { 0, 0 }, stride[2] = { 0, 0 };
	std::vector<int64_t> bounds[2];
	std::vector<DevCsr> csr[2];			/* column pieces of this rank's rows of M / M^T */
	int row_side[2] = { 0, 1 };			/* side of the rows of csr[t] */
	void *slab[4] = { nullptr, nullptr, nullptr, nullptr };
	size_t slab_bytes[4] = { 0, 0, 0, 0 };	/* several ranks: each block sized by its own side (V, AV, P: rows of v; TMP: the other side) */
	void *gath[2] = { nullptr, nullptr };		/* gathered operands (nranks > 1) */
	int gath_holds[2] = { -1, -1 };			/* which block each one currently holds */
	int ag_chunks = 0;				/* pieces per all-gather: BLZ_AG_CHUNKS, 0 = choose from the slab size */
	hipStream_t xstream = nullptr;			/* exchange stream (all-gathers run beside the products) */
	hipEvent_t ev_prod = nullptr;			/* the producer of the block to exchange has been enqueued */
	std::vector<hipEvent_t> ev_piece;		/* piece k of the exchange has landed */
	u64 *small = nullptr, *partial = nullptr;
	u64 *dot_send = nullptr;	/* this rank's vtAv | vtAAv before the all-reduce (several ranks): 2 n^2 words */
	u64 *dot_recv = nullptr;	/* where the all-reduce lands: sums of the ranks' residues; the semi-inverse kernel reads them
					 * and writes residues to `small` -- unless the stop flag is up (round 3) */
	int max_dot_blocks = 0;
	DevCtl *ctl = nullptr;
	DevCtl host_ctl{};
	DevCtl *ctl_pinned = nullptr, *ctl_pinned_dev = nullptr;	/* host-mapped landing place of the control words */
	ncclComm_t comm = nullptr;
	blz_loop_group *loop = nullptr;		/* loopback communicator instead of RCCL (contexts of one process on one device) */
	int loop_rank = -1;			/* this context's rank in it (the matrix must be set with the same rank and rank count) */
	/* perm[side][original row] = row in the solver's numbering (empty = identity); inv is the inverse */
	std::vector<int32_t> perm[2], inv[2];
	bool reorder = true;		/* BLZ_NO_REORDER=1 keeps the file's numbering */
	bool pack = true;		/* BLZ_NO_PACK=1 keeps col_idx and val as two arrays */
	bool fuse_dot = true;		/* BLZ_NO_FUSE=1 keeps block_dot as its own kernel (A/B measurements) */
	bool fuse_local_off = false;	/* this matrix: the second product runs the staged form and block_dot its own kernel (gathers that hit) */
	int un = 0;			/* the caller's block width; cfg.n is the width in HBM (below) */
	bool use_graph = false;		/* BLZ_GRAPH=1: single-GPU iterations are replayed from a captured hipGraph */
	hipGraphExec_t iter_graph = nullptr;
	bool external_exchange = false;
	bool force_comm = false;	/* BLZ_FORCE_COMM=1: issue the collectives even on one rank (plumbing test) */
	/* asynchronous snapshot of v and p (checkpoints): pinned staging, its own stream, one in flight */
	hipStream_t cstream = nullptr;
	hipEvent_t ev_snap_go = nullptr, ev_snap_done = nullptr;
	void *snap_host[2] = { nullptr, nullptr };
	void *snap_dev[2] = { nullptr, nullptr };	/* device-side copy the D2H reads from while the loop goes on (nullptr: no room) */
	size_t snap_bytes = 0;
	std::atomic<bool> snap_pending{ false };	/* set by the owner (begin), cleared by whoever collects (wait: possibly another thread) */
	int64_t snap_iterations = 0;
	/* short-side exchange (tall / wide matrices on several ranks): product t multiplies the transpose of this rank's rows
	 * of the other orientation by its OWN slab and reduce-scatters the full-length partial products */
	bool short_side[2] = { false, false };
	DevCsr csr_short[2];
	void *part = nullptr;		/* partial product, nranks x stride[output side] rows (u64 words) */
	size_t part_bytes = 0;
	void *rs_recv = nullptr;	/* where the reduce-scatter of `part` lands (stride rows); k_reduce_modp copies it into the slab
					 * as residues unless the stop flag is up (round 3) */
	size_t rs_bytes = 0;
	double hot_share[2] = { 0.0, 0.0 };	/* share of the entries held by the rows / columns numbered first */
	double locality[2] = { 1.0, 1.0 };	/* lines per gathered entry in windows of rows, product M*x / M^T*x */
	int order_kind = 0;			/* which order blz_reorder_auto chose */
};

/* HIP-event span around one enqueue on the context's stream (only while profiling is on). */
struct Span {
	blz_ctx *c;
	hipEvent_t b = nullptr;
	hipStream_t st;
	Span(blz_ctx *ctx, int cls, hipStream_t on = nullptr) : c(ctx), st(on ? on : ctx->stream)
	{
		if (!c->profiling)
			return;
		hipEvent_t ev[2];
		for (auto &e : ev) {
			if (!c->pool.empty()) {
				e = c->pool.back();
				c->pool.pop_back();
			} else if (hipEventCreate(&e) != hipSuccess) {
				return;
			}
		}
		(void)hipEventRecord(ev[0], st);
		b = ev[1];
		c->spans.push_back(ProfSpan{ cls, ev[0], ev[1] });
	}
	~Span()
	{
		if (b)
			(void)hipEventRecord(b, st);
	}
};

static inline int side_of(int block) { return block == BLZ_TMP ? 1 : 0; }

static inline char *slab_ptr(const blz_ctx *c, int block) { return (char *)c->slab[block]; }

/* the operand a product reads when its source is `block`: the slab itself on one rank, else the gathered copy */
static inline const void *operand_ptr(const blz_ctx *c, int block)
{
	return c->nranks == 1 ? c->slab[block] : c->gath[side_of(block)];
}

static void free_csr(DevCsr &A)
{
	if (A.row_ptr) hipFree(A.row_ptr);
	if (A.col_idx) hipFree(A.col_idx);
	if (A.val) hipFree(A.val);
	if (A.palette) hipFree(A.palette);
	if (A.heavy) hipFree(A.heavy);
	if (A.heavy_multi) hipFree(A.heavy_multi);
	if (A.heavy_scratch) hipFree(A.heavy_scratch);
	if (A.medium_rows) hipFree(A.medium_rows);
	A = DevCsr{};
}

extern "C" int blz_device_count(void)
{
	int cnt = 0;
	if (hipGetDeviceCount(&cnt) != hipSuccess)
		return 0;
	return cnt;
}

extern "C" int blz_create(blz_ctx **out, int device, uint64_t prime, int n)
{
	if (!out)
		return blz_fail(BLZ_EINVAL, "blz_create: out is NULL");
	*out = nullptr;
	if (n < 1 || n > BLZ_MAX_N)
		return blz_fail(BLZ_EINVAL, "n = %d is outside 1..%d", n, BLZ_MAX_N);
	if (prime < 2 || prime >= (1ull << 62))
		return blz_fail(BLZ_EINVAL, "p must satisfy 2 <= p < 2**62 (the reference's cap of 2**30 - 35 is lifted, "
				"sequential/lanczos_modp.c:189)");
	int cnt = 0;
	if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
		return blz_fail(BLZ_ENOGPU, "no HIP device visible: libblz_hip has no CPU path");
	if (device < 0 || device >= cnt)
		return blz_fail(BLZ_EINVAL, "device %d out of range (%d visible)", device, cnt);
	HIPCHK(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	blz_ctx *c = new blz_ctx();
	c->device = device;
	c->prime = prime;
	/* Blocks live in HBM with the width rounded up to a power of two, the extra columns zero: block rows are then
	 * aligned to the 128-byte lines the gathers fetch, and every width runs the specialised kernels of the next
	 * power of two.  Zero columns stay zero through every step (they are never pivots, their coefficients are zero),
	 * so the caller's n columns are the reference's.  BLZ_NO_PAD=1 keeps the exact width (generic kernels). */
	c->un = n;
	c->cfg.n = n;
	{
		const char *npad = getenv("BLZ_NO_PAD");
		if (!(npad && npad[0] == '1'))
			while (c->cfg.n & (c->cfg.n - 1))
				c->cfg.n++;
	}
	const int np_ = c->cfg.n;
	c->cfg.word = prime < (1ull << 32) ? 4 : 8;
	c->cfg.mers = modp_mersenne(prime);
	c->cfg.m = make_modp(prime);
	c->cfg.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	c->cfg.spmv_blocks_per_cu = 0;	/* 0 = choose from the row length */
	if (const char *bp = getenv("BLZ_SPMV_BLOCKS_PER_CU"))
		if (atoi(bp) >= 1 && atoi(bp) <= 64)
			c->cfg.spmv_blocks_per_cu = atoi(bp);
	{
		const char *ns = getenv("BLZ_NO_STAGE");
		c->cfg.staged = !(ns && ns[0] == '1');
		const char *su = getenv("BLZ_STAGE_U");		/* gathers in flight per lane of the staged SpMV: 4 / 8 (A/B); 0 = by plan */
		c->cfg.stage_u = su ? atoi(su) : 0;
		const char *npr = getenv("BLZ_NO_PAIR");
		c->cfg.pair = !(npr && npr[0] == '1');
		const char *sdy = getenv("BLZ_STAGE_DYN");
		c->cfg.stage_dyn = sdy ? (sdy[0] == '1' ? 1 : 0) : -1;
	}
	{
		const char *nsd = getenv("BLZ_NO_SIDE");
		c->cfg.side = nullptr;
		c->cfg.ev_fork = c->cfg.ev_join = nullptr;
		if (!(nsd && nsd[0] == '1')) {
			HIPCHK(hipStreamCreateWithFlags(&c->cfg.side, hipStreamNonBlocking));
			HIPCHK(hipEventCreateWithFlags(&c->cfg.ev_fork, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&c->cfg.ev_join, hipEventDisableTiming));
		}
	}
	{
		const char *nm = getenv("BLZ_NO_MFMA");
		c->cfg.mfma = !(nm && nm[0] == '1');
		/* measured on MI355X: fixed cost 18 us vs 9 us, slope 52 vs 71 us per million rows at n = 8 (cross at 0.5 M rows);
		 * at n = 16 the vector kernel is compute-bound at 255 us per million rows (cross near 0.1 M rows) */
		const char *s8 = getenv("BLZ_MFMA_STAGE8");
		c->cfg.mfma_stage8 = s8 && s8[0] == '1';
		const char *mr = getenv("BLZ_MFMA_MIN_ROWS");
		c->cfg.mfma_min_rows = mr ? atoll(mr) : (n == 16 ? 100000 : 500000);
		c->cfg.mfma_img = nullptr;
		HIPCHK(hipMalloc(&c->cfg.mfma_img, ortho_mfma_image_bytes()));
	}
	{
		const char *npn = getenv("BLZ_NO_PANEL");
		c->cfg.panel = !(npn && npn[0] == '1');
	}
	const char *np = getenv("BLZ_NO_PACK");
	c->pack = !(np && np[0] == '1');
	const char *nr = getenv("BLZ_NO_REORDER");
	c->reorder = !(nr && nr[0] == '1');
	const char *nf = getenv("BLZ_NO_FUSE");
	c->fuse_dot = !(nf && nf[0] == '1');
	const char *ug = getenv("BLZ_GRAPH");
	c->use_graph = ug && ug[0] == '1';
	if (const char *ac = getenv("BLZ_AG_CHUNKS"))
		if (atoi(ac) >= 1 && atoi(ac) <= 64)
			c->ag_chunks = atoi(ac);
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
	HIPCHK(hipEventCreateWithFlags(&c->ev_prod, hipEventDisableTiming));
	HIPCHK(hipEventCreate(&c->ev0));
	HIPCHK(hipEventCreate(&c->ev1));
	HIPCHK(hipMalloc(&c->small, small_words(np_) * sizeof(u64)));
	HIPCHK(hipMemset(c->small, 0, small_words(np_) * sizeof(u64)));
	HIPCHK(hipMalloc(&c->dot_send, (size_t)2 * np_ * np_ * sizeof(u64)));
	HIPCHK(hipMemset(c->dot_send, 0, (size_t)2 * np_ * np_ * sizeof(u64)));
	HIPCHK(hipMalloc(&c->dot_recv, (size_t)2 * np_ * np_ * sizeof(u64)));
	HIPCHK(hipMemset(c->dot_recv, 0, (size_t)2 * np_ * np_ * sizeof(u64)));
	/* partial rows of the inner products: the fused path (n <= 8) needs room for the streaming kernel plus the
	 * outlier launches; the stand-alone kernel keeps the grid it was tuned with */
	c->max_dot_blocks = c->cfg.num_cu * (np_ <= 8 ? 16 : 8);
	HIPCHK(hipMalloc(&c->partial, (size_t)c->max_dot_blocks * 2 * np_ * np_ * sizeof(u64)));
	HIPCHK(hipMalloc(&c->ctl, sizeof(DevCtl)));
	HIPCHK(hipHostMalloc((void **)&c->ctl_pinned, sizeof(DevCtl), hipHostMallocMapped));
	HIPCHK(hipHostGetDevicePointer((void **)&c->ctl_pinned_dev, c->ctl_pinned, 0));
	HIPCHK(hipMemset(c->ctl, 0, sizeof(DevCtl)));
	*out = c;
	return BLZ_OK;
}

extern "C" void blz_destroy(blz_ctx *c)
{
	if (!c)
		return;
	hipSetDevice(c->device);
	if (c->stream)
		hipStreamSynchronize(c->stream);
	if (c->xstream)
		hipStreamSynchronize(c->xstream);
	for (auto &sp : c->spans) {
		hipEventDestroy(sp.a);
		hipEventDestroy(sp.b);
	}
	for (auto &e : c->pool)
		hipEventDestroy(e);
	if (c->iter_graph)
		hipGraphExecDestroy(c->iter_graph);
	if (c->comm && g_rccl.CommDestroy)
		g_rccl.CommDestroy(c->comm);
	for (int t = 0; t < 2; t++)
		for (auto &A : c->csr[t])
			free_csr(A);
	for (auto &A : c->csr_short)
		free_csr(A);
	if (c->part) hipFree(c->part);
	for (void *&b : c->slab)
		if (b) hipFree(b);
	for (void *&b : c->gath)
		if (b) hipFree(b);
	for (auto &e : c->ev_piece)
		hipEventDestroy(e);
	if (c->ev_prod) hipEventDestroy(c->ev_prod);
	if (c->xstream) hipStreamDestroy(c->xstream);
	if (c->cstream) {
		hipStreamSynchronize(c->cstream);
		hipStreamDestroy(c->cstream);
	}
	if (c->ev_snap_go) hipEventDestroy(c->ev_snap_go);
	if (c->ev_snap_done) hipEventDestroy(c->ev_snap_done);
	for (void *&h : c->snap_host)
		if (h) hipHostFree(h);
	for (void *&d : c->snap_dev)
		if (d) hipFree(d);
	if (c->small) hipFree(c->small);
	if (c->dot_send) hipFree(c->dot_send);
	if (c->dot_recv) hipFree(c->dot_recv);
	if (c->rs_recv) hipFree(c->rs_recv);
	if (c->cfg.mfma_img) hipFree(c->cfg.mfma_img);
	if (c->cfg.side) {
		hipStreamSynchronize(c->cfg.side);
		hipStreamDestroy(c->cfg.side);
	}
	if (c->cfg.ev_fork) hipEventDestroy(c->cfg.ev_fork);
	if (c->cfg.ev_join) hipEventDestroy(c->cfg.ev_join);
	if (c->partial) hipFree(c->partial);
	if (c->ctl) hipFree(c->ctl);
	if (c->ctl_pinned) hipHostFree(c->ctl_pinned);
	if (c->ev0) hipEventDestroy(c->ev0);
	if (c->ev1) hipEventDestroy(c->ev1);
	if (c->stream) hipStreamDestroy(c->stream);
	delete c;
}

extern "C" int blz_word_bytes(const blz_ctx *c) { return c ? c->cfg.word : 0; }

static int upload_csr(blz_ctx *c, const blz_csr &H, DevCsr &D, int64_t hot_rows = 0, bool dot_slab = false, double locality = 1.0)
{
	free_csr(D);
	D.locality = locality;
	D.rows = H.rows;
	D.cols = H.cols;
	D.nnz = H.nnz;
	HIPCHK(hipMalloc(&D.row_ptr, (size_t)(H.rows + 1) * sizeof(u32)));
	HIPCHK(hipMemcpy(D.row_ptr, H.row_ptr, (size_t)(H.rows + 1) * sizeof(u32), hipMemcpyHostToDevice));
	HIPCHK(hipMalloc(&D.col_idx, (size_t)(H.nnz + BLZ_STREAM_PAD) * sizeof(int)));
	HIPCHK(hipMemset(D.col_idx + H.nnz, 0, BLZ_STREAM_PAD * sizeof(int)));
	/* Packed stream: when the slab has at most 256 distinct values and fewer than 2^24 columns, each entry
	 * travels as ONE u32 (column | palette index << 24) instead of two; the SpMV is bound by the number of
	 * fabric requests, and this halves those of the matrix stream.  BLZ_NO_PACK=1 keeps the plain arrays. */
	bool packed = false;
	if (H.val && c->pack && H.cols < (1 << 24)) {
		std::vector<u32> pal;
		std::vector<u32> pk((size_t)H.nnz);
		std::vector<int> slot(4096, -1);	/* open-addressing hash: value -> palette index */
		packed = true;
		for (int64_t k = 0; k < H.nnz && packed; k++) {
			const u32 v = H.val[k];
			u32 h = (v * 2654435761u) >> 20;
			while (slot[h] >= 0 && pal[(size_t)slot[h]] != v)
				h = (h + 1) & 4095u;
			if (slot[h] < 0) {
				if (pal.size() == 256) {
					packed = false;
					break;
				}
				slot[h] = (int)pal.size();
				pal.push_back(v);
			}
			pk[(size_t)k] = (u32)H.col_idx[k] | ((u32)slot[h] << 24);
		}
		if (packed) {
			pal.resize(256, 0);
			HIPCHK(hipMalloc(&D.palette, 256 * sizeof(u32)));
			HIPCHK(hipMemcpy(D.palette, pal.data(), 256 * sizeof(u32), hipMemcpyHostToDevice));
			HIPCHK(hipMemcpy(D.col_idx, pk.data(), (size_t)H.nnz * sizeof(u32), hipMemcpyHostToDevice));
		}
	}
	if (!packed) {
		HIPCHK(hipMemcpy(D.col_idx, H.col_idx, (size_t)H.nnz * sizeof(int), hipMemcpyHostToDevice));
		if (H.val) {
			HIPCHK(hipMalloc(&D.val, (size_t)(H.nnz + BLZ_STREAM_PAD) * sizeof(u32)));
			HIPCHK(hipMemset(D.val + H.nnz, 0, BLZ_STREAM_PAD * sizeof(u32)));
			HIPCHK(hipMemcpy(D.val, H.val, (size_t)H.nnz * sizeof(u32), hipMemcpyHostToDevice));
		}
	}
	D.heavy_thr = spmv_heavy_threshold(c->cfg, H.rows, H.nnz);
	std::vector<HeavySeg> heavy;
	std::vector<HeavyRow> multi;
	std::vector<int> medium;
	int G = 1;
	while (G < c->cfg.n)
		G <<= 1;
	/* up to 256 entries per lane group of the wavefront (64 batches of 4 gathers); with one group per wavefront
	 * (n > 32) the tier is empty */
	const u32 medium_max = G < 64 ? (u32)(64 / G) * 256u : 0u;
	int64_t kept_rows = 0, kept_nnz = 0;
	double sq = 0.0;
	for (int64_t r = 0; r < H.rows; r++) {
		const u32 k0 = H.row_ptr[r], len = H.row_ptr[r + 1] - k0;
		if (len <= D.heavy_thr) {
			sq += (double)len * (double)len;
			kept_rows++;
			kept_nnz += len;
			continue;
		}
		if (len <= medium_max) {	/* one wavefront per row */
			medium.push_back((int)r);
			continue;
		}
		const u32 cnt = (len + HEAVY_SEG - 1) / HEAVY_SEG, per = (len + cnt - 1) / cnt;
		if (cnt > 1)
			multi.push_back(HeavyRow{(int)r, (int)heavy.size(), (int)cnt});
		for (u32 q = 0; q < cnt; q++)
			heavy.push_back(HeavySeg{(int)r, k0 + q * per, k0 + std::min(len, (q + 1) * per), cnt == 1});
	}
	if (kept_rows > 0) {	/* spread of the rows the streaming kernel keeps */
		const double mean = (double)kept_nnz / (double)kept_rows, var = sq / (double)kept_rows - mean * mean;
		D.uneven = var > 0.25 * mean * mean;
		D.kept_mean = mean;
	}
	D.outlier_share = H.nnz > 0 ? 1.0 - (double)kept_nnz / (double)H.nnz : 0.0;
	D.tail_batch = D.locality < 0.6 || H.nnz < 4000000;
	D.n_heavy = (int)heavy.size();
	D.n_multi = (int)multi.size();
	if (D.n_heavy) {
		HIPCHK(hipMalloc(&D.heavy, heavy.size() * sizeof(HeavySeg)));
		HIPCHK(hipMemcpy(D.heavy, heavy.data(), heavy.size() * sizeof(HeavySeg), hipMemcpyHostToDevice));
	}
	D.n_medium = (int)medium.size();
	if (D.n_medium) {
		HIPCHK(hipMalloc(&D.medium_rows, medium.size() * sizeof(int)));
		HIPCHK(hipMemcpy(D.medium_rows, medium.data(), medium.size() * sizeof(int), hipMemcpyHostToDevice));
	}
	if (D.n_multi) {
		HIPCHK(hipMalloc(&D.heavy_multi, multi.size() * sizeof(HeavyRow)));
		HIPCHK(hipMemcpy(D.heavy_multi, multi.data(), multi.size() * sizeof(HeavyRow), hipMemcpyHostToDevice));
		HIPCHK(hipMalloc(&D.heavy_scratch, heavy.size() * 64 * 2 * sizeof(u64)));
	}
	spmv_plan_staged(c->cfg, H.row_ptr, D, !dot_slab);
	spmv_plan_panel(c->cfg, H.row_ptr, D, hot_rows);
	return BLZ_OK;
}

/* the parameters a context wants its matrix prepared with (pieces per exchange, renumbering, panel capacity) */
struct PrepParams {
	int K, reorder, rows_per_line;
	int64_t hot_cap;
	double min_share;
};

static PrepParams prep_params(const blz_ctx *c, int64_t mrows, int64_t mcols, int64_t nnz, int nranks)
{
	PrepParams q;
	/* Pieces per exchange.  BLZ_AG_CHUNKS fixes it; otherwise up to 4 pieces of at least ~2 MB of the smaller
	 * block's slab (below that the collectives are latency-bound and cutting them up only adds launches).
	 * A single rank only cuts its products up when the collectives are forced on (tests). */
	q.K = 1;
	if (nranks > 1 || c->force_comm) {
		if (c->ag_chunks > 0) {
			q.K = c->ag_chunks;
		} else {
			/* Cutting a product into K pieces leaves 1/K of it exposed after the last all-gather but pays K
			 * collective start-ups (and 6-30 % more product time): exposed time
			 * T/K + K*t0 is smallest at K = sqrt(T/t0), with T = this rank's product at the measured
			 * ~55 G gathers/s and t0 ~ 25 us per all-gather call.  Pieces stay above ~2 MB per slab. */
			const int64_t rows = std::min(mrows, mcols) / nranks;
			const int64_t slab_bytes = rows * c->cfg.n * c->cfg.word;
			const double t_prod_us = (double)nnz / nranks / 55e3, t0_us = 25.0;
			const int64_t by_time = (int64_t)(std::sqrt(t_prod_us / t0_us) + 0.5);
			q.K = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4, by_time), slab_bytes / (2 << 20)));
		}
	}
	const char *ao = getenv("BLZ_REORDER_PLAIN");	/* 1: round 1's order without the scored choice (A/B) */
	q.reorder = !c->reorder ? 0 : ((ao && ao[0] == '1') ? 2 : 1);
	q.rows_per_line = std::max(1, 128 / (c->cfg.n * c->cfg.word));
	/* densest rows / columns in front when they hold enough of the entries (one rank, products in one piece: the SpMV
	 * keeps that many block rows of its operand in LDS, k_spmv_panel) */
	q.hot_cap = 0;
	if (c->cfg.panel && nranks == 1 && q.K == 1) {
		q.hot_cap = spmv_panel_capacity(c->cfg);
		if (const char *e = getenv("BLZ_PANEL_ROWS"))
			q.hot_cap = std::min<int64_t>(q.hot_cap, std::max<int64_t>(0, atoll(e)));
	}
	q.min_share = 0.25;	/* below that the dense block rows are served from L2 at no cost to the fabric: measured on the
				 * structured workload, 17 % of the entries in the panel = no change in time */
	if (const char *e = getenv("BLZ_PANEL_MIN_PCT"))
		q.min_share = atof(e) / 100.0;
	return q;
}

extern "C" int blz_prepare_for(const blz_ctx *c, const blz_coo *M, int right, int nranks, blz_prepared **out)
{
	if (!c || !M || !out || nranks < 1)
		return blz_fail(BLZ_EINVAL, "blz_prepare_for: bad argument");
	const PrepParams q = prep_params(c, M->nrows, M->ncols, M->nnz, nranks);
	return blz_prepare(M, right, nranks, q.K, q.reorder, q.rows_per_line, q.hot_cap, q.min_share, out);
}

extern "C" uint64_t blz_prepare_key(const blz_ctx *c, uint64_t content_hash, int64_t mrows, int64_t mcols, int64_t nnz,
				    int right, int nranks)
{
	if (!c)
		return 0;
	const PrepParams q = prep_params(c, mrows, mcols, nnz, nranks);
	uint64_t h = content_hash ^ 0x9E3779B97F4A7C15ull;
	const uint64_t parts[] = { c->prime, (uint64_t)c->cfg.n, (uint64_t)c->cfg.word, (uint64_t)(right != 0), (uint64_t)nranks,
				   (uint64_t)q.K, (uint64_t)q.reorder, (uint64_t)q.rows_per_line, (uint64_t)q.hot_cap,
				   (uint64_t)(q.min_share * 1e6), (uint64_t)mrows, (uint64_t)mcols, (uint64_t)nnz, 2 /* format */ };
	for (uint64_t x : parts)
		h = (h ^ x) * 0x100000001b3ull;
	return h ? h : 1;
}

extern "C" int blz_set_matrix_prepared(blz_ctx *c, const blz_prepared *P, int rank)
{
	if (!c || !P)
		return blz_fail(BLZ_EINVAL, "blz_set_matrix_prepared: NULL argument");
	const int nranks = P->nranks, right = P->right, K = P->chunks;
	if (rank < 0 || rank >= nranks)
		return blz_fail(BLZ_EINVAL, "blz_set_matrix: rank %d of %d", rank, nranks);
	if (nranks > 1 && (unsigned __int128)nranks * c->prime > ((unsigned __int128)1 << 64))
		return blz_fail(BLZ_EINVAL, "nranks * p must not exceed 2**64 (u64 all-reduce of residues)");
	if (K > 1 && nranks == 1 && !c->force_comm)
		return blz_fail(BLZ_EINVAL, "blz_set_matrix_prepared: the matrix was prepared in %d pieces for a single rank", K);
	if (c->loop && (rank != c->loop_rank || nranks != c->loop->nranks))
		return blz_fail(BLZ_EINVAL, "blz_set_matrix_prepared: rank %d of %d, but the context is rank %d of %d in its loopback communicator",
				rank, nranks, c->loop_rank, c->loop->nranks);
	HIPCHK(hipSetDevice(c->device));
	if (c->iter_graph) {
		hipGraphExecDestroy(c->iter_graph);
		c->iter_graph = nullptr;
	}
	c->right = right;
	c->rank = rank;
	c->nranks = nranks;
	c->fuse_local_off = false;
	/* side 0 = rows of v: rows of M for a left kernel, columns of M for a right kernel
	 * (sequential/lanczos_modp.c:592-593). */
	c->glob_rows[0] = right ? P->ncols : P->nrows;
	c->glob_rows[1] = right ? P->nrows : P->ncols;
	c->row_side[0] = right ? 1 : 0;		/* rows of M   */
	c->row_side[1] = right ? 0 : 1;		/* rows of M^T */
	const int rs = right ? 1 : 0, cs = 1 - rs;	/* side of M's rows / columns */
	for (int sd = 0; sd < 2; sd++) {
		c->perm[sd].clear();
		c->inv[sd].clear();
		c->bounds[sd].assign(P->bounds[sd], P->bounds[sd] + nranks + 1);
		c->stride[sd] = P->stride[sd];
		c->first[sd] = c->bounds[sd][rank];
		c->count[sd] = c->bounds[sd][rank + 1] - c->bounds[sd][rank];
	}
	if (P->has_perm) {
		c->perm[rs].assign(P->perm[0], P->perm[0] + P->nrows);
		c->perm[cs].assign(P->perm[1], P->perm[1] + P->ncols);
		for (int sd = 0; sd < 2; sd++) {
			c->inv[sd].resize(c->perm[sd].size());
			for (size_t r = 0; r < c->perm[sd].size(); r++)
				c->inv[sd][(size_t)c->perm[sd][r]] = (int32_t)r;
		}
	}
	/* what the renumbering found, per product (t = 0: M * x gathers by column; t = 1: M^T * x gathers by row) */
	c->hot_share[0] = P->share[0];
	c->hot_share[1] = P->share[1];
	c->locality[0] = P->locality[0];
	c->locality[1] = P->locality[1];
	c->order_kind = P->order_kind;
	for (int t = 0; t < 2; t++) {
		for (auto &A : c->csr[t])
			free_csr(A);
		c->csr[t].assign((size_t)K, DevCsr{});
	}
	int rc = BLZ_OK;
	/* Short-side form of a product whose operand side is at least 8 times longer than its output side (64-bit words:
	 * the partial sums travel as u64).  BLZ_SHORT_SIDE=0 / 1 forces it off / on (tests, A/B). */
	size_t part_need = 0, rs_need = 0;
	for (int t = 0; t < 2; t++) {
		free_csr(c->csr_short[t]);
		const int rs_t = c->row_side[t], cs_t = 1 - rs_t;
		bool on = (nranks > 1 || c->force_comm) && c->cfg.word == 8 && c->glob_rows[cs_t] >= 8 * c->glob_rows[rs_t];
		if (const char *e = getenv("BLZ_SHORT_SIDE"))
			on = e[0] == '1' && c->cfg.word == 8 && (nranks > 1 || c->force_comm);
		c->short_side[t] = on;
		if (on) {
			part_need = std::max(part_need, (size_t)nranks * (size_t)std::max<int64_t>(c->stride[rs_t], 1) * c->cfg.n * 8);
			rs_need = std::max(rs_need, (size_t)std::max<int64_t>(c->stride[rs_t], 1) * c->cfg.n * 8);
		}
	}
	for (int t = 0; t < 2 && rc == BLZ_OK; t++) {
		if (c->short_side[t]) {
			blz_csr sh;
			if ((rc = blz_prepared_slab_short(P, rank, t, &sh)) != BLZ_OK)
				break;
			rc = upload_csr(c, sh, c->csr_short[t]);
			blz_csr_free(&sh);
			continue;		/* the gathering form of this product is not uploaded at all */
		}
		blz_csr slab;
		memset(&slab, 0, sizeof slab);
		/* product `right` is the second one of the iteration: its (last) launch carries the inner products where that form exists */
		const bool dot_slab = t == right && c->fuse_dot && spmv_dot_supported(c->cfg);
		const bool whole = nranks == 1;		/* one rank: the slab IS the prepared CSR, no copy */
		if (whole)
			slab = P->full[t];
		else if ((rc = blz_prepared_slab(P, rank, t, &slab)) != BLZ_OK)
			break;
		if (K == 1) {
			/* product t gathers block rows by the column index of its slab: columns of M for t = 0, rows of M for t = 1 */
			const int64_t hot_t = c->cfg.panel ? P->hot[t == 0 ? 1 : 0] : 0;
			rc = upload_csr(c, slab, c->csr[t][0], hot_t, dot_slab, nranks == 1 ? c->locality[t] : 1.0);
			{
				/* Gathers that mostly hit (a banded / well-ordered matrix: the locality rule of spmv_plan_staged): the product
				 * is paced by the kernel, and there the staged form with 16-byte lanes plus the inner products as their own
				 * matrix-core kernel beat the fused lane-per-column form -- band matrix, 2 M x 2 M, n = 8: 308 + 55 us against
				 * 389 (profiles/r03_band_switches.txt).  Where the fabric paces the gathers the fused form stays (the inner
				 * products ride along for 20 us). */
				DevCsr &D = c->csr[t][0];
				if (rc == BLZ_OK && dot_slab && nranks == 1 && block_dot_mfma_supported(c->cfg) && D.panel_rows == 0 &&
				    D.locality < 0.3 && !D.uneven && D.outlier_share < 0.02) {
					spmv_plan_staged(c->cfg, slab.row_ptr, D, true);
					c->fuse_local_off = D.st_ok;
				}
			}
			const char *xe = getenv("BLZ_XCD_RANGES");	/* 0 / 1 force it off / on (A/B) */
			/* (an operand of a few MB sits in every L2 anyway: nothing to separate) */
			const bool big = (double)slab.cols * c->cfg.n * c->cfg.word > 8e6;
			c->csr[t][0].xcd_ranges = xe ? xe[0] == '1' : (c->locality[t] < 0.6 && big);
		} else {
			/* columns are positions in the gathered operand of the opposite side: piece k = [k*w, (k+1)*w) */
			const int csd = 1 - c->row_side[t];
			const int64_t width = (c->stride[csd] / K) * nranks;
			std::vector<blz_csr> piece((size_t)K);
			rc = blz_csr_split_columns(&slab, width, K, piece.data());
			for (int k = 0; k < K && rc == BLZ_OK; k++)
				rc = upload_csr(c, piece[(size_t)k], c->csr[t][(size_t)k], 0, dot_slab);
			for (auto &pc : piece)
				blz_csr_free(&pc);
		}
		if (!whole)
			blz_csr_free(&slab);
	}
	if (rc != BLZ_OK)
		return rc;

	if (part_need > c->part_bytes) {
		if (c->part)
			hipFree(c->part);
		c->part = nullptr;
		HIPCHK(hipMalloc(&c->part, part_need));
		c->part_bytes = part_need;
	}
	if (rs_need > c->rs_bytes) {
		if (c->rs_recv)
			hipFree(c->rs_recv);
		c->rs_recv = nullptr;
		HIPCHK(hipMalloc(&c->rs_recv, rs_need));
		c->rs_bytes = rs_need;
	}
	/* One rank: every block can hold either side (the reference sizes all four for the larger one, :597-605, and
	 * blz_spmv takes any pair of blocks).  Several ranks: a block lives on ONE side (side_of), and a rank's slab of the
	 * long side of a tall matrix is many times its slab of the short one (relat9 shape at N = 8: 1.5 M rows against
	 * 69 k) -- each block is sized by its own side (round 2 gave all four the larger: 23 x what V, AV and P need there). */
	for (int b = 0; b < 4; b++) {
		const int64_t rows = nranks == 1 ? std::max(c->stride[0], c->stride[1]) : c->stride[side_of(b)];
		const size_t bytes = (size_t)std::max<int64_t>(rows, 1) * c->cfg.n * c->cfg.word;
		if (c->slab[b])
			hipFree(c->slab[b]);
		c->slab[b] = nullptr;
		HIPCHK(hipMalloc(&c->slab[b], bytes));
		HIPCHK(hipMemset(c->slab[b], 0, bytes));		/* sequential/lanczos_modp.c:617-622 */
		c->slab_bytes[b] = bytes;
	}
	for (int sd = 0; sd < 2; sd++) {
		if (c->gath[sd])
			hipFree(c->gath[sd]);
		c->gath[sd] = nullptr;
		c->gath_holds[sd] = -1;
		/* side sd is gathered by the product whose rows live on the other side; in its short-side form nothing is */
		const int t_gath = c->row_side[0] == 1 - sd ? 0 : 1;
		if (nranks > 1 && !c->short_side[t_gath]) {
			const size_t gb = (size_t)std::max<int64_t>(c->stride[sd], 1) * nranks * c->cfg.n * c->cfg.word;
			HIPCHK(hipMalloc(&c->gath[sd], gb));
			HIPCHK(hipMemset(c->gath[sd], 0, gb));
		}
	}
	while ((int)c->ev_piece.size() < K) {
		hipEvent_t e;
		HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		c->ev_piece.push_back(e);
	}
	HIPCHK(hipMemset(c->ctl, 0, sizeof(DevCtl)));
	c->host_ctl = DevCtl{};
	c->have_matrix = true;
	return BLZ_OK;
}

/* The one-call form: prepare for this context, keep this rank's share, drop the rest.  Several contexts of one process
 * (or several processes on a node) should prepare ONCE and call blz_set_matrix_prepared instead. */
extern "C" int blz_set_matrix(blz_ctx *c, const blz_coo *M, int right, int rank, int nranks)
{
	if (!c || !M)
		return blz_fail(BLZ_EINVAL, "blz_set_matrix: NULL argument");
	if (nranks < 1 || rank < 0 || rank >= nranks)
		return blz_fail(BLZ_EINVAL, "blz_set_matrix: rank %d of %d", rank, nranks);
	blz_prepared *P = nullptr;
	int rc = blz_prepare_for(c, M, right, nranks, &P);
	if (rc != BLZ_OK)
		return rc;
	rc = blz_set_matrix_prepared(c, P, rank);
	blz_prepared_free(P);
	return rc;
}

extern "C" int64_t blz_rows(const blz_ctx *c, int block)
{
	return (c && block >= 0 && block < 4) ? c->glob_rows[side_of(block)] : -1;
}

extern "C" int64_t blz_local_rows(const blz_ctx *c, int block, int64_t *first)
{
	if (!c || block < 0 || block > 3)
		return -1;
	if (first)
		*first = c->first[side_of(block)];
	return c->count[side_of(block)];
}

extern "C" int64_t blz_local_nnz(const blz_ctx *c, int transpose)
{
	if (!c || !c->have_matrix)
		return -1;
	int64_t nnz = 0;
	if (c->short_side[transpose ? 1 : 0])
		return c->csr_short[transpose ? 1 : 0].nnz;
	for (const auto &A : c->csr[transpose ? 1 : 0])
		nnz += A.nnz;
	return nnz;
}

extern "C" int blz_locality(const blz_ctx *c, double locality[2], int *order_kind)
{
	if (!c || !c->have_matrix || !locality)
		return blz_fail(BLZ_EINVAL, "blz_locality: bad argument");
	locality[0] = c->locality[0];
	locality[1] = c->locality[1];
	if (order_kind)
		*order_kind = c->order_kind;
	return BLZ_OK;
}

extern "C" int64_t blz_panel_rows(const blz_ctx *c, int transpose, double *share)
{
	if (!c || !c->have_matrix)
		return -1;
	const int t = transpose ? 1 : 0;
	if (share)
		*share = c->hot_share[t == 0 ? 1 : 0];
	return c->csr[t].empty() ? 0 : c->csr[t][0].panel_rows;
}

extern "C" int64_t blz_matrix_stream_bytes(const blz_ctx *c, int transpose)
{
	if (!c || !c->have_matrix)
		return -1;
	int64_t bytes = 0;
	for (const auto &A : c->csr[transpose ? 1 : 0])
		bytes += (A.rows + 1) * 4 + A.nnz * 4 + (A.val ? A.nnz * 4 : 0);
	return bytes;
}

/* host u64 words -> device words of the context's width */
static int put_words(blz_ctx *c, void *dst, const uint64_t *src, int64_t words)
{
	if (words <= 0)
		return BLZ_OK;
	if (c->cfg.word == 8) {
		HIPCHK(hipMemcpy(dst, src, (size_t)words * 8, hipMemcpyHostToDevice));
		return BLZ_OK;
	}
	std::vector<u32> tmp((size_t)words);
	for (int64_t k = 0; k < words; k++)
		tmp[(size_t)k] = (u32)src[k];
	HIPCHK(hipMemcpy(dst, tmp.data(), (size_t)words * 4, hipMemcpyHostToDevice));
	return BLZ_OK;
}

static int get_words(blz_ctx *c, uint64_t *dst, const void *src, int64_t words)
{
	if (words <= 0)
		return BLZ_OK;
	if (c->cfg.word == 8) {
		HIPCHK(hipMemcpy(dst, src, (size_t)words * 8, hipMemcpyDeviceToHost));
		return BLZ_OK;
	}
	std::vector<u32> tmp((size_t)words);
	HIPCHK(hipMemcpy(tmp.data(), src, (size_t)words * 4, hipMemcpyDeviceToHost));
	for (int64_t k = 0; k < words; k++)
		dst[k] = tmp[(size_t)k];
	return BLZ_OK;
}

/* `rows` block rows: host rows are c->un words wide, device rows c->cfg.n (zero-padded) */
static int put_rows(blz_ctx *c, void *dst, const uint64_t *src, int64_t rows)
{
	const int un = c->un, np = c->cfg.n;
	if (un == np || rows <= 0)
		return put_words(c, dst, src, rows * np);
	std::vector<uint64_t> wide((size_t)rows * np, 0);
	for (int64_t r = 0; r < rows; r++)
		memcpy(wide.data() + (size_t)r * np, src + (size_t)r * un, (size_t)un * sizeof(uint64_t));
	return put_words(c, dst, wide.data(), rows * np);
}

static int get_rows(blz_ctx *c, uint64_t *dst, const void *src, int64_t rows)
{
	const int un = c->un, np = c->cfg.n;
	if (un == np || rows <= 0)
		return get_words(c, dst, src, rows * np);
	std::vector<uint64_t> wide((size_t)rows * np);
	int rc = get_words(c, wide.data(), src, rows * np);
	for (int64_t r = 0; r < rows && rc == BLZ_OK; r++)
		memcpy(dst + (size_t)r * un, wide.data() + (size_t)r * np, (size_t)un * sizeof(uint64_t));
	return rc;
}

#define NEED_MATRIX(c)                                                                  \
	do {                                                                            \
		if (!(c) || !(c)->have_matrix)                                          \
			return blz_fail(BLZ_EINVAL, "no matrix loaded (blz_set_matrix)"); \
		HIPCHK(hipSetDevice((c)->device));                                      \
	} while (0)

extern "C" int blz_short_side(const blz_ctx *c, int transpose)
{
	return (c && c->have_matrix) ? (int)c->short_side[transpose ? 1 : 0] : -1;
}

/* the full-length partial product a short-side blz_spmv left behind (external-exchange mode: the caller does the
 * reduce-scatter): rows(dst side) x n words in the ORIGINAL numbering, sums not yet reduced mod p */
extern "C" int blz_get_partial(blz_ctx *c, int transpose, uint64_t *host)
{
	NEED_MATRIX(c);
	const int t = transpose ? 1 : 0;
	if (!host || !c->short_side[t])
		return blz_fail(BLZ_EINVAL, "blz_get_partial: product %d is not in the short-side form", t);
	HIPCHK(hipStreamSynchronize(c->stream));
	const int rs_t = c->row_side[t], np = c->cfg.n, un = c->un;
	std::vector<uint64_t> pad((size_t)c->nranks * c->stride[rs_t] * np);
	HIPCHK(hipMemcpy(pad.data(), c->part, pad.size() * 8, hipMemcpyDeviceToHost));
	for (int g = 0; g < c->nranks; g++)
		for (int64_t q = 0; q < c->bounds[rs_t][g + 1] - c->bounds[rs_t][g]; q++) {
			const int64_t solver_row = c->bounds[rs_t][g] + q;
			const int64_t orig = c->inv[rs_t].empty() ? solver_row : c->inv[rs_t][(size_t)solver_row];
			memcpy(host + (size_t)orig * un, pad.data() + ((size_t)g * c->stride[rs_t] + q) * np, (size_t)un * 8);
		}
	return BLZ_OK;
}

/* host block in ORIGINAL numbering -> one contiguous array in the solver's numbering (and back) */
static void to_solver_order(const blz_ctx *c, int sd, const uint64_t *host, uint64_t *out)
{
	const int n = c->un;
	const int64_t rows = c->glob_rows[sd];
	if (c->perm[sd].empty()) {
		memcpy(out, host, (size_t)rows * n * sizeof(uint64_t));
		return;
	}
	for (int64_t r = 0; r < rows; r++)
		memcpy(out + (size_t)c->perm[sd][(size_t)r] * n, host + (size_t)r * n, (size_t)n * sizeof(uint64_t));
}

extern "C" int blz_set_block(blz_ctx *c, int block, const uint64_t *host)
{
	NEED_MATRIX(c);
	if (block < 0 || block > 3 || !host)
		return blz_fail(BLZ_EINVAL, "blz_set_block: bad argument");
	HIPCHK(hipStreamSynchronize(c->stream));
	const int sd = side_of(block), n = c->un, np = c->cfg.n;
	std::vector<uint64_t> tmp;
	const uint64_t *src = host;
	if (!c->perm[sd].empty()) {
		tmp.resize((size_t)std::max<int64_t>(c->glob_rows[sd], 1) * n);
		to_solver_order(c, sd, host, tmp.data());
		src = tmp.data();
	}
	HIPCHK(hipStreamSynchronize(c->xstream));
	/* this rank's rows */
	int rc = put_rows(c, c->slab[block], src + c->first[sd] * n, c->count[sd]);
	if (rc != BLZ_OK || c->nranks == 1 || !c->gath[sd])
		return rc;	/* (a side whose product runs in the short-side form has no gathered copy) */
	/* and, with several ranks, the whole gathered operand (an all-gather done by the caller): piece k of rank g
	 * sits at (k * nranks + g) * piece rows */
	const int K = (int)c->csr[0].size();
	const int64_t piece = c->stride[sd] / K;
	for (int g = 0; g < c->nranks; g++) {
		const int64_t b0 = c->bounds[sd][g], cnt = c->bounds[sd][g + 1] - b0;
		for (int k = 0; k < K; k++) {
			const int64_t q0 = (int64_t)k * piece, q1 = std::min<int64_t>(q0 + piece, cnt);
			if (q1 <= q0)
				break;
			char *dst = (char *)c->gath[sd] + (size_t)(((int64_t)k * c->nranks + g) * piece) * np * c->cfg.word;
			rc = put_rows(c, dst, src + (b0 + q0) * n, q1 - q0);
			if (rc != BLZ_OK)
				return rc;
		}
	}
	c->gath_holds[sd] = block;
	return BLZ_OK;
}

extern "C" int blz_get_block(blz_ctx *c, int block, uint64_t *host)
{
	NEED_MATRIX(c);
	if (block < 0 || block > 3 || !host)
		return blz_fail(BLZ_EINVAL, "blz_get_block: bad argument");
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipStreamSynchronize(c->xstream));
	const int sd = side_of(block), n = c->un;
	if (c->perm[sd].empty())
		return get_rows(c, host + c->first[sd] * n, slab_ptr(c, block), c->count[sd]);
	std::vector<uint64_t> tmp((size_t)std::max<int64_t>(c->count[sd], 1) * n);
	int rc = get_rows(c, tmp.data(), slab_ptr(c, block), c->count[sd]);
	if (rc != BLZ_OK)
		return rc;
	for (int64_t q = 0; q < c->count[sd]; q++)
		memcpy(host + (size_t)c->inv[sd][(size_t)(c->first[sd] + q)] * n, tmp.data() + (size_t)q * n,
		       (size_t)n * sizeof(uint64_t));
	return BLZ_OK;
}

extern "C" int blz_owner_of_row(const blz_ctx *c, int block, int64_t row)
{
	if (!c || !c->have_matrix || block < 0 || block > 3)
		return -1;
	const int sd = side_of(block);
	if (row < 0 || row >= c->glob_rows[sd])
		return -1;
	const int64_t r = c->perm[sd].empty() ? row : c->perm[sd][(size_t)row];
	int g = 0;
	while (g + 1 < c->nranks && c->bounds[sd][g + 1] <= r)
		g++;
	return g;
}

static int small_off(const blz_ctx *c, int which, int *words)
{
	const int nn = c->cfg.n * c->cfg.n;
	*words = which == BLZ_D ? c->cfg.n : nn;
	switch (which) {
	case BLZ_VTAV: return 0;
	case BLZ_VTAAV: return nn;
	case BLZ_WINV: return 2 * nn;
	case BLZ_D: return 3 * nn;
	default: return -1;
	}
}

extern "C" int blz_set_small(blz_ctx *c, int which, const uint64_t *host)
{
	if (!c || !host)
		return blz_fail(BLZ_EINVAL, "blz_set_small: NULL argument");
	HIPCHK(hipSetDevice(c->device));
	int words, off = small_off(c, which, &words);
	if (off < 0)
		return blz_fail(BLZ_EINVAL, "blz_set_small: unknown operand %d", which);
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<u64> dev((size_t)words, 0);		/* the caller's un x un (or un) words inside the padded operand */
	const int un = c->un, np = c->cfg.n;
	if (which == BLZ_D)
		memcpy(dev.data(), host, (size_t)un * sizeof(u64));
	else
		for (int i = 0; i < un; i++)
			memcpy(dev.data() + (size_t)i * np, host + (size_t)i * un, (size_t)un * sizeof(u64));
	HIPCHK(hipMemcpy(c->small + off, dev.data(), (size_t)words * sizeof(u64), hipMemcpyHostToDevice));
	return BLZ_OK;
}

extern "C" int blz_get_small(blz_ctx *c, int which, uint64_t *host)
{
	if (!c || !host)
		return blz_fail(BLZ_EINVAL, "blz_get_small: NULL argument");
	HIPCHK(hipSetDevice(c->device));
	int words, off = small_off(c, which, &words);
	if (off < 0)
		return blz_fail(BLZ_EINVAL, "blz_get_small: unknown operand %d", which);
	HIPCHK(hipStreamSynchronize(c->stream));
	std::vector<u64> dev((size_t)words);
	HIPCHK(hipMemcpy(dev.data(), c->small + off, (size_t)words * sizeof(u64), hipMemcpyDeviceToHost));
	const int un = c->un, np = c->cfg.n;
	if (which == BLZ_D)
		memcpy(host, dev.data(), (size_t)un * sizeof(u64));
	else
		for (int i = 0; i < un; i++)
			memcpy(host + (size_t)i * un, dev.data() + (size_t)i * np, (size_t)un * sizeof(u64));
	return BLZ_OK;
}

extern "C" int blz_init_v(blz_ctx *c)
{
	NEED_MATRIX(c);
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipStreamSynchronize(c->xstream));
	for (int b = 0; b < 4; b++)
		HIPCHK(hipMemset(c->slab[b], 0, c->slab_bytes[b]));
	c->gath_holds[0] = c->gath_holds[1] = -1;
	HIPCHK(hipMemset(c->ctl, 0, sizeof(DevCtl)));
	c->host_ctl = DevCtl{};
	/* :624-625: one sequential stream over the whole block in ORIGINAL row order; a rank keeps the rows it owns. */
	const int n = c->un;
	const int64_t keep = c->count[0] * n, lo = c->first[0], hi = c->first[0] + c->count[0];
	std::vector<uint64_t> mine((size_t)std::max<int64_t>(keep, 1));
	uint64_t s[4];
	blz_rng_seed(s);
	for (int64_t r = 0; r < c->glob_rows[0]; r++) {
		const int64_t nr = c->perm[0].empty() ? r : c->perm[0][(size_t)r];
		if (nr >= lo && nr < hi) {
			for (int l = 0; l < n; l++)
				mine[(size_t)((nr - lo) * n + l)] = blz_rng_next(s) % c->prime;
		} else {
			for (int l = 0; l < n; l++)
				(void)blz_rng_next(s);
		}
	}
	return put_rows(c, slab_ptr(c, BLZ_V), mine.data(), c->count[0]);
}

/* ---- exchange steps (RCCL over xGMI).  No-ops on a single rank. ---- */

static inline bool exchanging(const blz_ctx *c)
{
	return !c->external_exchange && (c->nranks > 1 || (c->force_comm && (c->comm || c->loop)));
}

/* where k_dot_finalize puts this rank's sums: straight into `small` on one rank, into the send buffer otherwise */
static inline u64 *dot_out(blz_ctx *c);

/* ---- the four collectives of the solver, over RCCL or over the loopback group ---- */

enum { LOOP_GATHER = 0, LOOP_SUM64 = 1, LOOP_SUM64_SEGMENT = 2, LOOP_SUM32 = 3 };

/* one loopback collective of rank c->rank on stream st: `count` = bytes per rank (gather) or words (sums) */
static int loop_collective(blz_ctx *c, int kind, const void *send, void *recv, size_t count, hipStream_t st)
{
	blz_loop_group *g = c->loop;
	const int r = c->loop_rank, N = g->nranks;
	g->send[(size_t)r] = send;
	HIPCHK(hipEventRecord(g->ready[(size_t)r], st));
	if (!g->meet())
		return blz_fail(BLZ_ECOMM, "loopback communicator: a rank did not arrive (or gave up)");
	for (int q = 0; q < N; q++)
		if (q != r)
			HIPCHK(hipStreamWaitEvent(st, g->ready[(size_t)q], 0));
	if (kind == LOOP_GATHER) {
		for (int q = 0; q < N; q++)
			HIPCHK(launch_copy(c->cfg, (char *)recv + (size_t)q * count, g->send[(size_t)q], count, st));
	} else {
		const void *src[BLZ_LOOP_MAX_RANKS];
		for (int q = 0; q < N; q++)
			src[q] = kind == LOOP_SUM64_SEGMENT ? (const void *)((const u64 *)g->send[(size_t)q] + (size_t)r * count) : g->send[(size_t)q];
		HIPCHK(launch_sum_buffers(src, N, recv, (long long)count, kind == LOOP_SUM32 ? 4 : 8, st));
	}
	HIPCHK(hipEventRecord(g->done[(size_t)r], st));
	if (!g->meet())
		return blz_fail(BLZ_ECOMM, "loopback communicator: a rank did not arrive (or gave up)");
	for (int q = 0; q < N; q++)
		if (q != r)
			HIPCHK(hipStreamWaitEvent(st, g->done[(size_t)q], 0));
	return BLZ_OK;
}

static int coll_allgather(blz_ctx *c, const void *send, void *recv, size_t bytes, hipStream_t st)
{
	if (c->loop)
		return loop_collective(c, LOOP_GATHER, send, recv, bytes, st);
	if (!c->comm)
		return blz_fail(BLZ_ECOMM, "nranks > 1 but blz_comm_init was not called");
	NCCLCHK(g_rccl.AllGather(send, recv, bytes, ncclUint8, c->comm, st));
	return BLZ_OK;
}

static int coll_allreduce_u64(blz_ctx *c, const void *send, void *recv, size_t words, hipStream_t st)
{
	if (c->loop)
		return loop_collective(c, LOOP_SUM64, send, recv, words, st);
	if (!c->comm)
		return blz_fail(BLZ_ECOMM, "nranks > 1 but blz_comm_init was not called");
	NCCLCHK(g_rccl.AllReduce(send, recv, words, ncclUint64, ncclSum, c->comm, st));
	return BLZ_OK;
}

static int coll_allreduce_i32(blz_ctx *c, const void *send, void *recv, size_t words, hipStream_t st)
{
	if (c->loop)
		return loop_collective(c, LOOP_SUM32, send, recv, words, st);
	if (!c->comm)
		return blz_fail(BLZ_ECOMM, "nranks > 1 but blz_comm_init was not called");
	NCCLCHK(g_rccl.AllReduce(send, recv, words, ncclInt32, ncclSum, c->comm, st));
	return BLZ_OK;
}

/* recv[i] = sum over the ranks of send_q[rank * words + i] */
static int coll_reduce_scatter_u64(blz_ctx *c, const void *send, void *recv, size_t words, hipStream_t st)
{
	if (c->loop)
		return loop_collective(c, LOOP_SUM64_SEGMENT, send, recv, words, st);
	if (!c->comm)
		return blz_fail(BLZ_ECOMM, "nranks > 1 but blz_comm_init was not called");
	NCCLCHK(g_rccl.ReduceScatter(send, recv, words, ncclUint64, ncclSum, c->comm, st));
	return BLZ_OK;
}

static int allreduce_dots(blz_ctx *c)
{
	if (!exchanging(c))
		return BLZ_OK;
	Span sp(c, PK_AR);
	/* residues < p and nranks * p <= 2^64 (checked in blz_set_matrix): the u64 sum cannot wrap; the
	 * semi_inverse kernel reduces it mod p.  (mpi/lanczos_modp.c:1209-1247 does this by hand.)
	 * From the rank's own partial sums (dot_send) into a landing place of its own (dot_recv), never into `small`: the
	 * collective is enqueued by the host whatever the stop flag says, and the iterations a batch enqueues past the stop
	 * would leave raw sums (up to nranks * (p - 1)) in `small` -- the semi-inverse kernel, the one that turns them
	 * into residues, is a no-op by then.  It reads dot_recv and writes `small` only while the flag is down. */
	return coll_allreduce_u64(c, c->dot_send, c->dot_recv, (size_t)2 * c->cfg.n * c->cfg.n, c->stream);
}

static inline u64 *dot_out(blz_ctx *c) { return exchanging(c) ? c->dot_send : c->small; }

/* where the semi-inverse finds vtAv | vtAAv */
static inline const u64 *dot_sums(blz_ctx *c) { return exchanging(c) ? c->dot_recv : c->small; }

/*
 * One product of the iteration: slab[dst] = (transpose ? M^T : M)[this rank's rows] * block `src`, with the
 * exchange of `src` pipelined against it.  The exchange stream all-gathers piece k of every rank's slab into
 * gath[side]; the compute stream waits for piece k only, then multiplies by csr[transpose][k] (entries whose
 * columns lie in piece k), accumulating into slab[dst] from the second piece on.  With K pieces the product hides
 * all but 1/K of itself behind the exchange; K = 1 is one all-gather followed by one product.
 * with_dot: the last piece carries block_dot_products as its epilogue (second product only); *nb = partial rows.
 */
static int enqueue_product(blz_ctx *c, int transpose, int src, int dst, bool with_dot, int *nb)
{
	const bool xchg = exchanging(c);
	const int cls = transpose == !c->right ? PK_SPMV1 : PK_SPMV2;
	if (c->short_side[transpose]) {
		/* partial product of this rank's own rows of the operand, full length on the output side; reduce-scatter of the
		 * u64 sums; mod p.  Nothing is gathered. */
		const int rs_t = c->row_side[transpose];
		if (with_dot)
			return blz_fail(BLZ_EINVAL, "enqueue_product: the short-side form has no fused inner products");
		{
			Span sp(c, cls);
			HIPCHK(launch_spmv(c->cfg, c->csr_short[transpose], slab_ptr(c, src), c->part, 0, c->ctl, c->stream));
		}
		if (xchg) {
			{
				Span sp(c, PK_RS);
				int rc_ = coll_reduce_scatter_u64(c, c->part, c->rs_recv, (size_t)c->stride[rs_t] * c->cfg.n, c->stream);
				if (rc_ != BLZ_OK)
					return rc_;
			}
			/* out of place and stop-aware: past the stop `part` is stale (possibly the OTHER product's), the collective
			 * runs all the same, and slab[dst] must keep the last real product (blz_final_check reads TMP) */
			HIPCHK(launch_reduce_modp(c->cfg, c->slab[dst], c->rs_recv, c->count[rs_t] * c->cfg.n, c->ctl, c->stream));
		}
		return BLZ_OK;
	}
	const int K = (int)c->csr[transpose].size(), sd = side_of(src);
	if (xchg) {
		if (!c->comm && !c->loop)
			return blz_fail(BLZ_ECOMM, "nranks > 1 but blz_comm_init was not called");
		/* the exchange may start once everything enqueued so far (the producer of `src`, and every earlier
		 * reader of the gathered buffer it overwrites) has run */
		HIPCHK(hipEventRecord(c->ev_prod, c->stream));
		HIPCHK(hipStreamWaitEvent(c->xstream, c->ev_prod, 0));
		const size_t piece = (size_t)(c->stride[sd] / K) * c->cfg.n * c->cfg.word;
		char *recv = c->nranks == 1 ? (char *)c->slab[src] : (char *)c->gath[sd];	/* 1 rank: in place */
		for (int k = 0; k < K; k++) {
			{
				Span sp(c, src == BLZ_V ? PK_AG_V : PK_AG_T, c->xstream);
				int rc_ = coll_allgather(c, (char *)c->slab[src] + (size_t)k * piece, recv + (size_t)k * c->nranks * piece, piece,
							 c->xstream);
				if (rc_ != BLZ_OK)
					return rc_;
			}
			HIPCHK(hipEventRecord(c->ev_piece[(size_t)k], c->xstream));
		}
		c->gath_holds[sd] = src;
	} else if (c->nranks > 1 && c->gath_holds[sd] != src) {
		return blz_fail(BLZ_EINVAL, "external-exchange mode: block %d has not been gathered (blz_set_block)", src);
	}
	const void *X = operand_ptr(c, src);
	for (int k = 0; k < K; k++) {
		if (xchg)
			HIPCHK(hipStreamWaitEvent(c->stream, c->ev_piece[(size_t)k], 0));
		Span sp(c, cls);
		const DevCsr &A = c->csr[transpose][(size_t)k];
		if (with_dot && k == K - 1)
			HIPCHK(launch_spmv_dot(c->cfg, A, X, slab_ptr(c, dst), slab_ptr(c, BLZ_V), k > 0, c->partial,
					       c->max_dot_blocks, nb, c->ctl, c->stream));
		else
			HIPCHK(launch_spmv(c->cfg, A, X, slab_ptr(c, dst), k > 0, c->ctl, c->stream));
	}
	return BLZ_OK;
}

static int enqueue_spmv(blz_ctx *c, int transpose, int src, int dst)
{
	return enqueue_product(c, transpose, src, dst, false, nullptr);
}

static int enqueue_dot(blz_ctx *c)
{
	int nb = 0;
	{
	Span sp(c, PK_DOT);
	HIPCHK(launch_block_dot(c->cfg, slab_ptr(c, BLZ_V), slab_ptr(c, BLZ_AV), c->count[0], c->partial,
				std::min(c->max_dot_blocks, c->cfg.num_cu * 8), &nb, c->ctl, c->stream));
	HIPCHK(launch_dot_finalize(c->cfg, c->partial, nb, dot_out(c), c->ctl, c->stream));
	}
	return allreduce_dots(c);
}

static int enqueue_ortho(blz_ctx *c, bool img_ready = false)
{
	Span sp(c, PK_ORTHO);
	HIPCHK(launch_orthogonalize(c->cfg, slab_ptr(c, BLZ_V), slab_ptr(c, BLZ_AV), slab_ptr(c, BLZ_P), c->count[0],
				    c->small, c->ctl, c->stream, img_ready));
	return BLZ_OK;
}

static int fetch_ctl(blz_ctx *c)
{
	HIPCHK(launch_publish_ctl(c->ctl, c->ctl_pinned_dev, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->host_ctl = *c->ctl_pinned;
	return BLZ_OK;
}

extern "C" int blz_spmv(blz_ctx *c, int transpose, int src_block, int dst_block)
{
	NEED_MATRIX(c);
	if (src_block < 0 || src_block > 3 || dst_block < 0 || dst_block > 3 || src_block == dst_block)
		return blz_fail(BLZ_EINVAL, "blz_spmv: bad block selectors");
	{	/* several ranks: a block lives on one side and its slab is sized for that side only */
		const int t = transpose ? 1 : 0;
		if (c->nranks > 1 && (side_of(dst_block) != c->row_side[t] || side_of(src_block) != 1 - c->row_side[t]))
			return blz_fail(BLZ_EINVAL, "blz_spmv: on several ranks product %d reads a block of side %d and writes one of side %d",
					t, 1 - c->row_side[t], c->row_side[t]);
	}
	int rc = enqueue_spmv(c, transpose ? 1 : 0, src_block, dst_block);
	if (rc != BLZ_OK)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	return BLZ_OK;
}

extern "C" int blz_block_dot(blz_ctx *c, uint64_t *vtAv, uint64_t *vtAAv)
{
	NEED_MATRIX(c);
	int rc = enqueue_dot(c);
	if (rc != BLZ_OK)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	const int np = c->cfg.n, un = c->un, nn = np * np;
	std::vector<u64> h((size_t)2 * nn);
	if (exchanging(c)) {	/* the sums over the ranks -> residues in `small`, where a following blz_semi_inverse looks */
		HIPCHK(launch_reduce_modp(c->cfg, c->small, c->dot_recv, 2 * nn, c->ctl, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	HIPCHK(hipMemcpy(h.data(), c->small, (size_t)2 * nn * sizeof(u64), hipMemcpyDeviceToHost));
	for (int i = 0; i < un; i++) {
		if (vtAv)
			memcpy(vtAv + (size_t)i * un, h.data() + (size_t)i * np, (size_t)un * sizeof(u64));
		if (vtAAv)
			memcpy(vtAAv + (size_t)i * un, h.data() + nn + (size_t)i * np, (size_t)un * sizeof(u64));
	}
	return BLZ_OK;
}

extern "C" int blz_semi_inverse(blz_ctx *c, int *npiv, uint64_t *winv, uint64_t *d)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_semi_inverse: NULL context");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(launch_semi_inverse(c->cfg, c->small, c->small, c->ctl, 0, 0, c->stream));
	int rc = fetch_ctl(c);
	if (rc != BLZ_OK)
		return rc;
	if (npiv)
		*npiv = c->host_ctl.npiv;
	if (winv && (rc = blz_get_small(c, BLZ_WINV, winv)) != BLZ_OK)
		return rc;
	if (d && (rc = blz_get_small(c, BLZ_D, d)) != BLZ_OK)
		return rc;
	return BLZ_OK;
}

extern "C" int blz_orthogonalize(blz_ctx *c)
{
	NEED_MATRIX(c);
	int rc = enqueue_ortho(c);
	if (rc != BLZ_OK)
		return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	return BLZ_OK;
}

/* One pass of the loop body, sequential/lanczos_modp.c:635-656, enqueued on the stream. */
static int enqueue_iteration(blz_ctx *c)
{
	int rc;
	if ((rc = enqueue_product(c, !c->right, BLZ_V, BLZ_TMP, false, nullptr)) != BLZ_OK) return rc;	/* :635 */
	if (c->fuse_dot && !c->fuse_local_off && spmv_dot_supported(c->cfg) && c->count[0] > 0 && !c->short_side[c->right]) {
		int nb = 0;							/* :636 + :640 in one kernel */
		if ((rc = enqueue_product(c, c->right, BLZ_TMP, BLZ_AV, true, &nb)) != BLZ_OK) return rc;
		{
			Span sp(c, PK_DOT);
			HIPCHK(launch_dot_finalize(c->cfg, c->partial, nb, dot_out(c), c->ctl, c->stream));
		}
		if ((rc = allreduce_dots(c)) != BLZ_OK) return rc;
	} else {
		if ((rc = enqueue_product(c, c->right, BLZ_TMP, BLZ_AV, false, nullptr)) != BLZ_OK) return rc;	/* :636 */
		if ((rc = enqueue_dot(c)) != BLZ_OK) return rc;				/* :640 */
	}
	/* the semi-inverse kernel also writes the coefficient image when the block update runs on the matrix cores
	 * (round 2: a launch of its own between the two) */
	const bool img = ortho_uses_mfma(c->cfg, c->count[0]);
	{
		Span sp(c, PK_SEMI);
		HIPCHK(launch_semi_inverse(c->cfg, dot_sums(c), c->small, c->ctl, 1, img, c->stream));	/* :644 */
	}
	return enqueue_ortho(c, img);							/* :652-656 */
}

extern "C" int blz_iterate(blz_ctx *c, int max_iters, int *done, int *stopped, float *ms)
{
	NEED_MATRIX(c);
	if (max_iters < 0)
		return blz_fail(BLZ_EINVAL, "blz_iterate: max_iters < 0");
	if (c->external_exchange && c->nranks > 1)
		return blz_fail(BLZ_EINVAL, "blz_iterate: the context is in external-exchange mode");
	const long long before = c->host_ctl.iterations;
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	const bool graph = c->use_graph && c->nranks == 1 && !c->force_comm && !c->profiling && max_iters > 1;
	if (graph && !c->iter_graph) {
		/* the loop body is a fixed sequence of launches with fixed arguments (all state lives in device memory):
		 * record it once, replay it per iteration */
		hipGraph_t g = nullptr;
		HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
		int rc = enqueue_iteration(c);
		hipError_t e = hipStreamEndCapture(c->stream, &g);
		if (rc != BLZ_OK)
			return rc;
		HIPCHK(e);
		HIPCHK(hipGraphInstantiate(&c->iter_graph, g, nullptr, nullptr, 0));
		hipGraphDestroy(g);
	}
	for (int it = 0; it < max_iters; it++) {
		if (graph) {
			HIPCHK(hipGraphLaunch(c->iter_graph, c->stream));
			continue;
		}
		int rc = enqueue_iteration(c);
		if (rc != BLZ_OK)
			return rc;
	}
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	int rc = fetch_ctl(c);
	if (rc != BLZ_OK)
		return rc;
	if (ms)
		HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
	if (done)
		*done = (int)(c->host_ctl.iterations - before);
	if (stopped)
		*stopped = c->host_ctl.stop;
	return BLZ_OK;
}

extern "C" int64_t blz_iterations(const blz_ctx *c) { return c ? c->host_ctl.iterations : -1; }

extern "C" int blz_set_iterations(blz_ctx *c, int64_t iterations)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_set_iterations: NULL context");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->host_ctl.iterations = iterations;
	HIPCHK(hipMemcpy(c->ctl, &c->host_ctl, sizeof(DevCtl), hipMemcpyHostToDevice));
	return BLZ_OK;
}

extern "C" int blz_final_check(blz_ctx *c, int *v_nonzero, int *vtm_zero)
{
	NEED_MATRIX(c);
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemset(&c->ctl->flag_v_nonzero, 0, 2 * sizeof(int)));
	HIPCHK(launch_any_nonzero(c->cfg, slab_ptr(c, BLZ_V), c->count[0] * c->cfg.n, &c->ctl->flag_v_nonzero, c->stream));
	HIPCHK(launch_any_nonzero(c->cfg, slab_ptr(c, BLZ_TMP), c->count[1] * c->cfg.n, &c->ctl->flag_t_nonzero, c->stream));
	if (c->nranks > 1 && !c->external_exchange) {
		/* out of place (the sums land in dot_recv, idle at this point, and are copied back): the loopback communicator
		 * reads the other ranks' send buffers while they write their own results */
		int rc_ = coll_allreduce_i32(c, &c->ctl->flag_v_nonzero, c->dot_recv, 2, c->stream);
		if (rc_ != BLZ_OK)
			return rc_;
		HIPCHK(hipMemcpyAsync(&c->ctl->flag_v_nonzero, c->dot_recv, 2 * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
	}
	int rc = fetch_ctl(c);
	if (rc != BLZ_OK)
		return rc;
	if (v_nonzero)
		*v_nonzero = c->host_ctl.flag_v_nonzero != 0;
	if (vtm_zero)
		*vtm_zero = c->host_ctl.flag_t_nonzero == 0;
	return BLZ_OK;
}

extern "C" int blz_time_kernel(blz_ctx *c, int which, int reps, float *ms_mean)
{
	NEED_MATRIX(c);
	if (reps < 1 || !ms_mean || which < 0 || which > 3)
		return blz_fail(BLZ_EINVAL, "blz_time_kernel: bad argument");
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	for (int r = 0; r < reps; r++) {
		int rc = BLZ_OK;
		switch (which) {
		case 0: rc = enqueue_spmv(c, !c->right, BLZ_V, BLZ_TMP); break;
		case 1: rc = enqueue_spmv(c, c->right, BLZ_TMP, BLZ_AV); break;
		case 2: rc = enqueue_dot(c); break;
		case 3: rc = enqueue_ortho(c); break;
		}
		if (rc != BLZ_OK)
			return rc;
	}
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipEventSynchronize(c->ev1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	*ms_mean = ms / reps;
	return BLZ_OK;
}

extern "C" int blz_profile(blz_ctx *c, int enable)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_profile: NULL context");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (auto &sp : c->spans) {
		c->pool.push_back(sp.a);
		c->pool.push_back(sp.b);
	}
	c->spans.clear();
	c->profiling = enable != 0;
	return BLZ_OK;
}

extern "C" int blz_profile_read(blz_ctx *c, double *ms_sum, int64_t *launches)
{
	if (!c || !ms_sum || !launches)
		return blz_fail(BLZ_EINVAL, "blz_profile_read: NULL argument");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	for (int k = 0; k < PK_COUNT; k++) {
		ms_sum[k] = 0;
		launches[k] = 0;
	}
	for (auto &sp : c->spans) {
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
		ms_sum[sp.cls] += ms;
		launches[sp.cls] += 1;
	}
	return BLZ_OK;
}

/*
 * Asynchronous snapshot of (v, p, iteration count) for checkpoints -- openMP/lanczos_modp.c:1013-1022 stops the loop,
 * and round 1 did too (two synchronous blz_get_block calls and the file write on the loop thread).
 * blz_snapshot_begin: between two blz_iterate calls; copies this rank's rows of v and p device-to-device on the compute
 * stream (HBM speed), then enqueues the device-to-host copies of that snapshot into pinned staging on a stream of their
 * own and returns: the loop goes on while PCIe drains the snapshot (measured with a checkpoint every SECOND: 10.8 % loss on
 * the config-5 quarter shape when the loop waited for the 2 x 1.6 GB transfer, tools/exp_checkpoint.py).  If the two extra
 * slabs do not fit in HBM the copies read v and p directly and the compute stream waits for them.
 * blz_snapshot_wait: blocks until the copies have landed and unpacks this rank's rows into v / p (original numbering,
 * caller's width).  It touches only the snapshot's own staging and events, so it MAY be called from another host thread
 * (the checkpoint writer) while the owning thread is inside blz_iterate -- the one exception to one-thread-per-handle.
 */
extern "C" int blz_snapshot_begin(blz_ctx *c)
{
	NEED_MATRIX(c);
	if (c->snap_pending.load(std::memory_order_acquire))
		return blz_fail(BLZ_EINVAL, "blz_snapshot_begin: the previous snapshot has not been collected (blz_snapshot_wait)");
	if (!c->cstream) {
		HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
		HIPCHK(hipEventCreateWithFlags(&c->ev_snap_go, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&c->ev_snap_done, hipEventDisableTiming));
	}
	const size_t bytes = (size_t)std::max<int64_t>(c->count[0], 1) * c->cfg.n * c->cfg.word;
	if (bytes > c->snap_bytes) {
		for (int b = 0; b < 2; b++) {
			if (c->snap_host[b]) hipHostFree(c->snap_host[b]);
			if (c->snap_dev[b]) hipFree(c->snap_dev[b]);
			c->snap_host[b] = c->snap_dev[b] = nullptr;
			HIPCHK(hipHostMalloc(&c->snap_host[b], bytes, hipHostMallocDefault));
		}
		/* two more slabs of HBM let the PCIe transfer run beside the loop; without them the loop waits for it */
		if (hipMalloc(&c->snap_dev[0], bytes) != hipSuccess || hipMalloc(&c->snap_dev[1], bytes) != hipSuccess) {
			(void)hipGetLastError();
			for (void *&d : c->snap_dev) {
				if (d) hipFree(d);
				d = nullptr;
			}
		}
		c->snap_bytes = bytes;
	}
	const bool staged = c->snap_dev[0] && c->snap_dev[1];
	if (staged) {
		/* device-to-device on the compute stream by a streaming kernel (2 x slab at HBM speed: 1.3 ms for 2 x 1.6 GB; the
		 * runtime's own device-to-device copy cost ~60 ms here), then the host copy reads
		 * the snapshot while later iterations overwrite v and p */
		HIPCHK(launch_copy(c->cfg, c->snap_dev[0], c->slab[BLZ_V], bytes, c->stream));
		HIPCHK(launch_copy(c->cfg, c->snap_dev[1], c->slab[BLZ_P], bytes, c->stream));
	}
	HIPCHK(hipEventRecord(c->ev_snap_go, c->stream));
	HIPCHK(hipStreamWaitEvent(c->cstream, c->ev_snap_go, 0));
	HIPCHK(hipMemcpyAsync(c->snap_host[0], staged ? c->snap_dev[0] : c->slab[BLZ_V], bytes, hipMemcpyDeviceToHost, c->cstream));
	HIPCHK(hipMemcpyAsync(c->snap_host[1], staged ? c->snap_dev[1] : c->slab[BLZ_P], bytes, hipMemcpyDeviceToHost, c->cstream));
	HIPCHK(hipEventRecord(c->ev_snap_done, c->cstream));
	if (!staged)
		HIPCHK(hipStreamWaitEvent(c->stream, c->ev_snap_done, 0));	/* later kernels overwrite v and p in place */
	c->snap_iterations = c->host_ctl.iterations;
	c->snap_pending.store(true, std::memory_order_release);
	return BLZ_OK;
}

extern "C" int blz_snapshot_wait(blz_ctx *c, uint64_t *v, uint64_t *p, int64_t *iterations)
{
	if (!c || (!v) != (!p))
		return blz_fail(BLZ_EINVAL, "blz_snapshot_wait: NULL argument");
	if (!c->snap_pending.load(std::memory_order_acquire))
		return blz_fail(BLZ_EINVAL, "blz_snapshot_wait: no snapshot in flight");
	HIPCHK(hipSetDevice(c->device));
	/* polled, not hipEventSynchronize: this runs on the writer's thread while the owner keeps launching, and a blocking
	 * wait inside the runtime held the owner's launches back for the whole transfer (the loop lost as much time as if it
	 * had waited for the copy itself) */
	for (;;) {
		const hipError_t q = hipEventQuery(c->ev_snap_done);
		if (q == hipSuccess)
			break;
		if (q != hipErrorNotReady)
			return blz_fail(BLZ_EHIP, "blz_snapshot_wait: %s", hipGetErrorString(q));
		usleep(500);
	}
	const int un = c->un, np = c->cfg.n, sd = 0;
	uint64_t *dst[2] = { v, p };
	for (int b = 0; b < 2 && v; b++) {	/* v == p == NULL: the snapshot is dropped (a writer that failed elsewhere still collects) */
		const char *src = (const char *)c->snap_host[b];
		for (int64_t q = 0; q < c->count[sd]; q++) {
			const int64_t solver_row = c->first[sd] + q;
			const int64_t orig = c->inv[sd].empty() ? solver_row : c->inv[sd][(size_t)solver_row];
			uint64_t *out = dst[b] + (size_t)orig * un;
			if (c->cfg.word == 8) {
				memcpy(out, src + (size_t)q * np * 8, (size_t)un * 8);
			} else {
				const u32 *row = (const u32 *)(src + (size_t)q * np * 4);
				for (int l = 0; l < un; l++)
					out[l] = row[l];
			}
		}
	}
	if (iterations)
		*iterations = c->snap_iterations;
	c->snap_pending.store(false, std::memory_order_release);
	return BLZ_OK;
}

extern "C" int blz_sync(blz_ctx *c)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_sync: NULL context");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipStreamSynchronize(c->xstream));
	return BLZ_OK;
}

extern "C" int blz_set_exchange_mode(blz_ctx *c, int external)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_set_exchange_mode: NULL context");
	c->external_exchange = external != 0;
	return BLZ_OK;
}

extern "C" int blz_comm_unique_id(void *id_out, size_t id_bytes)
{
	if (!id_out || id_bytes < sizeof(ncclUniqueId))
		return blz_fail(BLZ_EINVAL, "blz_comm_unique_id: need %zu bytes", sizeof(ncclUniqueId));
	int rc = rccl_load();
	if (rc != BLZ_OK)
		return rc;
	ncclUniqueId id;
	NCCLCHK(g_rccl.GetUniqueId(&id));
	memcpy(id_out, &id, sizeof id);
	return BLZ_OK;
}

extern "C" int blz_comm_init(blz_ctx *c, const void *id, size_t id_bytes, int rank, int nranks)
{
	if (!c || !id || id_bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks)
		return blz_fail(BLZ_EINVAL, "blz_comm_init: bad argument");
	HIPCHK(hipSetDevice(c->device));
	int rc = rccl_load();
	if (rc != BLZ_OK)
		return rc;
	ncclUniqueId uid;
	memcpy(&uid, id, sizeof uid);
	NCCLCHK(g_rccl.CommInitRank(&c->comm, nranks, uid, rank));
	const char *f = getenv("BLZ_FORCE_COMM");
	c->force_comm = f && f[0] == '1';
	return BLZ_OK;
}

extern "C" int blz_comm_info(const blz_ctx *c, int *nranks_seen, int *rank_seen)
{
	if (!c)
		return blz_fail(BLZ_EINVAL, "blz_comm_info: NULL context");
	int cnt = -1, rk = -1;
	if (c->loop) {
		cnt = c->loop->nranks;
		rk = c->loop_rank;
	} else if (c->comm) {
		NCCLCHK(g_rccl.CommCount(c->comm, &cnt));
		NCCLCHK(g_rccl.CommUserRank(c->comm, &rk));
	}
	if (nranks_seen)
		*nranks_seen = cnt;
	if (rank_seen)
		*rank_seen = rk;
	return BLZ_OK;
}

extern "C" int blz_exchange_pieces(const blz_ctx *c, int transpose)
{
	if (!c || !c->have_matrix)
		return -1;
	const int t = transpose ? 1 : 0;
	return c->short_side[t] ? 0 : (int)c->csr[t].size();
}

extern "C" int blz_exchange_pieces_for(const blz_ctx *c, int64_t mrows, int64_t mcols, int64_t nnz, int nranks)
{
	if (!c || nranks < 1)
		return -1;
	return prep_params(c, mrows, mcols, nnz, nranks).K;
}

/* ---- loopback communicator: several contexts of one process on one device (tests of the multi-rank path on one GPU) ---- */

extern "C" int blz_loop_group_create(int nranks, blz_loop_group **out)
{
	if (!out || nranks < 1 || nranks > BLZ_LOOP_MAX_RANKS)
		return blz_fail(BLZ_EINVAL, "blz_loop_group_create: 1 .. %d ranks", BLZ_LOOP_MAX_RANKS);
	blz_loop_group *g = new blz_loop_group();
	g->nranks = nranks;
	if (const char *e = getenv("BLZ_LOOP_TIMEOUT_S"))
		if (atoi(e) >= 1)
			g->timeout_s = atoi(e);
	g->send.assign((size_t)nranks, nullptr);
	g->ready.assign((size_t)nranks, nullptr);
	g->done.assign((size_t)nranks, nullptr);
	*out = g;
	return BLZ_OK;
}

extern "C" void blz_loop_group_destroy(blz_loop_group *g)
{
	if (!g)
		return;
	(void)hipDeviceSynchronize();
	for (hipEvent_t e : g->ready)
		if (e) hipEventDestroy(e);
	for (hipEvent_t e : g->done)
		if (e) hipEventDestroy(e);
	delete g;
}

extern "C" int blz_comm_init_loopback(blz_ctx *c, blz_loop_group *g, int rank)
{
	if (!c || !g || rank < 0 || rank >= g->nranks)
		return blz_fail(BLZ_EINVAL, "blz_comm_init_loopback: bad argument");
	if (c->comm || c->loop)
		return blz_fail(BLZ_EINVAL, "blz_comm_init_loopback: the context has a communicator already");
	HIPCHK(hipSetDevice(c->device));
	{
		std::lock_guard<std::mutex> lk(g->mu);
		if (g->ready[(size_t)rank])
			return blz_fail(BLZ_EINVAL, "blz_comm_init_loopback: rank %d of the group is taken", rank);
		HIPCHK(hipEventCreateWithFlags(&g->ready[(size_t)rank], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&g->done[(size_t)rank], hipEventDisableTiming));
	}
	c->loop = g;
	c->loop_rank = rank;
	c->rank = rank;
	const char *f = getenv("BLZ_FORCE_COMM");
	c->force_comm = f && f[0] == '1';
	return BLZ_OK;
}
