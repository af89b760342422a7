// tools/ubench5.hip -- round 3: what separates the n = 16 block SpMV from the bare gather loop (config-5 shape).
//
// Round 2 left the staged SpMV at 40-44.6 G gathers/s on the config-5 quarter shape (128-byte block rows out of a 1.6 GB
// operand) against 48 G/s for tools/ubench2's bare loop at that table size; round 3's dynamic-rows form (all four lane
// groups of a wavefront busy in every round) made it SLOWER, so idle lane groups are not the gap.  This file adds to the
// bare loop, one at a time, what the kernel does and the loop does not:
//   v0  the bare loop (8 bytes per lane, 16 lanes per row, U gathers in flight per lane)
//   v1  16 bytes per lane, 8 lanes per row (half the vector-memory instructions per row)
//   v2  v0 + the output stream: after every LEN gathers the group stores its 128-byte sum row (streaming, 1 row per LEN)
//   v3  v2 with non-temporal stores
//   v4  v0 with the wavefront draining its gathers after every batch (s_waitcnt vmcnt(0)): the kernel's batch structure
//   v5  v2 + v4 (what the kernel does), and the same at fewer workgroups per CU (the kernel's 19 resident wavefronts)
//   v6  the same rows written by OTHER workgroups (one in eight only writes, seven only gather): same traffic, but no
//       wavefront has a store among its gathers -- separates "the memory system dislikes the mix" from "a store stalls
//       the wavefront that issued it" (vmcnt retires in issue order: the wait for a batch of gathers also waits for the
//       acknowledgement of every older store)
//   v7  v2 with the stores of KEEP finished rows issued together (sums parked in registers): same bytes, 1/KEEP of the stalls
//   v8  v2 with the finished row handed to LDS and written by one dedicated wavefront of the workgroup
//   v9  every wavefront owns a CONTIGUOUS range of output rows, parks CH finished rows in LDS and writes them as one burst
//       of CH x 128 contiguous bytes, 16 bytes per lane (1 KB per store instruction): does the write stream cost less in
//       larger contiguous pieces?  (v2 at 16 bytes per lane -- 1 KB per store instead of 512 bytes -- lost 11 % to the
//       stores where 8 bytes per lane lost 19 %.)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench5 tools/ubench5.hip ; run: tools/ubench5 [table MB] [M gathers]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

enum { STORE_NONE = 0, STORE_PLAIN = 1, STORE_NT = 2 };

// LPR lanes per 128-byte row (16: 8 bytes per lane, 8: 16 bytes per lane); LEN gathers per output row; DRAIN: wait for the
// whole batch before the adds (the compiler otherwise waits slot by slot)
template <int U, int LPR, int STORE, bool DRAIN>
__global__ void __launch_bounds__(256) k_rows(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, int len,
					      u64 *__restrict__ y, u64 *out)
{
	const int lane = threadIdx.x & (LPR - 1);
	const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / LPR, ng = (long long)gridDim.x * (256 / LPR);
	const long long nrows = count / len;
	u64 sink = 0;
	for (long long r = g; r < nrows; r += ng) {
		u64 a0 = 0, a1 = 0;
		const long long k0 = r * len;
		for (int k = 0; k < len; k += U) {
			int c[U];
#pragma unroll
			for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
			if (LPR == 16) {
				u64 x[U];
#pragma unroll
				for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 16 + lane];
				if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
				for (int j = 0; j < U; j++) a0 += k + j < len ? x[j] : 0;
			} else {
				ull2 x[U];
#pragma unroll
				for (int j = 0; j < U; j++) x[j] = *(const ull2 *)(table + (size_t)c[j] * 16 + 2 * lane);
				if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
				for (int j = 0; j < U; j++) {
					a0 += k + j < len ? x[j].x : 0;
					a1 += k + j < len ? x[j].y : 0;
				}
			}
		}
		if (STORE == STORE_NONE) {
			sink += a0 + a1;
		} else if (LPR == 16) {
			if (STORE == STORE_NT) __builtin_nontemporal_store(a0, &y[(size_t)r * 16 + lane]);
			else y[(size_t)r * 16 + lane] = a0;
		} else {
			ull2 v = { a0, a1 };
			*(ull2 *)(y + (size_t)r * 16 + 2 * lane) = v;
		}
	}
	if (sink == 0x1234567) out[0] = sink;
}

// v6: one workgroup in eight streams the output rows (all of them), the others gather without storing
template <int U>
__global__ void __launch_bounds__(256) k_split_roles(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, int len,
						    u64 *__restrict__ y, u64 *out)
{
	const long long nrows = count / len;
	if ((blockIdx.x & 7) == 7) {
		const long long w = (long long)(blockIdx.x >> 3) * 256 + threadIdx.x, nw = (long long)((gridDim.x + 7) >> 3) * 256;
		for (long long i = w; i < nrows * 16; i += nw)
			y[i] = (u64)i;
		return;
	}
	const int lane = threadIdx.x & 15;
	const long long b = (long long)blockIdx.x - (blockIdx.x >> 3), nb = (long long)gridDim.x - (gridDim.x >> 3);
	const long long g = (b * 256 + threadIdx.x) / 16, ng = nb * 16;
	u64 sink = 0;
	for (long long r = g; r < nrows; r += ng) {
		const long long k0 = r * len;
		for (int k = 0; k < len; k += U) {
			int c[U];
			u64 x[U];
#pragma unroll
			for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
#pragma unroll
			for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 16 + lane];
#pragma unroll
			for (int j = 0; j < U; j++) sink += k + j < len ? x[j] : 0;
		}
	}
	if (sink == 0x1234567) out[0] = sink;
}

// v7: the sums of KEEP finished rows wait in registers and are stored back to back
template <int U, int KEEP>
__global__ void __launch_bounds__(256) k_batched_stores(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, int len,
						       u64 *__restrict__ y, u64 *out)
{
	const int lane = threadIdx.x & 15;
	const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / 16, ng = (long long)gridDim.x * 16;
	const long long nrows = count / len;
	u64 keep[KEEP];
	long long at[KEEP];
	int have = 0;
	for (long long r = g; r < nrows; r += ng) {
		u64 a0 = 0;
		const long long k0 = r * len;
		for (int k = 0; k < len; k += U) {
			int c[U];
			u64 x[U];
#pragma unroll
			for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
#pragma unroll
			for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 16 + lane];
#pragma unroll
			for (int j = 0; j < U; j++) a0 += k + j < len ? x[j] : 0;
		}
#pragma unroll
		for (int q = 0; q < KEEP; q++)
			if (q == have) {
				keep[q] = a0;
				at[q] = r;
			}
		if (++have == KEEP) {
#pragma unroll
			for (int q = 0; q < KEEP; q++)
				y[(size_t)at[q] * 16 + lane] = keep[q];
			have = 0;
		}
	}
#pragma unroll
	for (int q = 0; q < KEEP; q++)
		if (q < have)
			y[(size_t)at[q] * 16 + lane] = keep[q];
	if (have == 77) out[0] = 1;
}

// v8: three wavefronts of the workgroup gather; a finished row goes to an LDS ring and the fourth wavefront writes it out
template <int U>
__global__ void __launch_bounds__(256) k_writer_wave(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, int len,
						    u64 *__restrict__ y, u64 *out)
{
	constexpr int SLOTS = 32;		/* per gathering wavefront: 32 rows of 128 bytes */
	__shared__ u64 ring[3][SLOTS][16];
	__shared__ long long ring_row[3][SLOTS];
	__shared__ volatile int head[3], tail[3], done[3];
	const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, lane = wl & 15, grp = wl >> 4;
	if (threadIdx.x < 3) {
		head[threadIdx.x] = 0;
		tail[threadIdx.x] = 0;
		done[threadIdx.x] = 0;
	}
	__syncthreads();
	const long long nrows = count / len;
	if (wave == 3) {
		/* the writer: drains the three rings until their owners are done */
		int t[3] = { 0, 0, 0 };
		for (long long turns = 0; turns < (1ll << 28); turns++) {	/* bounded: a benchmark must not hang */
			bool any = false, all_done = true;
			for (int w = 0; w < 3; w++) {
				const int h = head[w];
				all_done = all_done && done[w] && t[w] == h;
				while (t[w] != h) {
					/* four rows per instruction: lane group q takes slot t + q */
					const int s_ = t[w] + grp;
					if (s_ - h < 0) {
						const long long r = ring_row[w][s_ % SLOTS];
						y[(size_t)r * 16 + lane] = ring[w][s_ % SLOTS][lane];
					}
					t[w] = h - t[w] < 4 ? h : t[w] + 4;
					any = true;
				}
				if (lane == 0 && grp == 0)
					tail[w] = t[w];
			}
			if (all_done)
				break;
			if (!any)
				__builtin_amdgcn_s_sleep(8);
		}
		return;
	}
	const long long g = ((long long)blockIdx.x * 3 + wave) * 4 + grp, ng = (long long)gridDim.x * 12;
	int h = 0;
	for (long long r = g; r < nrows; r += ng) {
		u64 a0 = 0;
		const long long k0 = r * len;
		for (int k = 0; k < len; k += U) {
			int c[U];
			u64 x[U];
#pragma unroll
			for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
#pragma unroll
			for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 16 + lane];
#pragma unroll
			for (int j = 0; j < U; j++) a0 += k + j < len ? x[j] : 0;
		}
		/* the wavefront's four groups finish a row each, in lockstep: four slots at once */
		for (int spins = 0; h + 4 - tail[wave] > SLOTS && spins < (1 << 22); spins++)	/* bounded: a benchmark must not hang */
			__builtin_amdgcn_s_sleep(2);
		ring[wave][(h + grp) % SLOTS][lane] = a0;
		if (lane == 0)
			ring_row[wave][(h + grp) % SLOTS] = r;
		h += 4;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (wl == 0)
			head[wave] = h;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	if (wl == 0)
		done[wave] = 1;
	if (h == 77) out[0] = 1;
}

// v9: contiguous rows per wavefront, CH finished rows parked in LDS, written as one contiguous burst
template <int U, int CH>
__global__ void __launch_bounds__(256) k_burst_rows(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, int len,
						   u64 *__restrict__ y, u64 *out)
{
	__shared__ __attribute__((aligned(16))) u64 park[4][CH][16];
	const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, lane = wl & 15, grp = wl >> 4;
	const long long nrows = count / len, nw = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + wave;
	long long per = (nrows + nw - 1) / nw;
	per = (per + CH - 1) / CH * CH;
	const long long lo = w * per, hi = lo + per < nrows ? lo + per : nrows;
	for (long long base = lo; base < hi; base += CH) {
		for (int q = 0; q < CH; q += 4) {
			const long long r = base + q + grp;
			u64 a0 = 0;
			if (r < hi) {
				const long long k0 = r * len;
				for (int k = 0; k < len; k += U) {
					int c[U];
					u64 x[U];
#pragma unroll
					for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
#pragma unroll
					for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 16 + lane];
#pragma unroll
					for (int j = 0; j < U; j++) a0 += k + j < len ? x[j] : 0;
				}
			}
			park[wave][q + grp][lane] = a0;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		/* CH rows x 128 bytes, 16 bytes per lane: lane l of store number s takes bytes 16 (64 s + l) .. of the burst */
#pragma unroll
		for (int s_ = 0; s_ < CH / 8; s_++) {
			const int at = 64 * s_ + wl, row = at >> 3, part = at & 7;
			if (base + row < hi) {
				const ull2 v = *(const ull2 *)&park[wave][row][2 * part];
				*(ull2 *)(y + (size_t)(base + row) * 16 + 2 * part) = v;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	if (hi == -77) out[0] = 1;
}

template <int U, int LPR, int STORE, bool DRAIN>
static void run(const char *name, const u64 *table, const int *idx, long long count, int len, u64 *y, u64 *out, int ncu, int per_cu)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int blocks = ncu * per_cu;
	hipLaunchKernelGGL((k_rows<U, LPR, STORE, DRAIN>), dim3(blocks), dim3(256), 0, 0, table, idx, count, len, y, out);
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_rows<U, LPR, STORE, DRAIN>), dim3(blocks), dim3(256), 0, 0, table, idx, count, len, y, out);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("%-58s U=%d  %d B/lane  blocks/CU %d : %8.3f ms  %6.1f G rows/s\n", name, U, 128 / LPR, per_cu, best, count / best / 1e6);
	fflush(stdout);
}

template <class K>
static void run_k(const char *name, K kern, const u64 *table, const int *idx, long long count, int len, u64 *y, u64 *out, int ncu, int per_cu)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int blocks = ncu * per_cu;
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, idx, count, len, y, out);
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, idx, count, len, y, out);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("%-58s U=8  8 B/lane  blocks/CU %d : %8.3f ms  %6.1f G rows/s\n", name, per_cu, best, count / best / 1e6);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const double mb = argc > 1 ? atof(argv[1]) : 1600.0;
	const long long count = (long long)((argc > 2 ? atof(argv[2]) : 200.0) * 1e6);
	const int len = 40;
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	printf("device %s, %d CUs; 128-byte rows, table %.0f MB, %lld gathers per launch, one output row per %d gathers\n", prop.name, ncu, mb,
	       count, len);
	const long long rows = (long long)(mb * 1e6 / 128);
	u64 *table, *y, *out;
	int *idx;
	CHK(hipMalloc(&table, (size_t)rows * 128));
	CHK(hipMemset(table, 1, (size_t)rows * 128));
	CHK(hipMalloc(&y, (size_t)(count / len + 1) * 128));
	CHK(hipMalloc(&out, 4096));
	CHK(hipMalloc(&idx, count * 4));
	{
		std::vector<int> h(count);
		uint64_t s = 88172645463325252ull;
		for (long long k = 0; k < count; k++) {
			s ^= s << 13; s ^= s >> 7; s ^= s << 17;
			h[k] = (int)(s % (uint64_t)rows);
		}
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
	}
	run<8, 16, STORE_NONE, false>("v0 bare loop", table, idx, count, len, y, out, ncu, 8);
	run<4, 16, STORE_NONE, false>("v0 bare loop", table, idx, count, len, y, out, ncu, 8);
	run<8, 8, STORE_NONE, false>("v1 16 bytes per lane", table, idx, count, len, y, out, ncu, 8);
	run<4, 8, STORE_NONE, false>("v1 16 bytes per lane", table, idx, count, len, y, out, ncu, 8);
	run<8, 16, STORE_PLAIN, false>("v2 + output rows", table, idx, count, len, y, out, ncu, 8);
	run<8, 16, STORE_NT, false>("v3 + output rows, non-temporal", table, idx, count, len, y, out, ncu, 8);
	run<8, 8, STORE_PLAIN, false>("v2 + output rows, 16 bytes per lane", table, idx, count, len, y, out, ncu, 8);
	run<8, 16, STORE_NONE, true>("v4 batch drained before the adds", table, idx, count, len, y, out, ncu, 8);
	run<8, 16, STORE_PLAIN, true>("v5 output rows + drained batches", table, idx, count, len, y, out, ncu, 8);
	run<8, 16, STORE_PLAIN, true>("v5 output rows + drained batches", table, idx, count, len, y, out, ncu, 5);
	run<8, 16, STORE_PLAIN, true>("v5 output rows + drained batches", table, idx, count, len, y, out, ncu, 4);
	run<8, 8, STORE_PLAIN, true>("v5 output rows + drained batches, 16 bytes per lane", table, idx, count, len, y, out, ncu, 8);
	run<8, 8, STORE_PLAIN, true>("v5 output rows + drained batches, 16 bytes per lane", table, idx, count, len, y, out, ncu, 5);
	run_k("v6 rows written by other workgroups (1 in 8)", k_split_roles<8>, table, idx, count, len, y, out, ncu, 8);
	run_k("v7 stores of 2 rows together", k_batched_stores<8, 2>, table, idx, count, len, y, out, ncu, 8);
	run_k("v7 stores of 4 rows together", k_batched_stores<8, 4>, table, idx, count, len, y, out, ncu, 8);
	run_k("v7 stores of 8 rows together", k_batched_stores<8, 8>, table, idx, count, len, y, out, ncu, 8);
	run_k("v8 rows through LDS to a writer wavefront", k_writer_wave<8>, table, idx, count, len, y, out, ncu, 8);
	run_k("v9 contiguous rows per wavefront, bursts of 8 rows (1 KB)", k_burst_rows<8, 8>, table, idx, count, len, y, out, ncu, 8);
	run_k("v9 contiguous rows per wavefront, bursts of 16 rows (2 KB)", k_burst_rows<8, 16>, table, idx, count, len, y, out, ncu, 8);
	run_k("v9 contiguous rows per wavefront, bursts of 32 rows (4 KB)", k_burst_rows<8, 32>, table, idx, count, len, y, out, ncu, 8);
	run_k("v9 contiguous rows per wavefront, bursts of 64 rows (8 KB)", k_burst_rows<8, 64>, table, idx, count, len, y, out, ncu, 8);
	return 0;
}
