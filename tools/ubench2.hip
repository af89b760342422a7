// tools/ubench2.hip -- round 2 micro-benchmarks for the block SpMV's gather (run on the GPU box).
//
// Question 1: the SpMV is bound by ~55 G row gathers per second, each tallied as one
//   128-byte line fill.  Is that a byte bound (7 TB/s of lines) or a request bound?  If the L2 -> fabric path can be
//   made to issue 64-byte (or 32-byte) requests -- memory the L2 does not cache (hipDeviceMallocUncached /
//   hipDeviceMallocFinegrained), or loads with the sc0 sc1 bits -- a 64-byte block row (n = 8, u64) would cost half a line.
//   Every variant is a kernel of its own name, so a `rocprofv3 --pmc TCC_EA0_RDREQ_32B/_64B/_128B` pass over this
//   binary attributes the request sizes.
// Question 2 (item 2): config 5 gathers 128-byte rows out of a 6.4 GB block and runs at 35-41 G gathers/s, not 55.
//   Sweep the table size with 128-byte rows (TLB reach?), with uniformly random indices and with indices confined to a
//   window that moves slowly over the table (what a column-window ordering of the entries would produce).
// Question 3: what an LDS-resident panel of hot block rows is worth (hot columns of a heavy-tailed matrix): gathers of
//   which a fraction f hits a panel staged once per workgroup in LDS.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench2 tools/ubench2.hip ; run: tools/ubench2 [q1] [q2] [q3]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint64_t u64;
typedef uint32_t u32;

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3 };
enum { A_DEFAULT = 0, A_FINE = 1, A_UNCACHED = 2 };

template <int POLICY>
__device__ __forceinline__ u64 ld(const u64 *p)
{
	if (POLICY == NT)
		return __builtin_nontemporal_load(p);
	if (POLICY == SC1)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (POLICY == SC0SC1)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	return *p;
}

// A group of RW lanes gathers random rows of RW 8-byte words; U independent loads in flight per lane.
// ALLOC only names the kernel after the kind of memory the table was allocated in.
template <int U, int RW, int POLICY, int ALLOC>
__global__ void __launch_bounds__(256) k_gather(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, u64 *out)
{
	const int lane = threadIdx.x & (RW - 1);
	const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / RW, ng = (long long)gridDim.x * (256 / RW);
	u64 acc = 0;
	for (long long k = g * U; k + U <= count; k += ng * U) {
		int c[U];
		u64 x[U];
#pragma unroll
		for (int j = 0; j < U; j++) c[j] = idx[k + j];
#pragma unroll
		for (int j = 0; j < U; j++) x[j] = ld<POLICY>(&table[(size_t)c[j] * RW + lane]);
#pragma unroll
		for (int j = 0; j < U; j++) acc += x[j];
	}
	if (acc == 0x1234567) out[0] = acc;
}

static const char *policy_name[] = { "plain", "nt", "sc1", "sc0sc1" };
static const char *alloc_name[] = { "hipMalloc", "finegrained", "uncached" };

static void *alloc_table(int kind, size_t bytes)
{
	void *p = nullptr;
	hipError_t e;
	if (kind == A_DEFAULT)
		e = hipMalloc(&p, bytes);
	else
		e = hipExtMallocWithFlags(&p, bytes, kind == A_FINE ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
	if (e != hipSuccess) {
		printf("  (%s allocation of %.1f MB failed: %s)\n", alloc_name[kind], bytes / 1e6, hipGetErrorString(e));
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}

template <int U, int RW, int POLICY, int ALLOC>
static double run_gather(const u64 *table, const int *idx, long long count, u64 *out, int ncu, double table_mb, int blocks_per_cu, const char *note)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int blocks = ncu * blocks_per_cu;
	hipLaunchKernelGGL((k_gather<U, RW, POLICY, ALLOC>), dim3(blocks), dim3(256), 0, 0, table, idx, count, out);
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_gather<U, RW, POLICY, ALLOC>), dim3(blocks), dim3(256), 0, 0, table, idx, count, out);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("gather %-11s %-6s %3dB rows  table %8.1f MB  U=%2d  blocks/CU %d : %8.3f ms  %7.1f GB/s useful  %6.1f G rows/s  %s\n",
	       alloc_name[ALLOC], policy_name[POLICY], RW * 8, table_mb, U, blocks_per_cu, best, count * 8.0 * RW / best / 1e6,
	       count / best / 1e6, note);
	fflush(stdout);
	return count / best / 1e6;
}

static void fill_random(std::vector<int> &h, long long count, long long rows, uint64_t seed)
{
	uint64_t s = seed;
	for (long long k = 0; k < count; k++) {
		s ^= s << 13; s ^= s >> 7; s ^= s << 17;
		h[k] = (int)(s % (uint64_t)rows);
	}
}

// indices confined to a window of `win` rows whose base sweeps the table once over the whole index stream
static void fill_windowed(std::vector<int> &h, long long count, long long rows, long long win, uint64_t seed)
{
	uint64_t s = seed;
	if (win > rows) win = rows;
	for (long long k = 0; k < count; k++) {
		s ^= s << 13; s ^= s >> 7; s ^= s << 17;
		const long long base = (long long)((double)k / (double)count * (double)(rows - win));
		h[k] = (int)(base + (long long)(s % (uint64_t)win));
	}
}

template <int RW, int ALLOC>
static void q1_alloc(int ncu, const int *idx, long long count, u64 *out, double mb)
{
	const long long rows = (long long)(mb * 1e6 / (8 * RW));
	u64 *table = (u64 *)alloc_table(ALLOC, (size_t)rows * 8 * RW);
	if (!table) return;
	CHK(hipMemset(table, 1, (size_t)rows * 8 * RW));
	CHK(hipDeviceSynchronize());
	run_gather<8, RW, PLAIN, ALLOC>(table, idx, count, out, ncu, mb, 8, "");
	run_gather<8, RW, SC1, ALLOC>(table, idx, count, out, ncu, mb, 8, "");
	run_gather<8, RW, SC0SC1, ALLOC>(table, idx, count, out, ncu, mb, 8, "");
	if (ALLOC == A_DEFAULT)
		run_gather<8, RW, NT, ALLOC>(table, idx, count, out, ncu, mb, 8, "");
	CHK(hipFree(table));
}

// ------------------------------------------------------------------------------------------------ question 3
// Entries: low 31 bits = row index, top bit set = the row is in the hot panel (index into LDS).  The panel (HOT rows of
// RW words) is copied into LDS once per workgroup; one workgroup of 1024 threads per CU.
template <int U, int RW>
__global__ void __launch_bounds__(1024) k_gather_panel(const u64 *__restrict__ table, const u32 *__restrict__ idx, long long count,
						      int hot_rows, u64 *out)
{
	extern __shared__ u64 panel[];
	for (int i = threadIdx.x; i < hot_rows * RW; i += 1024)
		panel[i] = table[i];
	__syncthreads();
	const int lane = threadIdx.x & (RW - 1);
	const long long g = ((long long)blockIdx.x * 1024 + threadIdx.x) / RW, ng = (long long)gridDim.x * (1024 / RW);
	u64 acc = 0;
	for (long long k = g * U; k + U <= count; k += ng * U) {
		u32 c[U];
		u64 x[U];
#pragma unroll
		for (int j = 0; j < U; j++) c[j] = idx[k + j];
#pragma unroll
		for (int j = 0; j < U; j++) {
			if (c[j] & 0x80000000u)
				x[j] = panel[(size_t)(c[j] & 0x7FFFFFFFu) * RW + lane];
			else
				x[j] = table[(size_t)c[j] * RW + lane];
		}
#pragma unroll
		for (int j = 0; j < U; j++) acc += x[j];
	}
	if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv)
{
	bool q1 = argc == 1, q2 = argc == 1, q3 = argc == 1;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "q1")) q1 = true;
		if (!strcmp(argv[i], "q2")) q2 = true;
		if (!strcmp(argv[i], "q3")) q3 = true;
	}
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.name, ncu);
	u64 *out;
	CHK(hipMalloc(&out, 4096));
	const long long count = 40000000;
	std::vector<int> h(count);
	int *idx;
	CHK(hipMalloc(&idx, count * 4));

	if (q1) {
		printf("== Q1: request size by allocation kind and load policy, 125 MB table (the GL7d19 block) ==\n");
		const double mb = 125.0;
		fill_random(h, count, (long long)(mb * 1e6 / 64), 88172645463325252ull);
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
		q1_alloc<8, A_DEFAULT>(ncu, idx, count, out, mb);
		q1_alloc<8, A_FINE>(ncu, idx, count, out, mb);
		q1_alloc<8, A_UNCACHED>(ncu, idx, count, out, mb);
		fill_random(h, count, (long long)(mb * 1e6 / 32), 88172645463325252ull);
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
		q1_alloc<4, A_DEFAULT>(ncu, idx, count, out, mb);
		q1_alloc<4, A_FINE>(ncu, idx, count, out, mb);
		q1_alloc<4, A_UNCACHED>(ncu, idx, count, out, mb);
		fill_random(h, count, (long long)(mb * 1e6 / 128), 88172645463325252ull);
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
		q1_alloc<16, A_DEFAULT>(ncu, idx, count / 2, out, mb);
		q1_alloc<16, A_FINE>(ncu, idx, count / 2, out, mb);
		q1_alloc<16, A_UNCACHED>(ncu, idx, count / 2, out, mb);
	}

	if (q2) {
		printf("== Q2: 128-byte rows, table size sweep (config 5 gathers from 6.4 GB), uniform and windowed indices ==\n");
		const double sizes_mb[] = { 125.0, 800.0, 1600.0, 3200.0, 6400.0, 12800.0 };
		for (double mb : sizes_mb) {
			const long long rows = (long long)(mb * 1e6 / 128);
			u64 *table = (u64 *)alloc_table(A_DEFAULT, (size_t)rows * 128);
			if (!table) continue;
			CHK(hipMemset(table, 1, (size_t)rows * 128));
			fill_random(h, count, rows, 88172645463325252ull);
			CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
			run_gather<4, 16, PLAIN, A_DEFAULT>(table, idx, count, out, ncu, mb, 8, "uniform");
			if (mb >= 3200.0) {
				run_gather<8, 16, PLAIN, A_DEFAULT>(table, idx, count, out, ncu, mb, 8, "uniform");
				run_gather<4, 16, PLAIN, A_DEFAULT>(table, idx, count, out, ncu, mb, 4, "uniform");
				const long long wins[] = { 800, 200 };
				for (long long wmb : wins) {
					char note[64];
					snprintf(note, sizeof note, "window %lld MB", wmb);
					fill_windowed(h, count, rows, wmb * 1000000 / 128, 88172645463325252ull);
					CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
					run_gather<4, 16, PLAIN, A_DEFAULT>(table, idx, count, out, ncu, mb, 8, note);
				}
			}
			CHK(hipFree(table));
		}
	}

	if (q3) {
		printf("== Q3: a fraction f of the gathers hits a panel of hot 64-byte rows staged in LDS (1 workgroup of 1024 per CU) ==\n");
		const double mb = 125.0;
		const long long rows = (long long)(mb * 1e6 / 64);
		u64 *table = (u64 *)alloc_table(A_DEFAULT, (size_t)rows * 64);
		CHK(hipMemset(table, 1, (size_t)rows * 64));
		const int hot_rows = 2048;	// 128 KB of LDS
		const double fr[] = { 0.0, 0.25, 0.5, 0.75 };
		std::vector<u32> hu(count);
		for (double f : fr) {
			uint64_t s = 88172645463325252ull;
			for (long long k = 0; k < count; k++) {
				s ^= s << 13; s ^= s >> 7; s ^= s << 17;
				const bool hot = (double)(s >> 40) / (double)(1 << 24) < f;
				const u32 r = (u32)((s & 0xFFFFFFFFFFull) % (uint64_t)(hot ? hot_rows : rows));
				hu[k] = r | (hot ? 0x80000000u : 0u);
			}
			CHK(hipMemcpy(idx, hu.data(), count * 4, hipMemcpyHostToDevice));
			hipEvent_t e0, e1;
			CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
			CHK(hipFuncSetAttribute((const void *)k_gather_panel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, hot_rows * 64));
			hipLaunchKernelGGL((k_gather_panel<4, 8>), dim3(ncu), dim3(1024), hot_rows * 64, 0, table, (const u32 *)idx, count, hot_rows, out);
			CHK(hipDeviceSynchronize());
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL((k_gather_panel<4, 8>), dim3(ncu), dim3(1024), hot_rows * 64, 0, table, (const u32 *)idx, count, hot_rows, out);
			CHK(hipEventRecord(e1));
			CHK(hipEventSynchronize(e1));
			float ms;
			CHK(hipEventElapsedTime(&ms, e0, e1));
			printf("panel f=%.2f : %8.3f ms  %6.1f G rows/s (cold gathers alone at 55 G/s would take %.3f ms)\n", f, ms, count / ms / 1e6,
			       count * (1.0 - f) / 55e6);
			fflush(stdout);
		}
		CHK(hipFree(table));
	}
	return 0;
}
