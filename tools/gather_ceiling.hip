// tools/gather_ceiling.hip -- the block SpMV's ceiling ON THIS BOX, for bench.py to put beside the kernel's rate.
//
// The products are random gathers of whole block rows (64 bytes at n = 8, 128 bytes at n = 16) out of the operand block; what
// bounds them is the rate at which the fabric fills 128-byte lines (tools/ubench2, tools/ubench5), and that rate differs by
// up to 10 % from one MI355X box of the pool to the next (profiles/r03_*).  A ceiling read from a committed file therefore
// says little about the run it is printed with.  This program measures the bare loop -- index load, gather, add; nothing
// of the SpMV's arithmetic, row bookkeeping or matrix values -- for one row size, table size and row length, with and
// without the output-row stream (one row stored per LEN gathers), at 8 and at 16 bytes per lane, and prints ONE line of JSON.
//
//   tools/gather_ceiling <row bytes: 64|128> <table MB> <gathers per output row> [million gathers per launch = 40]
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_ceiling tools/gather_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint64_t u64;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

// RW words of 8 bytes per row; WPL words per lane (1: 8 bytes, 2: 16 bytes); U gathers in flight per lane
template <int RW, int WPL, int U, bool STORE>
__global__ void __launch_bounds__(256) k_gather(const u64 *__restrict__ table, const int *__restrict__ idx, long long nrows, int len,
						u64 *__restrict__ y, u64 *out)
{
	constexpr int LPR = RW / WPL;
	const int lane = threadIdx.x & (LPR - 1);
	const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / LPR, ng = (long long)gridDim.x * (256 / LPR);
	u64 sink = 0;
	for (long long r = g; r < nrows; r += ng) {
		u64 a0 = 0, a1 = 0;
		const long long k0 = r * len;
		for (int k = 0; k < len; k += U) {
			int c[U];
#pragma unroll
			for (int j = 0; j < U; j++) c[j] = idx[k0 + (k + j < len ? k + j : len - 1)];
			if (WPL == 1) {
				u64 x[U];
#pragma unroll
				for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * RW + lane];
#pragma unroll
				for (int j = 0; j < U; j++) a0 += k + j < len ? x[j] : 0;
			} else {
				ull2 x[U];
#pragma unroll
				for (int j = 0; j < U; j++) x[j] = *(const ull2 *)(table + (size_t)c[j] * RW + 2 * lane);
#pragma unroll
				for (int j = 0; j < U; j++) {
					a0 += k + j < len ? x[j].x : 0;
					a1 += k + j < len ? x[j].y : 0;
				}
			}
		}
		if (!STORE) {
			sink += a0 + a1;
		} else if (WPL == 1) {
			y[(size_t)r * RW + lane] = a0;
		} else {
			ull2 v = { a0, a1 };
			*(ull2 *)(y + (size_t)r * RW + 2 * lane) = v;
		}
	}
	if (sink == 0x1234567) out[0] = sink;
}

template <int RW, int WPL, bool STORE>
static double rate(const u64 *table, const int *idx, long long nrows, int len, u64 *y, u64 *out, int ncu)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0));
	CHK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; rep++) {		// the first launch warms the TLB
		CHK(hipEventRecord(e0, 0));
		k_gather<RW, WPL, 8, STORE><<<ncu * 8, 256>>>(table, idx, nrows, len, y, out);
		CHK(hipEventRecord(e1, 0));
		CHK(hipEventSynchronize(e1));
		float ms = 0;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (rep && ms < best) best = ms;
	}
	CHK(hipEventDestroy(e0));
	CHK(hipEventDestroy(e1));
	return (double)nrows * len / (best * 1e-3);
}

template <int RW>
static void measure(double mb, int len, long long count, int ncu, int row_bytes)
{
	const long long rows = (long long)(mb * 1e6 / row_bytes), nrows = count / len;
	u64 *table, *y, *out;
	int *idx;
	CHK(hipMalloc(&table, (size_t)rows * row_bytes));
	CHK(hipMemset(table, 1, (size_t)rows * row_bytes));
	/* GC_Y_ALLOC=uncached|fine (experiment, profiles/r03_gather_ceiling_output_alloc.txt): the output rows in memory that the
	 * L2 does not hold dirty -- does the write stream cost less when it never waits for an eviction? */
	const char *ya = getenv("GC_Y_ALLOC");
	if (ya && ya[0] == 'u')
		CHK(hipExtMallocWithFlags((void **)&y, (size_t)(nrows + 1) * row_bytes, hipDeviceMallocUncached));
	else if (ya && ya[0] == 'f')
		CHK(hipExtMallocWithFlags((void **)&y, (size_t)(nrows + 1) * row_bytes, hipDeviceMallocFinegrained));
	else
		CHK(hipMalloc(&y, (size_t)(nrows + 1) * row_bytes));
	CHK(hipMalloc(&out, 4096));
	CHK(hipMalloc(&idx, (size_t)nrows * len * 4));
	{
		std::vector<int> h((size_t)nrows * len);
		uint64_t s = 88172645463325252ull;
		for (size_t k = 0; k < h.size(); k++) {
			s ^= s << 13; s ^= s >> 7; s ^= s << 17;
			h[k] = (int)(s % (uint64_t)rows);
		}
		CHK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	}
	const double b8 = rate<RW, 1, false>(table, idx, nrows, len, y, out, ncu);
	const double b16 = rate<RW, 2, false>(table, idx, nrows, len, y, out, ncu);
	const double s8 = rate<RW, 1, true>(table, idx, nrows, len, y, out, ncu);
	const double s16 = rate<RW, 2, true>(table, idx, nrows, len, y, out, ncu);
	printf("{\"row_bytes\": %d, \"table_mb\": %.1f, \"gathers_per_output_row\": %d, \"gathers_per_launch\": %lld, "
	       "\"bare_8B_per_lane\": %.4g, \"bare_16B_per_lane\": %.4g, \"with_output_rows_8B_per_lane\": %.4g, "
	       "\"with_output_rows_16B_per_lane\": %.4g, \"unit\": \"gathers/s\", \"workgroups_per_cu\": 8, \"in_flight_per_lane\": 8}\n",
	       row_bytes, mb, len, nrows * len, b8, b16, s8, s16);
}

int main(int argc, char **argv)
{
	if (argc < 4) {
		fprintf(stderr, "usage: gather_ceiling <row bytes: 64|128> <table MB> <gathers per output row> [million gathers = 40]\n");
		return 2;
	}
	const int row_bytes = atoi(argv[1]);
	const double mb = atof(argv[2]);
	int len = atoi(argv[3]);
	const long long count = (long long)((argc > 4 ? atof(argv[4]) : 40.0) * 1e6);
	if ((row_bytes != 64 && row_bytes != 128) || mb < 1.0 || mb > 65536.0 || len < 1 || count < 1000000 || count > 2000000000ll) {
		fprintf(stderr, "gather_ceiling: arguments out of range\n");
		return 2;
	}
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	if (row_bytes == 64)
		measure<8>(mb, len, count, prop.multiProcessorCount, row_bytes);
	else
		measure<16>(mb, len, count, prop.multiProcessorCount, row_bytes);
	return 0;
}
