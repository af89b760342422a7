// tools/ubench4.hip -- do gathers that HIT in L2 help a launch whose other gathers MISS?
//
// The block SpMV runs at ~55 G row gathers/s when (nearly) every gather misses L2, and a gather loop over an L2-resident
// table runs at ~176 G/s (profiles/r01_ubench_alu_and_gather.txt).  On renumbered matrices 50-70 % of the gathers hit
// (FETCH_SIZE, profiles/traffic_relat9_n1.json) and the kernels still run at ~55 G/s.  This measures a gather loop whose
// indices hit a small hot set (1 MB, resident in every L2) with probability h and a 125 MB table otherwise:
//   mixed      one launch, hot and cold indices interleaved at random
//   split      two launches: all the hot indices, then all the cold ones (same totals)
//   by-block   one launch, every workgroup gets only hot or only cold indices (stream sorted into contiguous runs)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench4 tools/ubench4.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint64_t u64;

template <int U>
__global__ void __launch_bounds__(256) k_gather(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, u64 *out)
{
	const int lane = threadIdx.x & 7;
	const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / 8, ng = (long long)gridDim.x * 32;
	u64 acc = 0;
	for (long long k = g * U; k + U <= count; k += ng * U) {
		int c[U];
		u64 x[U];
#pragma unroll
		for (int j = 0; j < U; j++) c[j] = idx[k + j];
#pragma unroll
		for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 8 + lane];
#pragma unroll
		for (int j = 0; j < U; j++) acc += x[j];
	}
	if (acc == 0x1234567) out[0] = acc;
}

// contiguous chunks per workgroup instead of a grid stride: workgroup b walks [b * per, (b + 1) * per)
template <int U>
__global__ void __launch_bounds__(256) k_gather_chunks(const u64 *__restrict__ table, const int *__restrict__ idx, long long count, u64 *out)
{
	const int lane = threadIdx.x & 7, grp = threadIdx.x / 8;
	const long long per = (count / gridDim.x) / (32 * U) * (32 * U);
	const long long lo = (long long)blockIdx.x * per, hi = lo + per;
	u64 acc = 0;
	for (long long k = lo + (long long)grp * U; k + U <= hi; k += 32 * U) {
		int c[U];
		u64 x[U];
#pragma unroll
		for (int j = 0; j < U; j++) c[j] = idx[k + j];
#pragma unroll
		for (int j = 0; j < U; j++) x[j] = table[(size_t)c[j] * 8 + lane];
#pragma unroll
		for (int j = 0; j < U; j++) acc += x[j];
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <typename F>
static double best_ms(F f)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	f();
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++) {
		CHK(hipEventRecord(e0));
		f();
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	return best;
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount, blocks = ncu * 8;
	const long long count = 40000000, rows = 125000000 / 64, hot_rows = (1 << 20) / 64;
	u64 *table, *out;
	int *idx, *idx2;
	CHK(hipMalloc(&table, rows * 64));
	CHK(hipMemset(table, 1, rows * 64));
	CHK(hipMalloc(&out, 4096));
	CHK(hipMalloc(&idx, count * 4));
	CHK(hipMalloc(&idx2, count * 4));
	std::vector<int> h(count), hs(count);
	printf("device %s, %d CUs; 40 M gathers of 64-byte rows, table 125 MB, hot set 1 MB\n", prop.name, ncu);
	for (int pct : { 0, 25, 50, 75, 90, 100 }) {
		uint64_t s = 88172645463325252ull;
		long long nhot = 0;
		std::vector<int> hot, cold;
		for (long long k = 0; k < count; k++) {
			s ^= s << 13; s ^= s >> 7; s ^= s << 17;
			const bool is_hot = (int)((s >> 40) % 100) < pct;
			const int v = is_hot ? (int)(s % (uint64_t)hot_rows) : (int)(hot_rows + s % (uint64_t)(rows - hot_rows));
			h[k] = v;
			(is_hot ? hot : cold).push_back(v);
			nhot += is_hot;
		}
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
		const double mixed = best_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, idx, count, out); });
		// split: hot indices first, then cold, as two launches over the two parts of one array
		std::copy(hot.begin(), hot.end(), hs.begin());
		std::copy(cold.begin(), cold.end(), hs.begin() + hot.size());
		CHK(hipMemcpy(idx2, hs.data(), count * 4, hipMemcpyHostToDevice));
		const long long nh = (long long)hot.size(), nc = (long long)cold.size();
		const double split = best_ms([&] {
			if (nh >= 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, idx2, nh, out);
			if (nc >= 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, idx2 + nh, nc, out);
		});
		// by block: one launch over the sorted stream in contiguous chunks (a workgroup sees hot only or cold only)
		const double byblock = best_ms([&] { hipLaunchKernelGGL(k_gather_chunks<8>, dim3(blocks), dim3(256), 0, 0, table, idx2, count, out); });
		const double chunks_mixed = best_ms([&] { hipLaunchKernelGGL(k_gather_chunks<8>, dim3(blocks), dim3(256), 0, 0, table, idx, count, out); });
		printf("hot %3d %%: mixed %7.3f ms (%6.1f G/s)   split in two launches %7.3f ms (%6.1f G/s)   sorted, chunk per workgroup %7.3f ms (%6.1f G/s)   mixed, chunk per workgroup %7.3f ms\n",
		       pct, mixed, count / mixed / 1e6, split, count / split / 1e6, byblock, count / byblock / 1e6, chunks_mixed);
		fflush(stdout);
	}
	return 0;
}
