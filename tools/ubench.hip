// tools/ubench.hip -- micro-benchmarks that set the design constants of the kernels (run on the GPU box):
//   1. issue rate of the integer instructions the mod-p MAC is made of (v_mad_u64_u32, v_mul_lo/hi_u32, adds, f64 fma)
//   2. random 64-byte-row gather bandwidth as a function of table size and loads in flight (what bounds the block SpMV)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run: tools/ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint64_t u64;
typedef uint32_t u32;

template <int OP>
__global__ void __launch_bounds__(256) k_alu(u64 *out, u32 a0, u32 b0, int iters)
{
	u64 acc[8];
	u32 a = a0 + threadIdx.x, b = b0 + blockIdx.x;
	double fa = a, fb = b;
	double facc[8];
#pragma unroll
	for (int j = 0; j < 8; j++) { acc[j] = j + threadIdx.x; facc[j] = j; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (OP == 0) acc[j] = (u64)a * (u32)(b + j) + acc[j];			// v_mad_u64_u32
			if (OP == 1) acc[j] = (u32)acc[j] * a + j;				// v_mul_lo_u32
			if (OP == 2) acc[j] = __umulhi((u32)acc[j], a) + j;			// v_mul_hi_u32
			if (OP == 3) acc[j] = acc[j] + ((u64)a << 7) + b;			// 64-bit adds
			if (OP == 4) facc[j] = fma(facc[j], fa, fb);				// v_fma_f64
			if (OP == 5) acc[j] = __umul24((u32)acc[j], a) + j;			// v_mul_u32_u24
			if (OP == 6) {								// 128-bit += 32x64 (SpMV MAC)
				unsigned __int128 t = ((unsigned __int128)acc[j ^ 1] << 64 | acc[j]) + (unsigned __int128)a * (acc[j] | 1);
				acc[j] = (u64)t; acc[j ^ 1] = (u64)(t >> 64);
			}
			if (OP == 7) {								// 128-bit += 64x64 (dense MAC)
				unsigned __int128 t = ((unsigned __int128)acc[j ^ 1] << 64 | acc[j]) + (unsigned __int128)(acc[j] | 1) * (acc[j ^ 1] | 3);
				acc[j] = (u64)t; acc[j ^ 1] = (u64)(t >> 64);
			}
		}
	}
	u64 s = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) s += acc[j] + (u64)facc[j];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run_alu(const char *name, u64 *out, int ncu)
{
	const int iters = 4096, blocks = ncu * 8;
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_alu<OP>), dim3(blocks), dim3(256), 0, 0, out, 12345u, 777u, 16);
	CHK(hipDeviceSynchronize());
	CHK(hipEventRecord(e0));
	hipLaunchKernelGGL((k_alu<OP>), dim3(blocks), dim3(256), 0, 0, out, 12345u, 777u, iters);
	CHK(hipEventRecord(e1));
	CHK(hipEventSynchronize(e1));
	float ms;
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double ops = (double)blocks * 256 * iters * 8;		// lane-ops
	const double waveops_per_simd = ops / 64 / (ncu * 4.0);
	printf("%-34s %8.3f ms  %8.2f Glane-op/s  -> %6.2f cycles per wave-op per SIMD @2.4GHz\n", name, ms, ops / ms / 1e6,
	       ms * 1e-3 * 2.4e9 / waveops_per_simd);
}

// Each group of RW lanes gathers random (8*RW)-byte rows of a table; `U` independent loads in flight per lane.
template <int U, int RW = 8, int POLICY = 0>
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
		for (int j = 0; j < U; j++) x[j] = POLICY == 1 ? __builtin_nontemporal_load(&table[(size_t)c[j] * RW + lane]) : (POLICY == 2 ? __hip_atomic_load(&table[(size_t)c[j] * RW + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : table[(size_t)c[j] * RW + lane]);
#pragma unroll
		for (int j = 0; j < U; j++) acc += x[j];
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <int U, int RW = 8, int POLICY = 0>
static void run_gather(const u64 *table, const int *idx, long long count, u64 *out, int ncu, double table_mb, int blocks_per_cu)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int blocks = ncu * blocks_per_cu;
	hipLaunchKernelGGL((k_gather<U, RW, POLICY>), dim3(blocks), dim3(256), 0, 0, table, idx, count, out);
	CHK(hipDeviceSynchronize());
	CHK(hipEventRecord(e0));
	hipLaunchKernelGGL((k_gather<U, RW, POLICY>), dim3(blocks), dim3(256), 0, 0, table, idx, count, out);
	CHK(hipEventRecord(e1));
	CHK(hipEventSynchronize(e1));
	float ms;
	CHK(hipEventElapsedTime(&ms, e0, e1));
	printf("gather[policy %d] %3dB rows  table %7.1f MB  in-flight/lane %2d  blocks/CU %d : %8.3f ms  %8.1f GB/s gathered, %6.1f G rows/s (+%.1f GB/s index stream)\n",
	       POLICY, RW * 8, table_mb, U, blocks_per_cu, ms, count * 8.0 * RW / ms / 1e6, count / ms / 1e6, count * 4.0 / ms / 1e6);
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	printf("device %s, %d CUs, clock %d MHz\n", prop.name, ncu, prop.clockRate / 1000);
	u64 *out;
	CHK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 8));
	run_alu<0>("v_mad_u64_u32", out, ncu);
	run_alu<1>("v_mul_lo_u32 (+add)", out, ncu);
	run_alu<2>("v_mul_hi_u32 (+add)", out, ncu);
	run_alu<3>("64-bit add x2", out, ncu);
	run_alu<4>("v_fma_f64", out, ncu);
	run_alu<5>("v_mul_u32_u24 (+add)", out, ncu);
	run_alu<6>("acc128 += u32*u64 (SpMV MAC)", out, ncu);
	run_alu<7>("acc128 += u64*u64 (dense MAC)", out, ncu);

	const long long count = 40000000;	// gathers per launch (2.56 GB gathered)
	std::vector<int> h(count);
	int *idx;
	CHK(hipMalloc(&idx, count * 4));
	const double sizes_mb[] = { 4.0, 35.0, 125.0, 250.0, 790.0 };
	for (double mb : sizes_mb) {
		const long long rows = (long long)(mb * 1e6 / 64);
		u64 *table;
		CHK(hipMalloc(&table, rows * 64));
		CHK(hipMemset(table, 1, rows * 64));
		uint64_t s = 88172645463325252ull;
		for (long long k = 0; k < count; k++) {
			s ^= s << 13; s ^= s >> 7; s ^= s << 17;
			h[k] = (int)(s % (uint64_t)rows);
		}
		CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
		run_gather<8>(table, idx, count, out, ncu, mb, 8);
		if (mb == 125.0) {
			run_gather<4>(table, idx, count, out, ncu, mb, 8);
			run_gather<16>(table, idx, count, out, ncu, mb, 8);
			run_gather<8, 8, 1>(table, idx, count, out, ncu, mb, 8);	// nontemporal loads
			run_gather<8, 8, 2>(table, idx, count, out, ncu, mb, 8);	// sc1 (agent-scope relaxed atomic) loads
			run_gather<8, 8, 0>(table, idx, count, out, ncu, mb, 4);
			run_gather<8, 8, 1>(table, idx, count, out, ncu, mb, 4);
			// row size: 32-byte and 128-byte rows out of the same bytes (indices rescaled on the fly by the table stride)
			for (long long k = 0; k < count; k++) h[k] = h[k] / 2;
			CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
			run_gather<8, 16>(table, idx, count / 2, out, ncu, mb, 8);
			for (long long k = 0; k < count; k++) h[k] = h[k] * 4 + (k & 3);
			CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
			run_gather<8, 4>(table, idx, count, out, ncu, mb, 8);
			// sorted indices: the same kernel reading rows in order (what a locality-friendly matrix would see)
			for (long long k = 0; k < count; k++) h[k] = (int)(k % rows);
			CHK(hipMemcpy(idx, h.data(), count * 4, hipMemcpyHostToDevice));
			run_gather<8>(table, idx, count, out, ncu, mb, 8);
		}
		CHK(hipFree(table));
	}
	return 0;
}
