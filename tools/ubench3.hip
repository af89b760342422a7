/*
 * ubench3 -- what a streaming kernel reaches on this device: the ceiling to hold the dense block kernels against.
 *   read1 / read3 : sum of 1 / 3 arrays (the inner products read 2)
 *   copy          : 1 read + 1 write
 *   upd5          : a' = f(a, b, c), b' = g(a, b, c) in place -- 3 reads + 2 writes, the traffic of orthogonalize()
 * 16 bytes per lane, consecutive lanes on consecutive addresses; persistent grids of G workgroups per CU (grid-stride in
 * whole-grid steps) and one-shot grids; two sizes per array: 122 MB (GL7d19 block, n = 8) and 1.6 GB (config 5 at 1/4, n = 16).
 * Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench3 tools/ubench3.hip
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;

__global__ void __launch_bounds__(256) k_read(const uint4 *a, const uint4 *b, const uint4 *c, long long n, int arrays, u64 *out)
{
	u64 acc = 0;
	for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
		uint4 x = a[i];
		acc += x.x ^ x.y ^ x.z ^ x.w;
		if (arrays > 1) {
			uint4 y = b[i], z = c[i];
			acc += y.x ^ y.y ^ y.z ^ y.w;
			acc += z.x ^ z.y ^ z.z ^ z.w;
		}
	}
	if (acc == 0x1234567887654321ull)
		out[0] = acc;
}

__global__ void __launch_bounds__(256) k_copy(const uint4 *a, uint4 *b, long long n)
{
	for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
		b[i] = a[i];
}

__global__ void __launch_bounds__(256) k_upd5(uint4 *a, uint4 *b, const uint4 *c, long long n)
{
	for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
		uint4 x = a[i], y = b[i], z = c[i];
		uint4 r = { x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w };
		uint4 q = { x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w };
		a[i] = r;
		b[i] = q;
	}
}

template <typename F>
static double time_us(F launch)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0));
	CHK(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++)
		launch();
	std::vector<float> t;
	for (int rep = 0; rep < 7; rep++) {
		CHK(hipEventRecord(e0, 0));
		launch();
		CHK(hipEventRecord(e1, 0));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		t.push_back(ms);
	}
	std::sort(t.begin(), t.end());
	return t[t.size() / 2] * 1e3;
}

int main(void)
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.name, ncu);
	const size_t sizes[2] = { (size_t)1911130 * 64, (size_t)12500000 * 128 };
	u64 *out;
	CHK(hipMalloc(&out, 4096));
	for (int si = 0; si < 2; si++) {
		const size_t bytes = sizes[si] & ~(size_t)4095;
		const long long n = bytes / 16;
		uint4 *a, *b, *c;
		CHK(hipMalloc(&a, bytes));
		CHK(hipMalloc(&b, bytes));
		CHK(hipMalloc(&c, bytes));
		CHK(hipMemset(a, 1, bytes));
		CHK(hipMemset(b, 2, bytes));
		CHK(hipMemset(c, 3, bytes));
		printf("== arrays of %.1f MB ==\n", bytes / 1e6);
		const int grids[] = { 2, 4, 8, 16, 0 };		/* workgroups per CU; 0 = one-shot */
		for (int g : grids) {
			const unsigned blocks = g ? (unsigned)(g * ncu) : (unsigned)((n + 255) / 256);
			double t;
			t = time_us([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, b, c, n, 1, out); });
			printf("  grid %2d/CU  read1 %8.1f us %6.2f TB/s", g, t, bytes / t / 1e6);
			t = time_us([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, b, c, n, 3, out); });
			printf("   read3 %8.1f us %6.2f TB/s", t, 3.0 * bytes / t / 1e6);
			t = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
			printf("   copy %8.1f us %6.2f TB/s", t, 2.0 * bytes / t / 1e6);
			t = time_us([&] { hipLaunchKernelGGL(k_upd5, dim3(blocks), dim3(256), 0, 0, a, b, c, n); });
			printf("   upd5 %8.1f us %6.2f TB/s\n", t, 5.0 * bytes / t / 1e6);
		}
		CHK(hipFree(a));
		CHK(hipFree(b));
		CHK(hipFree(c));
	}
	return 0;
}
