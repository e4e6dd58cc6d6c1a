// Microbenchmark (GPU box): peak rate of independent random 32-byte gathers (one probe-table bucket each) from tables of
// several sizes, UNR gathers in flight per lane -- the memory-system ceiling for the probe kernels.
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/gather_bench tools/gather_bench.hip && gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int UNR, int HALF>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *table, uint64_t mask, int iters, uint32_t *out) {
	uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B1u + 12345u;
	uint32_t acc = 0;
	for(int it = 0; it < iters; ++it) {
		uint4 a[UNR], b[UNR];
#pragma unroll
		for(int u = 0; u < UNR; ++u) {
			x = x * 1664525u + 1013904223u;
			const uint64_t row = ((uint64_t) (x ^ (x >> 15)) * 2654435761ull >> 7) & mask;
			a[u] = table[row * 2]; b[u] = HALF ? a[u] : table[row * 2 + 1];      // HALF: one 16-byte access per gather
		}
#pragma unroll
		for(int u = 0; u < UNR; ++u) acc ^= a[u].x ^ a[u].w ^ b[u].y ^ b[u].z;
	}
	if(acc == 0x12345678u) out[0] = acc;
}

template <int UNR, int HALF>
static void run(const uint4 *table, uint64_t rows, uint32_t *out, const char *label) {
	const int blocks = 256 * 32, iters = 64 / UNR * 4;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	gather_kernel<UNR, HALF><<<blocks, 256>>>(table, rows - 1, iters, out);
	hipEventRecord(e0);
	for(int r = 0; r < 3; ++r) gather_kernel<UNR, HALF><<<blocks, 256>>>(table, rows - 1, iters, out);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	const double n = (double) blocks * 256 * iters * UNR;
	printf("%s  %d in flight/lane: %7.1f G gathers/s  (%.0f M gathers in %.3f ms; %.2f TB/s at 32 B, %.2f at 64 B, %.2f at 128 B per gather)\n",
	       label, UNR, n / ms / 1e6, n / 1e6, ms, n * 32 / ms / 1e9, n * 64 / ms / 1e9, n * 128 / ms / 1e9);
}

int main() {
	uint32_t *out; hipMalloc(&out, 64);
	for(uint64_t mb : {4ull, 16ull, 64ull, 256ull, 2048ull}) {
		const uint64_t rows = mb * 1024 * 1024 / 32;
		uint4 *table; hipMalloc(&table, rows * 32);
		hipMemset(table, 1, rows * 32);
		char label[64]; snprintf(label, sizeof label, "table %5llu MB", (unsigned long long) mb);
		run<1, 0>(table, rows, out, label);
		run<2, 0>(table, rows, out, label);
		run<2, 1>(table, rows, out, "   16-B access ");
		hipFree(table);
	}
	return 0;
}
