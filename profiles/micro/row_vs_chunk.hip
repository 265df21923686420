// Microbenchmark (calibration, not product): does a 16-byte-per-lane load cost the texture path more when lanes are 32 B
// apart ("row" mapping: lane = row, two loads per 32-byte row) than when consecutive lanes read consecutive 16-byte
// chunks ("chunk" mapping)? Both variants read the same 2 KB per wave and write it back (2 KB), like the own-leaf
// traffic of the SOR kernel. build: hipcc --offload-arch=gfx950 -O3 row_vs_chunk.hip -o row_vs_chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_row(const float* __restrict__ in, float* __restrict__ out) {
	const int l = threadIdx.x;
	const float4* q = reinterpret_cast<const float4*>(in + (size_t)blockIdx.x * 512 + l * 8);
	float4 a = q[0], b = q[1];
	a.x += 1.0f;
	b.w += 1.0f;
	float4* o = reinterpret_cast<float4*>(out + (size_t)blockIdx.x * 512 + l * 8);
	o[0] = a;
	o[1] = b;
}
__global__ __launch_bounds__(64) void k_chunk(const float* __restrict__ in, float* __restrict__ out) {
	const int l = threadIdx.x;
	const float4* q = reinterpret_cast<const float4*>(in + (size_t)blockIdx.x * 512);
	float4 a = q[l], b = q[l + 64];
	a.x += 1.0f;
	b.w += 1.0f;
	float4* o = reinterpret_cast<float4*>(out + (size_t)blockIdx.x * 512);
	o[l] = a;
	o[l + 64] = b;
}

int main() {
	const size_t leaves = 32768 * 2, n = leaves * 512;
	float *a, *b;
	hipMalloc(&a, n * 4);
	hipMalloc(&b, n * 4);
	hipMemset(a, 0, n * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int variant = 0; variant < 2; ++variant) {
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0);
			for (int i = 0; i < 50; ++i) {
				if (variant == 0) hipLaunchKernelGGL(k_row, dim3(leaves), dim3(64), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
				else hipLaunchKernelGGL(k_chunk, dim3(leaves), dim3(64), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
			}
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			printf("%s: %.2f us per pass, %.0f GB/s (read+write %zu MB)\n", variant ? "chunk" : "row  ", 1e3 * ms / 50, 2.0 * n * 4 / (ms / 50 * 1e-3) / 1e9, 2 * n * 4 >> 20);
		}
	}
	return 0;
}
