// What does a ds_read_b128 cost by the number of active lanes? (round 5: would y-neighbour values carried by DPP, with LDS fix-ups on a few lanes per wave, save LDS time?)
// One workgroup of 8 waves per CU, every wave issues N ds_read_b128 at the 48-byte stride of the SOR kernel's rows, with all 64 lanes / lanes 0 and 63 / lane 0 only active.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int n) {
	__shared__ f4 a[3 * 1024];
	const int t = threadIdx.x, l = t & 63;
	for (int i = t; i < 3 * 1024; i += 512) a[i] = f4{(float)i, 1.0f, 2.0f, 3.0f};
	__syncthreads();
	const bool on = MODE == 0 ? true : (MODE == 1 ? (l == 0 || l == 63) : (MODE == 2 ? l == 0 : (l & 7) == 0));
	f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
	int idx = t * 3;
	for (int i = 0; i < n; ++i) {
		if (on) {
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const f4 v = a[(idx + u * 7) & 3071];
				acc += v;
			}
		}
		idx = (idx + 3) & 3071;
	}
	if (acc.x == 12345.0f) out[t] = acc.y;
}
int main() {
	float* d;
	hipMalloc(&d, 4096);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int n = 20000, blocks = 256;
	const char* names[] = {"all 64 lanes", "lanes 0 and 63", "lane 0", "every 8th lane"};
	for (int rep = 0; rep < 2; ++rep)
		for (int m = 0; m < 4; ++m) {
			hipEventRecord(e0);
			if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, n);
			if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, n);
			if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, d, n);
			if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, d, n);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			// per CU: 8 waves x n x 8 reads
			printf("%-16s %8.3f ms  = %.2f ns per wave-instruction per CU (8 waves issuing)\n", names[m], ms, 1e6 * ms / ((double)n * 8 * 8));
		}
	return 0;
}
