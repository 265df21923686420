// Microbenchmark (calibration, not product): what does the texture addresser charge a 12-byte gather whose exec mask has only some
// lanes on? Every wave issues 16 rounds of 8 buffer_load_dwordx3 gathers at pseudo-random (L2-resident) offsets under one of five
// exec patterns; the time per launch against the full-mask pattern says whether a partially masked gather is cheaper, and whether
// the unit of charge is the lane, the 4-lane quad or the instruction.
// build: hipcc --offload-arch=gfx950 -O3 ta_lanes.hip -o ta_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v3f __attribute__((ext_vector_type(3)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ v3f ld3(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");

template <int PATTERN>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, unsigned bytes) {
	const unsigned long long a = (unsigned long long)src;
	v4i r;
	r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
	r.z = (int)bytes;
	r.w = 0x00020000;
	const int l = threadIdx.x & 63;
	bool on = true;
	if (PATTERN == 1) on = (l & 7) == 7;   // 8 lanes, one in every second quad (8 of 16 quads)
	if (PATTERN == 2) on = l >= 56;        // 8 lanes in 2 quads
	if (PATTERN == 3) on = l < 32;         // 32 lanes, 8 quads
	if (PATTERN == 4) on = (l & 3) == 0;   // 16 lanes, one per quad (16 quads)
	// COHERENT: the lanes of a wave read 64 consecutive 12-byte voxels (what an advection gather of a locally uniform back-trace looks like);
	// each round moves the wave to another place of a 6 MB window that stays in the L2
	unsigned o = ((blockIdx.x & 63u) * 4096u + (threadIdx.x >> 6) * 512u + (unsigned)l) * 12u;
	float acc = 0.0f;
	if (on) {
#pragma unroll 1
		for (int round = 0; round < 16; ++round) {
			v3f v[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) v[q] = ld3(r, (int)(o + (unsigned)q * 96u + (unsigned)round * 6144u), 0, 0);
#pragma unroll
			for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y + v[q].z;
		}
	}
	if (acc == 12345.678f) out[threadIdx.x] = acc;
}

int main() {
	const unsigned bytes = 96u << 20;  // inside the L2s / Infinity Cache after the first touch
	float *src, *out;
	hipMalloc(&src, bytes);
	hipMalloc(&out, 4096);
	hipMemset(src, 0, bytes);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const char* names[5] = {"all 64 lanes", "lanes l%8==7 (8 lanes, 8 quads)", "lanes >= 56 (8 lanes, 2 quads)", "lanes < 32 (32 lanes, 8 quads)", "lanes l%4==0 (16 lanes, 16 quads)"};
	for (int rep = 0; rep < 2; ++rep)
		for (int p = 0; p < 5; ++p) {
			float best = 1e9f;
			for (int t = 0; t < 5; ++t) {
				hipEventRecord(e0);
				const dim3 g(16384), b(512);
				switch (p) {
					case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, src, out, bytes); break;
					case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, src, out, bytes); break;
					case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, src, out, bytes); break;
					case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, src, out, bytes); break;
					default: hipLaunchKernelGGL(k<4>, g, b, 0, 0, src, out, bytes); break;
				}
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			// 16384 blocks x 8 waves x 128 gathers per wave, one texture addresser per CU (256)
			printf("%-36s %8.1f us  = %.1f addresser cycles per gather instruction at 2.4 GHz\n", names[p], 1e3 * best, best * 1e-3 * 2.4e9 / (16384.0 * 8 * 128 / 256));
		}
	return 0;
}
