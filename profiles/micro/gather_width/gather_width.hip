// gather_width.hip -- calibration for the layout lever of VERDICT r4 item 4: what does a coherent gather (the 64 lanes of a wave reading 64 consecutive
// voxels at a wave-dependent place, as an advection tap of a locally uniform back-trace does) cost per INSTRUCTION for 4-, 8-, 12- and 16-byte voxels,
// i.e. what would one buffer_load_dwordx4 over four fields interleaved as float4 cost against the four buffer_load_dword gathers of four separate
// fields that k_advect_scalars_n issues today? Each wave issues 16 rounds of 8 gathers; the window stays in the L2 / Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 gather_width.hip -o gather_width
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ float ld1(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ v2f ld2(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ v3f ld3(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
__device__ v4f ld4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// W = floats per voxel and per load; FIELDS = separate arrays gathered at the same voxel (W = 1 only). FIELDS = -3: the three components of 12-byte
// voxels as three dword loads (lanes 12 bytes apart); FIELDS = -2: a z-pair of a 4-byte field as two dword loads at +0 / +4 (what ld_zpair fuses into a dwordx2)
template <int W, int FIELDS>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, unsigned bytes, unsigned field_stride) {
	const unsigned long long a = (unsigned long long)src;
	v4i r;
	r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
	r.z = (int)bytes;
	r.w = 0x00020000;
	const int l = threadIdx.x & 63;
	// voxel number: a wave reads 64 consecutive voxels starting at an odd place (a back-trace lands anywhere), another place every round
	unsigned vox = (blockIdx.x & 63u) * 4096u + (threadIdx.x >> 6) * 517u + (unsigned)l + 3u;
	float acc = 0.0f;
#pragma unroll 1
	for (int round = 0; round < 16; ++round) {
		float v[8][4];
#pragma unroll
		for (int q = 0; q < (FIELDS < 0 ? 8 / -FIELDS : 8 / FIELDS); ++q) {
			const unsigned o = (vox + (unsigned)q * 8u + (unsigned)round * 511u) * (unsigned)(FIELDS == -3 ? 12 : 4 * W);
			if (FIELDS < 0) {
#pragma unroll
				for (int f = 0; f < -FIELDS; ++f) v[q * -FIELDS + f][0] = ld1(r, (int)(o + 4u * (unsigned)f), 0, 0);
			} else if (W == 1) {
#pragma unroll
				for (int f = 0; f < FIELDS; ++f) v[q * FIELDS + f][0] = ld1(r, (int)(o + (unsigned)f * field_stride), 0, 0);
			} else if (W == 2) {
				const v2f t = ld2(r, (int)o, 0, 0); v[q][0] = t.x + t.y;
			} else if (W == 3) {
				const v3f t = ld3(r, (int)o, 0, 0); v[q][0] = t.x + t.y + t.z;
			} else {
				const v4f t = ld4(r, (int)o, 0, 0); v[q][0] = t.x + t.y + t.z + t.w;
			}
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) acc += v[q][0];
	}
	if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int W, int FIELDS>
void run(const char* name, const float* src, float* out, unsigned bytes) {
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int t = 0; t < 6; ++t) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL((k<W, FIELDS>), dim3(16384), dim3(512), 0, 0, src, out, bytes, 20u << 20);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	// 16384 blocks x 8 waves x 128 gather instructions per wave, one texture addresser / L1 per CU (256)
	printf("%-58s %8.1f us = %5.1f cycles per gather instruction per CU at 2.4 GHz\n", name, 1e3 * best, best * 1e-3 * 2.4e9 / (16384.0 * 8 * 128 / 256));
}

int main() {
	const unsigned bytes = 96u << 20;
	float *src, *out;
	(void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 4096); (void)hipMemset(src, 0, bytes);
	for (int rep = 0; rep < 2; ++rep) {
		run<1, 1>("dword   (4-byte voxels, one field)", src, out, bytes);
		run<1, 4>("dword x 4 fields at the same voxel (4 separate arrays)", src, out, bytes);
		run<2, 1>("dwordx2 (8-byte voxels)", src, out, bytes);
		run<3, 1>("dwordx3 (12-byte voxels: Vec3f)", src, out, bytes);
		run<4, 1>("dwordx4 (16-byte voxels: four fields as float4)", src, out, bytes);
		run<1, -3>("dword x 3 components of 12-byte voxels (lanes 12 B apart)", src, out, bytes);
		run<1, -2>("dword x 2: z and z+1 of a 4-byte field (two loads, +0 / +4)", src, out, bytes);
	}
	return 0;
}
