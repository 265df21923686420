// lds_tile_gather.hip -- round 6 calibration: would the advection kernels' trilinear taps cost less out of an LDS tile than out of the L1?
// k_advect_vector_n is bound by the L1 tag pipe (one lookup per cycle, ~26 lookups per 12-byte gather: profiles/floors.py). This emulates its access pattern on a dense 256^3 grid:
// one 512-thread workgroup per 8^3 leaf, a thread per voxel, two trilinear samples = 16 gathers of 12 bytes at voxel + (a back-trace that is uniform over the workgroup, anywhere within
// +-4 voxels) + the corner offset, plus the voxel's own load and one 12-byte store.
//   G: every gather a global load (the neighbour leaf's base from a 27-entry table in LDS, as the product does);
//   L: the workgroup first stages the 16^3-voxel neighbourhood of its leaf (48 KB: 6 coalesced 16-byte loads per thread out of up to 27 leaves) in LDS and gathers from there.
//   hipcc --offload-arch=gfx950 -O3 lds_tile_gather.hip -o lds_tile_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int LN = 32;  // leaves per axis: 256^3 voxels
struct f3 { float x, y, z; };
__device__ __forceinline__ int leaf_of(int lx, int ly, int lz) { return (((lx + LN) % LN) * LN + ((ly + LN) % LN)) * LN + ((lz + LN) % LN); }

template <int LDS>  // 0: global, 1: 12-byte packed tile, 2: 16-byte padded tile (64 KB), 3: three component planes (48 KB), z index XOR-swizzled by y so that a wave's 8 x 8 (y, z) window hits every bank twice
__global__ __launch_bounds__(512) void k(const float* __restrict__ u, float* __restrict__ out, const int* __restrict__ shifts) {
	__shared__ int s_base[27];
	__shared__ __attribute__((aligned(16))) float s_tile[(LDS == 1 || LDS == 3) ? 16 * 16 * 16 * 3 : (LDS == 2 ? 16 * 16 * 16 * 4 : 4)];
	const int leaf = blockIdx.x, n = threadIdx.x;
	const int lx = leaf / (LN * LN), ly = (leaf / LN) % LN, lz = leaf % LN;
	if (n < 27) s_base[n] = leaf_of(lx + n / 9 - 1, ly + (n / 3) % 3 - 1, lz + n % 3 - 1) * 512;
	const int sx = shifts[3 * leaf], sy = shifts[3 * leaf + 1], sz = shifts[3 * leaf + 2];  // the workgroup's back-trace, voxels, each in [-4, 3]
	if (LDS == 2) {
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < 8; ++kk) {
			const int v = n + 512 * kk, tx = v >> 8, ty = (v >> 4) & 15, tz = v & 15;
			const int cx = tx < 4 ? 0 : (tx < 12 ? 1 : 2), cy = ty < 4 ? 0 : (ty < 12 ? 1 : 2), cz = tz < 4 ? 0 : (tz < 12 ? 1 : 2);
			const float* t = u + (size_t)(s_base[cx * 9 + cy * 3 + cz] + ((((tx + 4) & 7) << 6) | (((ty + 4) & 7) << 3) | ((tz + 4) & 7))) * 3;
			*reinterpret_cast<float4*>(&s_tile[v * 4]) = make_float4(t[0], t[1], t[2], 0.0f);
		}
	}
	if (LDS == 3) {
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < 6; ++kk) {
			const int p = n + 512 * kk, row = p / 12, part = p - row * 12;
			const int tx = row >> 4, ty = row & 15;
			const int cx = tx < 4 ? 0 : (tx < 12 ? 1 : 2), cy = ty < 4 ? 0 : (ty < 12 ? 1 : 2), cz = part < 3 ? 0 : (part < 9 ? 1 : 2);
			const int vx = (tx + 4) & 7, vy = (ty + 4) & 7;
			const int zoff = cz == 0 ? 12 + part * 4 : (cz == 1 ? (part - 3) * 4 : (part - 9) * 4);
			const float4 v = *reinterpret_cast<const float4*>(u + (size_t)(s_base[cx * 9 + cy * 3 + cz] + ((vx << 6) | (vy << 3))) * 3 + zoff);
			const float e[4] = {v.x, v.y, v.z, v.w};
			const int swz = ((ty >> 1) & 1) << 3;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int f = part * 4 + q, vz = f / 3, c = f - vz * 3;
				s_tile[c * 4096 + row * 16 + (vz ^ swz)] = e[q];
			}
		}
	}
	if (LDS == 1) {
		__syncthreads();
#pragma unroll
		for (int kk = 0; kk < 6; ++kk) {
			const int p = n + 512 * kk, row = p / 12, part = p - row * 12;
			const int tx = row >> 4, ty = row & 15;
			const int cx = tx < 4 ? 0 : (tx < 12 ? 1 : 2), cy = ty < 4 ? 0 : (ty < 12 ? 1 : 2), cz = part < 3 ? 0 : (part < 9 ? 1 : 2);
			const int vx = (tx + 4) & 7, vy = (ty + 4) & 7;
			const int zf = cz == 0 ? 48 + (part) * 4 * 1 : (cz == 1 ? (part - 3) * 4 : (part - 9) * 4);  // float offset inside the source leaf's z-row of 24 floats
			const int zoff = cz == 0 ? 12 + part * 4 : (cz == 1 ? (part - 3) * 4 : (part - 9) * 4);
			(void)zf;
			const float4 v = *reinterpret_cast<const float4*>(u + (size_t)(s_base[cx * 9 + cy * 3 + cz] + ((vx << 6) | (vy << 3))) * 3 + zoff);
			*reinterpret_cast<float4*>(&s_tile[row * 48 + part * 4]) = v;
		}
	}
	__syncthreads();
	const int x = n >> 6, y = (n >> 3) & 7, z = n & 7;
	float ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		const int bx = x + sx + pass, by = y + sy - pass, bz = z + sz;  // lower corner of the cell, relative to the leaf origin: in [-5, 11]
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const int i = bx + (c >> 2), j = by + ((c >> 1) & 1), kq = bz + (c & 1);
			f3 v;
			if (LDS == 2) {
				const float4 t = *reinterpret_cast<const float4*>(&s_tile[((((i + 4) & 15) * 16 + ((j + 4) & 15)) * 16 + ((kq + 4) & 15)) * 4]);
				v.x = t.x, v.y = t.y, v.z = t.z;
			} else if (LDS == 3) {
				const int Y = (j + 4) & 15;
				const float* t = &s_tile[((((i + 4) & 15) * 16 + Y) * 16) + (((kq + 4) & 15) ^ (((Y >> 1) & 1) << 3))];
				v.x = t[0], v.y = t[4096], v.z = t[8192];
			} else if (LDS == 1) {
				const float* t = &s_tile[(((i + 4) & 15) * 16 + ((j + 4) & 15)) * 48 + ((kq + 4) & 15) * 3];
				v.x = t[0], v.y = t[1], v.z = t[2];
			} else if (LDS == 4) {  // planar velocity in global memory: three 4-byte gathers per tap
				const int slot = ((i + 8) >> 3) * 9 + ((j + 8) >> 3) * 3 + ((kq + 8) >> 3);
				const float* t = u + (size_t)(s_base[slot] + (((i & 7) << 6) | ((j & 7) << 3) | (kq & 7)));
				constexpr size_t plane = (size_t)LN * LN * LN * 512;
				v.x = t[0], v.y = t[plane], v.z = t[2 * plane];
			} else {
				const int slot = ((i + 8) >> 3) * 9 + ((j + 8) >> 3) * 3 + ((kq + 8) >> 3);
				const float* t = u + (size_t)(s_base[slot] + (((i & 7) << 6) | ((j & 7) << 3) | (kq & 7))) * 3;
				v.x = t[0], v.y = t[1], v.z = t[2];
			}
			ax += v.x * (float)(c + 1), ay += v.y, az += v.z;
		}
	}
	float* o = out + ((size_t)leaf * 512 + n) * 3;
	o[0] = ax, o[1] = ay, o[2] = az;
}

int main() {
	const size_t nvox = (size_t)LN * LN * LN * 512;
	float *u, *out;
	int* sh;
	(void)hipMalloc(&u, nvox * 12); (void)hipMalloc(&out, nvox * 12); (void)hipMalloc(&sh, LN * LN * LN * 12);
	(void)hipMemset(u, 0, nvox * 12);
	int* h = (int*)malloc(LN * LN * LN * 12);
	srand(7);
	for (int i = 0; i < LN * LN * LN * 3; ++i) h[i] = rand() % 8 - 4;
	(void)hipMemcpy(sh, h, LN * LN * LN * 12, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int rep = 0; rep < 3; ++rep)
		for (int v = 0; v < 5; ++v) {
			float best = 1e9f;
			for (int t = 0; t < 5; ++t) {
				(void)hipEventRecord(e0);
				if (v == 4) hipLaunchKernelGGL(k<4>, dim3(LN * LN * LN), dim3(512), 0, 0, u, out, sh);
				else if (v == 3) hipLaunchKernelGGL(k<3>, dim3(LN * LN * LN), dim3(512), 0, 0, u, out, sh);
				else if (v == 2) hipLaunchKernelGGL(k<2>, dim3(LN * LN * LN), dim3(512), 0, 0, u, out, sh);
				else if (v) hipLaunchKernelGGL(k<1>, dim3(LN * LN * LN), dim3(512), 0, 0, u, out, sh);
				else hipLaunchKernelGGL(k<0>, dim3(LN * LN * LN), dim3(512), 0, 0, u, out, sh);
				(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
				float ms; (void)hipEventElapsedTime(&ms, e0, e1);
				best = ms < best ? ms : best;
			}
			printf("%s: %7.1f us for 256^3 (16 gathers of 12 bytes per voxel + own store)\n", v == 4 ? "P  taps out of global memory, planar velocity (3 x 4-byte gathers) " : v == 3 ? "S  taps out of three swizzled component planes in LDS (48 KB)" : v == 2 ? "L4 taps out of a 16^3 LDS tile of 16-byte voxels (64 KB)" : (v ? "L  taps out of a 16^3 LDS tile (48 KB staged per leaf) " : "G  taps out of global memory (L1)                       "), 1e3 * best);
		}
	return 0;
}
