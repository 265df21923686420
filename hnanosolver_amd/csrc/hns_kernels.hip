// hns_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the HNanoSolver substep hot path.
//
// One 8^3 leaf per workgroup; wave64. All arithmetic is float32 and keeps the association of the reference kernels
// (reference src/Cuda/Kernel.cu) so that results are reproducible against oracle/hns_oracle.c; build with
// -ffp-contract=off: the only fused multiply-adds are the explicit __fmaf_rn() calls, which sit where the reference
// itself calls __fmaf_rn / fmaf (Kernel.cu:241-247, Stencils.hpp:20-22,131-135).
//
// Differences in mechanism (not in arithmetic) from the reference:
//   * no per-voxel coordinate stream: a voxel's coordinate is leaf origin + thread id (saves 12 B/voxel/kernel);
//   * no NanoVDB tree walk: a tap resolves its leaf through the 27-entry neighbour table of the workgroup's leaf
//     (staged in LDS) or, beyond one leaf away, an origin hash;
//   * velocity is planar (ux, uy, uz) on the device, so every component is a leaf-dense float field and every
//     2 KB leaf payload is one fully coalesced run;
//   * the red-black SOR loop runs ONE launch per iteration: the workgroup stages its leaf plus a two-voxel halo of
//     p in LDS, recomputes the red updates of the face-adjacent halo voxels itself (bit-identical to what the
//     neighbouring workgroup computes), then does black, ping-ponging p_in -> p_out. 12 B/voxel/iteration of HBM
//     traffic instead of the >=16 B of two in-place launches.
#include <cstdlib>
#include <cstring>

#include "hns_internal.hpp"

namespace hns {

// ---------------------------------------------------------------------------------------------------------------
// topology access on the device
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t d_hash_origin(int x, int y, int z) {
	uint32_t h = (uint32_t)(x >> 3) * 0x9E3779B1u;
	h ^= (uint32_t)(y >> 3) * 0x85EBCA77u;
	h ^= (uint32_t)(z >> 3) * 0xC2B2AE3Du;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	h ^= h >> 12;
	return h;
}

__device__ __forceinline__ int d_find_leaf(const GridDev& g, int ox, int oy, int oz) {
	uint32_t s = d_hash_origin(ox, oy, oz) & g.hash_mask;
	for (;;) {
		const int l = g.hash[s];
		if (l < 0) return -1;
		const int4 o = g.origins[l];
		if (o.x == ox && o.y == oy && o.z == oz) return l;
		s = (s + 1) & g.hash_mask;
	}
}

// Flat index of global voxel (i,j,k), or -1 when its leaf is absent. `org` is the workgroup's leaf origin and s_nbr its
// 27-neighbour table (LDS). Replaces IndexOffsetSampler<0>::offset (reference Stencils.hpp:59-61), minus the +1.
__device__ __forceinline__ int tap_index(const GridDev& g, const int* s_nbr, const int4 org, int i, int j, int k) {
	const int dx = (i >> 3) - (org.x >> 3), dy = (j >> 3) - (org.y >> 3), dz = (k >> 3) - (org.z >> 3);
	int leaf;
	if ((unsigned)(dx + 1) <= 2u && (unsigned)(dy + 1) <= 2u && (unsigned)(dz + 1) <= 2u)
		leaf = s_nbr[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)];
	else
		leaf = d_find_leaf(g, i & ~7, j & ~7, k & ~7);
	return leaf < 0 ? -1 : leaf * 512 + (((i & 7) << 6) | ((j & 7) << 3) | (k & 7));
}

// IndexSampler<float,0> (Stencils.hpp:81-89): value, or 0 outside the domain
__device__ __forceinline__ float ld0(const float* __restrict__ f, int idx) { return idx < 0 ? 0.0f : f[idx]; }

struct f3 {
	float x, y, z;
};

__device__ __forceinline__ f3 ld0v(const float* __restrict__ ux, const float* __restrict__ uy, const float* __restrict__ uz, int idx) {
	f3 r;
	r.x = idx < 0 ? 0.0f : ux[idx];
	r.y = idx < 0 ? 0.0f : uy[idx];
	r.z = idx < 0 ? 0.0f : uz[idx];
	return r;
}

// The eight corner indices of the trilinear stencil at base (i,j,k): order v[di][dj][dk] -> t[di*4+dj*2+dk]
__device__ __forceinline__ void tap8(const GridDev& g, const int* s_nbr, const int4 org, int i, int j, int k, int (&t)[8]) {
	// Fast path: all eight corners in one leaf (true for (7/8)^3 of positions) -> one leaf lookup.
	if ((i & 7) != 7 && (j & 7) != 7 && (k & 7) != 7) {
		const int b = tap_index(g, s_nbr, org, i, j, k);
		t[0] = b;
		t[1] = b < 0 ? -1 : b + 1;
		t[2] = b < 0 ? -1 : b + 8;
		t[3] = b < 0 ? -1 : b + 9;
		t[4] = b < 0 ? -1 : b + 64;
		t[5] = b < 0 ? -1 : b + 65;
		t[6] = b < 0 ? -1 : b + 72;
		t[7] = b < 0 ? -1 : b + 73;
	} else {
		t[0] = tap_index(g, s_nbr, org, i, j, k);
		t[1] = tap_index(g, s_nbr, org, i, j, k + 1);
		t[2] = tap_index(g, s_nbr, org, i, j + 1, k);
		t[3] = tap_index(g, s_nbr, org, i, j + 1, k + 1);
		t[4] = tap_index(g, s_nbr, org, i + 1, j, k);
		t[5] = tap_index(g, s_nbr, org, i + 1, j, k + 1);
		t[6] = tap_index(g, s_nbr, org, i + 1, j + 1, k);
		t[7] = tap_index(g, s_nbr, org, i + 1, j + 1, k + 1);
	}
}

// float lerp of TrilinearSampler (Stencils.hpp:140): a + w*(b-a), unfused
__device__ __forceinline__ float lerp_f(float a, float b, float w) { return a + w * (b - a); }
// Vec3f lerp on the device branch (Stencils.hpp:131-135): fmaf(w, b-a, a)
__device__ __forceinline__ float lerp_c(float a, float b, float w) { return __fmaf_rn(w, b - a, a); }

// IndexSampler<float,1>(Vec3f) (Stencils.hpp:117-153): Floor, 8 taps, lerp z then y then x
__device__ __forceinline__ float tri_f(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ f, float x, float y,
                                       float z) {
	const int i = __float2int_rd(x), j = __float2int_rd(y), k = __float2int_rd(z);
	x -= (float)i;
	y -= (float)j;
	z -= (float)k;
	int t[8];
	tap8(g, s_nbr, org, i, j, k, t);
	const float z0 = lerp_f(ld0(f, t[0]), ld0(f, t[1]), z);
	const float z1 = lerp_f(ld0(f, t[2]), ld0(f, t[3]), z);
	const float z2 = lerp_f(ld0(f, t[4]), ld0(f, t[5]), z);
	const float z3 = lerp_f(ld0(f, t[6]), ld0(f, t[7]), z);
	const float y0 = lerp_f(z0, z1, y);
	const float y1 = lerp_f(z2, z3, y);
	return lerp_f(y0, y1, x);
}

__device__ __forceinline__ float tri_c(const float* __restrict__ f, const int (&t)[8], float x, float y, float z) {
	const float z0 = lerp_c(ld0(f, t[0]), ld0(f, t[1]), z);
	const float z1 = lerp_c(ld0(f, t[2]), ld0(f, t[3]), z);
	const float z2 = lerp_c(ld0(f, t[4]), ld0(f, t[5]), z);
	const float z3 = lerp_c(ld0(f, t[6]), ld0(f, t[7]), z);
	const float y0 = lerp_c(z0, z1, y);
	const float y1 = lerp_c(z2, z3, y);
	return lerp_c(y0, y1, x);
}

// IndexSampler<Vec3f,1>(Vec3f): one index set shared by the three planar components
__device__ __forceinline__ f3 tri_v(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ ux,
                                    const float* __restrict__ uy, const float* __restrict__ uz, float x, float y, float z) {
	const int i = __float2int_rd(x), j = __float2int_rd(y), k = __float2int_rd(z);
	x -= (float)i;
	y -= (float)j;
	z -= (float)k;
	int t[8];
	tap8(g, s_nbr, org, i, j, k, t);
	f3 r;
	r.x = tri_c(ux, t, x, y, z);
	r.y = tri_c(uy, t, x, y, z);
	r.z = tri_c(uz, t, x, y, z);
	return r;
}

// Stage the workgroup's leaf id, origin and 27-neighbour table. Returns false for an out-of-range block.
struct LeafCtx {
	int leaf;
	int4 org;
};

__device__ __forceinline__ LeafCtx stage_leaf(const GridDev& g, int* s_nbr, int block) {
	LeafCtx c;
	c.leaf = g.sched ? g.sched[block] : block;
	c.org = g.origins[c.leaf];
	if (threadIdx.x < 27) s_nbr[threadIdx.x] = g.nbr27[c.leaf * 27 + threadIdx.x];
	__syncthreads();
	return c;
}

// ---------------------------------------------------------------------------------------------------------------
// collision helpers (reference Kernel.cu:8-74)
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ f3 sdf_normal(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ sdf, int i, int j, int k,
                                         float eps) {
	const float right = ld0(sdf, tap_index(g, s_nbr, org, i + 1, j, k));
	const float left = ld0(sdf, tap_index(g, s_nbr, org, i - 1, j, k));
	const float top = ld0(sdf, tap_index(g, s_nbr, org, i, j + 1, k));
	const float bottom = ld0(sdf, tap_index(g, s_nbr, org, i, j - 1, k));
	const float front = ld0(sdf, tap_index(g, s_nbr, org, i, j, k + 1));
	const float back = ld0(sdf, tap_index(g, s_nbr, org, i, j, k - 1));
	const float s = 0.5f * eps;
	f3 gr = {s * (right - left), s * (top - bottom), s * (front - back)};
	const float len = sqrtf(gr.x * gr.x + gr.y * gr.y + gr.z * gr.z);
	if (len > 1e-6f) {
		const float inv = 1.0f / len;
		gr.x = inv * gr.x;
		gr.y = inv * gr.y;
		gr.z = inv * gr.z;
	} else {
		gr.x = gr.y = gr.z = 0.0f;
	}
	return gr;
}

__device__ __forceinline__ f3 no_slip_blend(f3 v, f3 n, float blend) {
	// applyNoSlipBoundary (Kernel.cu:57-74) then v*(1-blend) + no_slip*blend (Kernel.cu:114,448,824)
	const float vdotn = v.x * n.x + v.y * n.y + v.z * n.z;
	const f3 t = {v.x - vdotn * n.x, v.y - vdotn * n.y, v.z - vdotn * n.z};
	const float a = 1.0f - blend;
	f3 r = {a * v.x + blend * t.x, a * v.y + blend * t.y, a * v.z + blend * t.z};
	return r;
}

// ---------------------------------------------------------------------------------------------------------------
// advect_vector (reference Kernel.cu:354-453): BFECC self-advection of the velocity, clamped
// ---------------------------------------------------------------------------------------------------------------

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_vector(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                       const float* __restrict__ uz, float* __restrict__ ox, float* __restrict__ oy,
                                                       float* __restrict__ oz, const float* __restrict__ sdf, const float scaled_dt,
                                                       const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const f3 vo = {ux[idx], uy[idx], uz[idx]};
	float bx = px - scaled_dt * vo.x, by = py - scaled_dt * vo.y, bz = pz - scaled_dt * vo.z;
	if (COLL) {
		if (tri_f(g, s_nbr, L.org, sdf, bx, by, bz) < 0.0f) {
			bx = px;
			by = py;
			bz = pz;
		}
	}
	const f3 vf = tri_v(g, s_nbr, L.org, ux, uy, uz, bx, by, bz);
	float fx = bx + scaled_dt * vf.x, fy = by + scaled_dt * vf.y, fz = bz + scaled_dt * vf.z;
	if (COLL) {
		if (tri_f(g, s_nbr, L.org, sdf, fx, fy, fz) < 0.0f) {
			fx = bx;
			fy = by;
			fz = bz;
		}
	}
	const f3 vb = tri_v(g, s_nbr, L.org, ux, uy, uz, fx, fy, fz);
	f3 vc = {vf.x + 0.5f * (vo.x - vb.x), vf.y + 0.5f * (vo.y - vb.y), vf.z + 0.5f * (vo.z - vb.z)};

	f3 mn = vo, mx = vo;
#pragma unroll
	for (int d = 0; d < 6; ++d) {  // order -x,+x,-y,+y,-z,+z (Kernel.cu:410-421)
		const int s = (d & 1) ? 1 : -1;
		const int t = tap_index(g, s_nbr, L.org, ci + (d < 2 ? s : 0), cj + ((d >> 1) == 1 ? s : 0), ck + (d >= 4 ? s : 0));
		const f3 nv = ld0v(ux, uy, uz, t);
		mn.x = fminf(mn.x, nv.x);
		mx.x = fmaxf(mx.x, nv.x);
		mn.y = fminf(mn.y, nv.y);
		mx.y = fmaxf(mx.y, nv.y);
		mn.z = fminf(mn.z, nv.z);
		mx.z = fmaxf(mx.z, nv.z);
	}
	mn.x = fminf(mn.x, vf.x);
	mx.x = fmaxf(mx.x, vf.x);
	mn.y = fminf(mn.y, vf.y);
	mx.y = fmaxf(mx.y, vf.y);
	mn.z = fminf(mn.z, vf.z);
	mx.z = fmaxf(mx.z, vf.z);
	vc.x = fmaxf(mn.x, fminf(vc.x, mx.x));
	vc.y = fmaxf(mn.y, fminf(vc.y, mx.y));
	vc.z = fmaxf(mn.z, fminf(vc.z, mx.z));

	if (COLL) {  // Kernel.cu:433-450
		const float sv = sdf[idx];
		if (sv < 0.0f) {
			vc.x = vc.y = vc.z = 0.0f;
		} else if (sv < 0.1f) {
			const f3 nrm = sdf_normal(g, s_nbr, L.org, sdf, ci, cj, ck, inv_dx);
			vc = no_slip_blend(vc, nrm, 1.0f - (sv / 1.5f));
		}
	}
	ox[idx] = vc.x;
	oy[idx] = vc.y;
	oz[idx] = vc.z;
}

// ---------------------------------------------------------------------------------------------------------------
// advect_scalar (reference Kernel.cu:269-352): single field, nested-lerp trilinear
// ---------------------------------------------------------------------------------------------------------------

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_scalar(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                       const float* __restrict__ uz, const float* __restrict__ in, float* __restrict__ out,
                                                       const float* __restrict__ sdf, const float scaled_dt) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const float phiOrig = in[idx];
	const f3 vc = {ux[idx], uy[idx], uz[idx]};
	float bx = px - scaled_dt * vc.x, by = py - scaled_dt * vc.y, bz = pz - scaled_dt * vc.z;
	if (COLL) {
		if (tri_f(g, s_nbr, L.org, sdf, bx, by, bz) < 0.0f) {
			bx = px;
			by = py;
			bz = pz;
		}
	}
	const float phiForward = tri_f(g, s_nbr, L.org, in, bx, by, bz);
	const f3 vf = tri_v(g, s_nbr, L.org, ux, uy, uz, bx, by, bz);
	float fx = bx + scaled_dt * vf.x, fy = by + scaled_dt * vf.y, fz = bz + scaled_dt * vf.z;
	if (COLL) {
		if (tri_f(g, s_nbr, L.org, sdf, fx, fy, fz) < 0.0f) {
			fx = bx;
			fy = by;
			fz = bz;
		}
	}
	const float phiBackward = tri_f(g, s_nbr, L.org, in, fx, fy, fz);
	const float error = phiOrig - phiBackward;
	float phiCorr = phiForward + 0.5f * error;
	float mn = phiOrig, mx = phiOrig;
#pragma unroll
	for (int d = 0; d < 6; ++d) {
		const int s = (d & 1) ? 1 : -1;
		const int t = tap_index(g, s_nbr, L.org, ci + (d < 2 ? s : 0), cj + ((d >> 1) == 1 ? s : 0), ck + (d >= 4 ? s : 0));
		const float nv = ld0(in, t);
		mn = fminf(mn, nv);
		mx = fmaxf(mx, nv);
	}
	mn = fminf(mn, phiForward);
	mx = fmaxf(mx, phiForward);
	out[idx] = fmaxf(mn, fminf(phiCorr, mx));
}

// ---------------------------------------------------------------------------------------------------------------
// advect_scalars (reference Kernel.cu:118-266): one backtrace shared by up to HNS_MAX_SCALARS fields,
// weight-product trilinear, out-of-domain taps read ELEMENT 0 (Kernel.cu:133,192,225)
// ---------------------------------------------------------------------------------------------------------------

#define HNS_MAX_SCALARS 8
struct ScalarPtrs {
	const float* in[HNS_MAX_SCALARS];
	float* out[HNS_MAX_SCALARS];
	int n;
};

__device__ __forceinline__ void setup_interp(const GridDev& g, const int* s_nbr, const int4 org, float x, float y, float z, int (&ix)[8],
                                             float (&w)[8]) {
	// setupInterpolation (Kernel.cu:163-196): order 000,100,010,110,001,101,011,111 in (x,y,z)
	const int i0 = __float2int_rd(x), j0 = __float2int_rd(y), k0 = __float2int_rd(z);
	const float tx = x - (float)i0, ty = y - (float)j0, tz = z - (float)k0;
	const float itx = 1.0f - tx, ity = 1.0f - ty, itz = 1.0f - tz;
	const float w00 = itx * ity, w10 = tx * ity, w01 = itx * ty, w11 = tx * ty;
	w[0] = w00 * itz;
	w[1] = w10 * itz;
	w[2] = w01 * itz;
	w[3] = w11 * itz;
	w[4] = w00 * tz;
	w[5] = w10 * tz;
	w[6] = w01 * tz;
	w[7] = w11 * tz;
	int t[8];
	tap8(g, s_nbr, org, i0, j0, k0, t);  // t[di*4+dj*2+dk]
	ix[0] = t[0];
	ix[1] = t[4];
	ix[2] = t[2];
	ix[3] = t[6];
	ix[4] = t[1];
	ix[5] = t[5];
	ix[6] = t[3];
	ix[7] = t[7];
#pragma unroll
	for (int q = 0; q < 8; ++q) ix[q] = ix[q] < 0 ? g.oob : ix[q];
}

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_scalars(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                        const float* __restrict__ uz, const ScalarPtrs P, const float* __restrict__ sdf,
                                                        const float scaled_dt) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const f3 vc = {ux[idx], uy[idx], uz[idx]};
	float bx = px - scaled_dt * vc.x, by = py - scaled_dt * vc.y, bz = pz - scaled_dt * vc.z;
	if (COLL) {  // tested twice in the reference (Kernel.cu:142-155); the second test cannot change the outcome of the first
		if (tri_f(g, s_nbr, L.org, sdf, bx, by, bz) < 0.0f) {
			bx = px;
			by = py;
			bz = pz;
		}
		if (tri_f(g, s_nbr, L.org, sdf, bx, by, bz) < 0.0f) {
			bx = px;
			by = py;
			bz = pz;
		}
	}
	int bi[8], fi[8];
	float bw[8], fw[8];
	setup_interp(g, s_nbr, L.org, bx, by, bz, bi, bw);
	f3 vf = {0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int q = 0; q < 8; ++q) {  // velF = velF + v * w (Kernel.cu:201-206), unfused
		vf.x = vf.x + bw[q] * ux[bi[q]];
		vf.y = vf.y + bw[q] * uy[bi[q]];
		vf.z = vf.z + bw[q] * uz[bi[q]];
	}
	float fx = bx + scaled_dt * vf.x, fy = by + scaled_dt * vf.y, fz = bz + scaled_dt * vf.z;
	if (COLL) {
		if (tri_f(g, s_nbr, L.org, sdf, fx, fy, fz) < 0.0f) {
			fx = bx;
			fy = by;
			fz = bz;
		}
	}
	setup_interp(g, s_nbr, L.org, fx, fy, fz, fi, fw);
	int nb[6];
#pragma unroll
	for (int d = 0; d < 6; ++d) {  // -x,+x,-y,+y,-z,+z (Kernel.cu:219)
		const int s = (d & 1) ? 1 : -1;
		const int t = tap_index(g, s_nbr, L.org, ci + (d < 2 ? s : 0), cj + ((d >> 1) == 1 ? s : 0), ck + (d >= 4 ? s : 0));
		nb[d] = t < 0 ? g.oob : t;
	}
	for (int s = 0; s < P.n; ++s) {
		const float* __restrict__ in = P.in[s];
		const float phiOrig = in[idx];
		float phiF = 0.0f, phiB = 0.0f;
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			phiF = __fmaf_rn(in[bi[q]], bw[q], phiF);
			phiB = __fmaf_rn(in[fi[q]], fw[q], phiB);
		}
		const float error = phiOrig - phiB;
		const float phiCorr = __fmaf_rn(0.5f, error, phiF);
		float mn = phiOrig, mx = phiOrig;
#pragma unroll
		for (int d = 0; d < 6; ++d) {
			const float v = in[nb[d]];
			mn = fminf(mn, v);
			mx = fmaxf(mx, v);
		}
		mn = fminf(mn, phiF);
		mx = fmaxf(mx, phiF);
		P.out[s][idx] = fmaxf(mn, fminf(phiCorr, mx));
	}
}

// ---------------------------------------------------------------------------------------------------------------
// 6-neighbour access of a leaf-dense float field for thread n of the leaf's workgroup
// ---------------------------------------------------------------------------------------------------------------

// value at (x+dx, y+dy, z+dz) for a unit step along one axis; faces resolve through the neighbour table
template <int AXIS, int DIR>
__device__ __forceinline__ float nbr_val(const float* __restrict__ f, const int* s_nbr, int leaf, int n) {
	constexpr int shift = AXIS == 0 ? 6 : (AXIS == 1 ? 3 : 0);
	constexpr int stride = 1 << shift;
	const int c = (n >> shift) & 7;
	if (DIR > 0) {
		if (c != 7) return f[leaf * 512 + n + stride];
		const int nl = s_nbr[13 + (AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1))];
		return nl < 0 ? 0.0f : f[nl * 512 + n - 7 * stride];
	} else {
		if (c != 0) return f[leaf * 512 + n - stride];
		const int nl = s_nbr[13 - (AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1))];
		return nl < 0 ? 0.0f : f[nl * 512 + n + 7 * stride];
	}
}

// ---------------------------------------------------------------------------------------------------------------
// divergence (reference Kernel.cu:499-519 and :455-496)
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(512) void k_divergence(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                    const float* __restrict__ uz, float* __restrict__ div, const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float cx = ux[idx], cy = uy[idx], cz = uz[idx];
	const float xp = (cx + nbr_val<0, 1>(ux, s_nbr, L.leaf, n)) * 0.5f;
	const float xm = (cx + nbr_val<0, -1>(ux, s_nbr, L.leaf, n)) * 0.5f;
	const float yp = (cy + nbr_val<1, 1>(uy, s_nbr, L.leaf, n)) * 0.5f;
	const float ym = (cy + nbr_val<1, -1>(uy, s_nbr, L.leaf, n)) * 0.5f;
	const float zp = (cz + nbr_val<2, 1>(uz, s_nbr, L.leaf, n)) * 0.5f;
	const float zm = (cz + nbr_val<2, -1>(uz, s_nbr, L.leaf, n)) * 0.5f;
	div[idx] = (xp - xm + yp - ym + zp - zm) * inv_dx;
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR, two-launch form: one colour in place (reference Kernel.cu:591-623 / :521-588)
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ float sor_update(float pxp, float pxm, float pyp, float pym, float pzp, float pzm, float divVal, float pOld,
                                            float dx2, float omega) {
	constexpr float inv6 = 0.166666667f;
	const float pGS = ((pxp + pxm + pyp + pym + pzp + pzm) - divVal * dx2) * inv6;  // Kernel.cu:621
	return pOld + omega * (pGS - pOld);                                              // Kernel.cu:622
}

__global__ __launch_bounds__(256) void k_rbgs_color(const GridDev g, const float* __restrict__ div, float* p, const float dx2, const float omega,
                                                    const int color) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int t = threadIdx.x;  // one thread per voxel of this colour
	const int x = t >> 5, y = (t >> 2) & 7;
	const int z = 2 * (t & 3) + ((x + y + color) & 1);  // origins are 8-aligned, so global parity == local parity
	const int n = (x << 6) | (y << 3) | z;
	const int idx = L.leaf * 512 + n;
	const float pxp = nbr_val<0, 1>(p, s_nbr, L.leaf, n), pxm = nbr_val<0, -1>(p, s_nbr, L.leaf, n);
	const float pyp = nbr_val<1, 1>(p, s_nbr, L.leaf, n), pym = nbr_val<1, -1>(p, s_nbr, L.leaf, n);
	const float pzp = nbr_val<2, 1>(p, s_nbr, L.leaf, n), pzm = nbr_val<2, -1>(p, s_nbr, L.leaf, n);
	p[idx] = sor_update(pxp, pxm, pyp, pym, pzp, pzm, div[idx], p[idx], dx2, omega);
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR, fused form: one launch = one full (red, black) iteration, p_in -> p_out
// ---------------------------------------------------------------------------------------------------------------
//
// LDS tile T[12][12][12] holds p_in over the leaf (offset +2) plus: the two nearest voxel layers of each face
// neighbour and the nearest voxel line of each edge neighbour. Phase R updates the 256 red voxels of the leaf and the
// 6*32 red voxels directly across each face (only where that neighbour leaf exists; outside the domain p stays 0);
// a halo red voxel (x=-1,y,z) needs p at (-2,y,z), (0,y,z), (-1,y+-1,z), (-1,y,z+-1): face layers and edge lines, never
// corners. Phase B updates the 256 black voxels from the new red values. Thread t owns the z-adjacent pair
// (2t, 2t+1) of the leaf: it loads p and div for the pair as one 8-byte access each and stores both new values
// as one 8-byte access, so every global access of the leaf payload is a contiguous 2 KB run per workgroup.
// Arithmetic per voxel is exactly sor_update(); the recomputed halo reds see the same inputs as the neighbouring
// workgroup's own update, so the result is bit-identical to the two-launch form.

#define T_IDX(x, y, z) (((x) + 2) * 144 + ((y) + 2) * 12 + ((z) + 2))

__global__ __launch_bounds__(256) void k_rbgs_fused(const GridDev g, const float* __restrict__ div, const float* __restrict__ p_in,
                                                    float* __restrict__ p_out, const float dx2, const float omega) {
	__shared__ float T[12 * 12 * 12];
	__shared__ int s_nbr[27];
	const int t = threadIdx.x;
	const int leaf = g.sched ? g.sched[blockIdx.x] : blockIdx.x;
	if (t < 27) s_nbr[t] = g.nbr27[leaf * 27 + t];

	// own leaf: pair (2t, 2t+1) = voxels (x, y, 2zp) and (x, y, 2zp+1)
	const int x = t >> 5, y = (t >> 2) & 7, zp = t & 3;
	const float2 pown = reinterpret_cast<const float2*>(p_in + leaf * 512)[t];
	const float2 down = reinterpret_cast<const float2*>(div + leaf * 512)[t];
	T[T_IDX(x, y, 2 * zp)] = pown.x;
	T[T_IDX(x, y, 2 * zp + 1)] = pown.y;
	__syncthreads();  // s_nbr visible

	{  // -x / +x faces: layers x=6,7 of the -x neighbour (offsets 384..511), x=0,1 of the +x neighbour (0..127)
		const int side = t >> 7, e = t & 127;
		const int nl = s_nbr[side ? 22 : 4];
		const float v = nl < 0 ? 0.0f : p_in[nl * 512 + (side ? 0 : 384) + e];
		T[T_IDX((side ? 8 : -2) + (e >> 6), (e >> 3) & 7, e & 7)] = v;
	}
	{  // -y / +y faces: rows y=6,7 / y=0,1
		const int side = t >> 7, e = t & 127;
		const int xx = e >> 4, yy = (e >> 3) & 1, zz = e & 7;
		const int nl = s_nbr[side ? 16 : 10];
		const float v = nl < 0 ? 0.0f : p_in[nl * 512 + xx * 64 + ((side ? 0 : 6) + yy) * 8 + zz];
		T[T_IDX(xx, (side ? 8 : -2) + yy, zz)] = v;
	}
	{  // -z / +z faces: z=6,7 / z=0,1
		const int side = t >> 7, e = t & 127;
		const int xx = e >> 4, yy = (e >> 1) & 7, zz = e & 1;
		const int nl = s_nbr[side ? 14 : 12];
		const float v = nl < 0 ? 0.0f : p_in[nl * 512 + xx * 64 + yy * 8 + (side ? 0 : 6) + zz];
		T[T_IDX(xx, yy, (side ? 8 : -2) + zz)] = v;
	}
	if (t < 96) {  // 12 edge lines of 8 voxels
		const int e = t >> 3, i = t & 7;
		const int grp = e >> 2, a = (e >> 1) & 1, b = e & 1;  // grp 0: (x,y) edges along z; 1: (x,z) along y; 2: (y,z) along x
		const int da = a ? 1 : -1, db = b ? 1 : -1;
		const int ta = a ? 8 : -1, tb = b ? 8 : -1;  // tile coordinate of the line
		const int sa = a ? 0 : 7, sb = b ? 0 : 7;    // source coordinate inside the neighbour leaf
		int nslot, src, dst;
		if (grp == 0) {
			nslot = (da + 1) * 9 + (db + 1) * 3 + 1;
			src = sa * 64 + sb * 8 + i;
			dst = T_IDX(ta, tb, i);
		} else if (grp == 1) {
			nslot = (da + 1) * 9 + 3 + (db + 1);
			src = sa * 64 + i * 8 + sb;
			dst = T_IDX(ta, i, tb);
		} else {
			nslot = 9 + (da + 1) * 3 + (db + 1);
			src = i * 64 + sa * 8 + sb;
			dst = T_IDX(i, ta, tb);
		}
		const int nl = s_nbr[nslot];
		T[dst] = nl < 0 ? 0.0f : p_in[nl * 512 + src];
	}

	// halo red site owned by this thread (t < 192): face f, in-face cell (a, b) with a+b+c0 even
	int hx = 0, hy = 0, hz = 0, hleaf = -1;
	float hdiv = 0.0f;
	if (t < 192) {
		const int f = t >> 5, r = t & 31;
		const int axis = f >> 1, side = f & 1;
		const int c0 = side ? 8 : -1;
		const int a = r >> 2;
		const int b = 2 * (r & 3) + ((a + (side ? 0 : 1)) & 1);
		const int cs = side ? 0 : 7;  // coordinate inside the neighbour leaf
		int src;
		if (axis == 0) {
			hx = c0, hy = a, hz = b;
			src = cs * 64 + a * 8 + b;
			hleaf = s_nbr[side ? 22 : 4];
		} else if (axis == 1) {
			hx = a, hy = c0, hz = b;
			src = a * 64 + cs * 8 + b;
			hleaf = s_nbr[side ? 16 : 10];
		} else {
			hx = a, hy = b, hz = c0;
			src = a * 64 + b * 8 + cs;
			hleaf = s_nbr[side ? 14 : 12];
		}
		if (hleaf >= 0) hdiv = div[hleaf * 512 + src];
	}
	__syncthreads();  // tile complete

	// ---- phase R: red = (x+y+z) even ----
	const int zr = 2 * zp + ((x + y) & 1);  // red voxel of the pair
	const int zb = zr ^ 1;                   // black voxel of the pair
	const int cr = T_IDX(x, y, zr);
	const float p_red_old = (zr & 1) ? pown.y : pown.x;
	const float d_red = (zr & 1) ? down.y : down.x;
	const float p_red = sor_update(T[cr + 144], T[cr - 144], T[cr + 12], T[cr - 12], T[cr + 1], T[cr - 1], d_red, p_red_old, dx2, omega);
	float h_new = 0.0f;
	const int ch = T_IDX(hx, hy, hz);
	if (hleaf >= 0) h_new = sor_update(T[ch + 144], T[ch - 144], T[ch + 12], T[ch - 12], T[ch + 1], T[ch - 1], hdiv, T[ch], dx2, omega);
	// red sites only read black sites, so the in-place tile update needs no extra barrier before the writes
	T[cr] = p_red;
	if (hleaf >= 0) T[ch] = h_new;
	__syncthreads();

	// ---- phase B: black = (x+y+z) odd, reads the new reds ----
	const int cb = T_IDX(x, y, zb);
	const float p_blk_old = (zb & 1) ? pown.y : pown.x;
	const float d_blk = (zb & 1) ? down.y : down.x;
	const float p_blk = sor_update(T[cb + 144], T[cb - 144], T[cb + 12], T[cb - 12], T[cb + 1], T[cb - 1], d_blk, p_blk_old, dx2, omega);

	float2 o;
	o.x = (zr & 1) ? p_blk : p_red;
	o.y = (zr & 1) ? p_red : p_blk;
	reinterpret_cast<float2*>(p_out + leaf * 512)[t] = o;
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR, fused form, ONE WAVE PER LEAF (the production kernel)
// ---------------------------------------------------------------------------------------------------------------
//
// Same algorithm and the same per-voxel arithmetic as k_rbgs_fused above, reorganised for the CDNA4 wave: the 64 lanes
// of one wave own the 64 z-rows of a leaf (lane = x*8+y, the memory order, so the 2 KB payload is read and written as
// two 16-byte accesses per lane, fully coalesced) and keep their row in registers. No workgroup barrier exists: the
// wave's private LDS tile only carries rows between lanes. Work per lane:
//   * own row: p[-2..9] (z halo from the +-z neighbour leaves) and div[0..7] in registers; lateral neighbours are the
//     rows of lanes x+-1 / y+-1 (or of the face-neighbour leaves), read from LDS as 2 x ds_read_b128 each;
//   * one z-halo red voxel: (x,y,-1) if x+y is odd, else (x,y,8);
//   * lanes 0..31 additionally recompute the red voxels of one face-adjacent halo row each (4 faces x 8 rows); the
//     depth-2 row behind it sits in a side area of the tile so that all four lateral reads have the same shape.
// Every candidate is evaluated for all 8 z of a row and accepted by colour, which keeps the code free of
// lane-dependent register indexing; rejected candidates never reach memory, so the result is bit-identical to
// the two-launch form.
//
// LDS tile: rows (x',y') in [-1,8]^2 -> R = (x'+1)*10 + (y'+1), plus rows 100..131 for the depth-2 rows.
// Row R occupies floats [4+12R-1, 4+12R+8]: z = -1 .. 8 (z=0 is 16-byte aligned; 48-byte row stride).

#define W_OFF(R, z) (4 + (R) * 12 + (z))
#define W_ROW(xp, yp) (((xp) + 1) * 10 + ((yp) + 1))
#define W_FLOATS (4 + 132 * 12)

struct Row8 {
	float v[8];
};

__device__ __forceinline__ Row8 lds_row(const float* T, int R) {
	const float4 a = *reinterpret_cast<const float4*>(T + W_OFF(R, 0));
	const float4 b = *reinterpret_cast<const float4*>(T + W_OFF(R, 4));
	Row8 r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
	return r;
}

__device__ __forceinline__ void lds_put_row(float* T, int R, const float (&v)[8]) {
	*reinterpret_cast<float4*>(T + W_OFF(R, 0)) = make_float4(v[0], v[1], v[2], v[3]);
	*reinterpret_cast<float4*>(T + W_OFF(R, 4)) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ Row8 glb_row(const float* __restrict__ f, int leaf, int row) {
	Row8 r;
	if (leaf < 0) {
#pragma unroll
		for (int z = 0; z < 8; ++z) r.v[z] = 0.0f;
		return r;
	}
	const float4* q = reinterpret_cast<const float4*>(f + (size_t)leaf * 512 + row * 8);
	const float4 a = q[0], b = q[1];
	r.v[0] = a.x, r.v[1] = a.y, r.v[2] = a.z, r.v[3] = a.w, r.v[4] = b.x, r.v[5] = b.y, r.v[6] = b.z, r.v[7] = b.w;
	return r;
}

// SOR candidates of a whole z-row: c[z+1] = centre row z = -1..8 (10 values), lateral rows xp/xm/yp/ym, d = div row
__device__ __forceinline__ void row_candidates(const Row8& xp, const Row8& xm, const Row8& yp, const Row8& ym, const float (&c)[10],
                                               const float (&d)[8], float dx2, float omega, float (&cand)[8]) {
#pragma unroll
	for (int z = 0; z < 8; ++z) cand[z] = sor_update(xp.v[z], xm.v[z], yp.v[z], ym.v[z], c[z + 2], c[z], d[z], c[z + 1], dx2, omega);
}

__global__ __launch_bounds__(64) void k_rbgs_wave(const GridDev g, const float* __restrict__ div, const float* __restrict__ p_in,
                                                  float* __restrict__ p_out, const float dx2, const float omega) {
	__shared__ __attribute__((aligned(16))) float T[W_FLOATS];
	const int l = threadIdx.x;
	// per-block record {leaf, nbr27[27]} in launch order: one dependent scalar fetch instead of sched -> nbr27
	const int* __restrict__ rec = g.blk + (size_t)blockIdx.x * 28;
	const int leaf = __builtin_amdgcn_readfirstlane(rec[0]);
	const int n_xm = __builtin_amdgcn_readfirstlane(rec[1 + 4]), n_xp = __builtin_amdgcn_readfirstlane(rec[1 + 22]);
	const int n_ym = __builtin_amdgcn_readfirstlane(rec[1 + 10]), n_yp = __builtin_amdgcn_readfirstlane(rec[1 + 16]);
	const int n_zm = __builtin_amdgcn_readfirstlane(rec[1 + 12]), n_zp = __builtin_amdgcn_readfirstlane(rec[1 + 14]);

	const int x = l >> 3, y = l & 7;
	const int par = (x + y) & 1;  // 0: even z are red; 1: odd z are red

	// ---- issue every global load up front ----
	const Row8 P = glb_row(p_in, leaf, l);
	const Row8 D = glb_row(div, leaf, l);
	float2 zlo = make_float2(0.0f, 0.0f), zhi = make_float2(0.0f, 0.0f);  // p(x,y,-2..-1), p(x,y,8..9)
	if (n_zm >= 0) zlo = *reinterpret_cast<const float2*>(p_in + (size_t)n_zm * 512 + l * 8 + 6);
	if (n_zp >= 0) zhi = *reinterpret_cast<const float2*>(p_in + (size_t)n_zp * 512 + l * 8);
	const int n_zh = par ? n_zm : n_zp;  // leaf of this lane's z-halo red voxel: (x,y,-1) if par else (x,y,8)
	float d_zh = 0.0f;
	if (n_zh >= 0) d_zh = div[(size_t)n_zh * 512 + l * 8 + (par ? 7 : 0)];

	// halo rows (lanes 0..31): face f, row i. A = adjacent row (recomputed), B = the row behind it
	const int f = (l >> 3) & 3, i = l & 7;
	const int n_f = f == 0 ? n_xm : (f == 1 ? n_xp : (f == 2 ? n_ym : n_yp));
	const int srcA = f == 0 ? 56 + i : (f == 1 ? i : (f == 2 ? i * 8 + 7 : i * 8));
	const int srcB = f == 0 ? 48 + i : (f == 1 ? 8 + i : (f == 2 ? i * 8 + 6 : i * 8 + 1));
	const int ax = f == 0 ? -1 : (f == 1 ? 8 : i), ay = f == 2 ? -1 : (f == 3 ? 8 : i);
	const bool halo_lane = l < 32;
	const int n_h = halo_lane ? n_f : -1;
	const Row8 HA = glb_row(p_in, n_h, srcA);
	const Row8 HB = glb_row(p_in, n_h, srcB);
	const Row8 HD = glb_row(div, n_h, srcA);

	// edge rows along z: tile rows (-1,-1), (-1,8), (8,-1), (8,8) (lanes 32..35)
	const int ea = (l >> 1) & 1, eb = l & 1;
	const bool erow_lane = (l >> 2) == 8;
	const int n_er = erow_lane ? rec[1 + (ea ? 2 : 0) * 9 + (eb ? 2 : 0) * 3 + 1] : -1;
	const Row8 ER = glb_row(p_in, n_er, (ea ? 0 : 7) * 8 + (eb ? 0 : 7));

	// edge singles: lines 0..3 = (x,z) edges along y, lines 4..7 = (y,z) edges along x; one voxel per lane
	const int ln = l >> 3, sa = (ln >> 1) & 1, sb = ln & 1;
	const int ta = sa ? 8 : -1, tb = sb ? 8 : -1;  // tile coordinates of the line
	const int ca = sa ? 0 : 7, cb = sb ? 0 : 7;    // coordinates inside the neighbour leaf
	const int e_slot = ln < 4 ? (sa ? 2 : 0) * 9 + 3 + (sb ? 2 : 0) : 9 + (sa ? 2 : 0) * 3 + (sb ? 2 : 0);
	const int e_src = ln < 4 ? ca * 64 + i * 8 + cb : i * 64 + ca * 8 + cb;
	const int e_row = ln < 4 ? W_ROW(ta, i) : W_ROW(i, ta);
	const int n_e = rec[1 + e_slot];
	float e_val = 0.0f;
	if (n_e >= 0) e_val = p_in[(size_t)n_e * 512 + e_src];

	// ---- stage rows in LDS ----
	const int R_own = W_ROW(x, y);
	lds_put_row(T, R_own, P.v);
	T[W_OFF(R_own, -1)] = zlo.y;
	T[W_OFF(R_own, 8)] = zhi.x;
	const int R_A = W_ROW(ax, ay), R_B = 100 + l;
	if (halo_lane) {
		lds_put_row(T, R_A, HA.v);
		lds_put_row(T, R_B, HB.v);
	}
	if (erow_lane) lds_put_row(T, W_ROW(ea ? 8 : -1, eb ? 8 : -1), ER.v);
	T[W_OFF(e_row, tb)] = e_val;
	__syncthreads();  // single-wave workgroup: orders the LDS traffic, no cross-wave rendezvous

	// ---- phase R ----
	// (a) halo rows
	float hnew[8];
	if (halo_lane) {
		const Row8 hxm = lds_row(T, f == 0 ? R_B : W_ROW(ax - 1, ay));
		const Row8 hxp = lds_row(T, f == 1 ? R_B : W_ROW(ax + 1, ay));
		const Row8 hym = lds_row(T, f == 2 ? R_B : W_ROW(ax, ay - 1));
		const Row8 hyp = lds_row(T, f == 3 ? R_B : W_ROW(ax, ay + 1));
		const float hc[10] = {T[W_OFF(R_A, -1)], HA.v[0], HA.v[1], HA.v[2], HA.v[3], HA.v[4], HA.v[5], HA.v[6], HA.v[7], T[W_OFF(R_A, 8)]};
		float cand[8];
		row_candidates(hxp, hxm, hyp, hym, hc, HD.v, dx2, omega, cand);
		const int hpar = (ax + ay) & 1;
#pragma unroll
		for (int z = 0; z < 8; ++z) hnew[z] = (((hpar + z) & 1) == 0 && n_h >= 0) ? cand[z] : HA.v[z];
	}
	// (b) own row + z-halo voxel
	const int R_xm = W_ROW(x - 1, y), R_xp = W_ROW(x + 1, y), R_ym = W_ROW(x, y - 1), R_yp = W_ROW(x, y + 1);
	float c[10] = {zlo.y, P.v[0], P.v[1], P.v[2], P.v[3], P.v[4], P.v[5], P.v[6], P.v[7], zhi.x};  // z = -1..8
	{
		const Row8 xm = lds_row(T, R_xm), xp = lds_row(T, R_xp), ym = lds_row(T, R_ym), yp = lds_row(T, R_yp);
		float cand[8];
		row_candidates(xp, xm, yp, ym, c, D.v, dx2, omega, cand);
		// z-halo red voxel at zh = par ? -1 : 8
		const int zh = par ? -1 : 8;
		const float zc = sor_update(T[W_OFF(R_xp, zh)], T[W_OFF(R_xm, zh)], T[W_OFF(R_yp, zh)], T[W_OFF(R_ym, zh)], par ? P.v[0] : zhi.y,
		                            par ? zlo.x : P.v[7], d_zh, par ? zlo.y : zhi.x, dx2, omega);
#pragma unroll
		for (int z = 0; z < 8; ++z) c[z + 1] = (((par + z) & 1) == 0) ? cand[z] : c[z + 1];
		if (n_zh >= 0) {
			if (par) c[0] = zc;
			else c[9] = zc;
		}
	}
	__syncthreads();  // all phase-R reads done before the tile is overwritten with the new reds
	{
		const float own[8] = {c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
		lds_put_row(T, R_own, own);
		if (halo_lane) lds_put_row(T, R_A, hnew);
	}
	__syncthreads();

	// ---- phase B ----
	{
		const Row8 xm = lds_row(T, R_xm), xp = lds_row(T, R_xp), ym = lds_row(T, R_ym), yp = lds_row(T, R_yp);
		float cand[8];
		row_candidates(xp, xm, yp, ym, c, D.v, dx2, omega, cand);
		float4 o0, o1;
		o0.x = par ? cand[0] : c[1];
		o0.y = par ? c[2] : cand[1];
		o0.z = par ? cand[2] : c[3];
		o0.w = par ? c[4] : cand[3];
		o1.x = par ? cand[4] : c[5];
		o1.y = par ? c[6] : cand[5];
		o1.z = par ? cand[6] : c[7];
		o1.w = par ? c[8] : cand[7];
		float4* q = reinterpret_cast<float4*>(p_out + (size_t)leaf * 512 + l * 8);
		q[0] = o0;
		q[1] = o1;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR, fused form, one wave per PAIR of z-adjacent leaves (the production kernel for paired leaves)
// ---------------------------------------------------------------------------------------------------------------
//
// z is the fastest-varying index of the leaf payload, so the +-z faces are the expensive halo: 8-byte pieces at a
// 32-byte stride (16 cache lines for 512 useful bytes). k_rbgs_wave spends most of its time in the texture addresser /
// L1 on exactly those accesses (profiles/r01_v2_*). Here one wave owns two leaves stacked along z: the face between
// them never leaves the registers, the strided z-halo loads are halved, and the recomputation of the face-adjacent
// halo rows (32 rows per leaf) is spread over all 64 lanes (lanes 0-31: lower leaf, 32-63: upper leaf).
// Arithmetic per voxel is sor_update(), exactly as in the other forms.
//
// LDS rows are kept as separate 16-byte halves LO (z 0..3) / HI (z 4..7) indexed by a row number chosen so that the
// lateral-neighbour reads of the 64 lanes are linear in the lane id (conflict-free ds_read_b128):
//   (x', y') x' in -1..8, y' in 0..7 -> 8*(x'+1) + y'          (x'=-1 / 8 are the -x / +x face rows)
//   (x', -1) -> 87 + 8*x'      (== row (x',0) - 1   mod 16)
//   (x',  8) -> 96 + 8*x'      (== row (x',7) + 1   mod 16)
//   edge rows (-1,-1) (-1,8) (8,-1) (8,8) -> 81..84; depth-2 rows of halo lane h -> 89 + 8*(h/6) + h%6
// ZM / ZP hold the z=-1 values of the lower tile and the z=8 values of the upper tile (core rows and face rows).

#define PR_ROWS 153

typedef float v2f __attribute__((ext_vector_type(2)));

// a z-row as four (even z, odd z) pairs: the SOR arithmetic below is written on pairs so that it compiles to packed
// v_pk_add_f32 / v_pk_mul_f32 (two voxels per VALU instruction); -ffp-contract=off keeps every operation separate
struct RowP {
	v2f q[4];
};

struct PairTile {
	float4 LO[2][PR_ROWS];
	float4 HI[2][PR_ROWS];
	float ZM[PR_ROWS];
	float ZP[PR_ROWS];
};

__device__ __forceinline__ RowP pt_row(const PairTile& S, int k, int R) {
	const float4 a = S.LO[k][R], b = S.HI[k][R];
	RowP r;
	r.q[0] = v2f{a.x, a.y}, r.q[1] = v2f{a.z, a.w}, r.q[2] = v2f{b.x, b.y}, r.q[3] = v2f{b.z, b.w};
	return r;
}
__device__ __forceinline__ void pt_put(PairTile& S, int k, int R, const RowP& r) {
	S.LO[k][R] = make_float4(r.q[0].x, r.q[0].y, r.q[1].x, r.q[1].y);
	S.HI[k][R] = make_float4(r.q[2].x, r.q[2].y, r.q[3].x, r.q[3].y);
}

// Row `row` of leaf `leaf` (-1 = absent -> zeros). Branch-free: an absent leaf reads leaf 0 and discards the data.
__device__ __forceinline__ RowP glb_rowp(const float* __restrict__ f, int leaf, int row) {
	const float4* q = reinterpret_cast<const float4*>(f + (size_t)(leaf < 0 ? 0 : leaf) * 512 + row * 8);
	const float4 a = q[0], b = q[1];
	const bool ok = leaf >= 0;
	RowP r;
	r.q[0] = v2f{ok ? a.x : 0.0f, ok ? a.y : 0.0f};
	r.q[1] = v2f{ok ? a.z : 0.0f, ok ? a.w : 0.0f};
	r.q[2] = v2f{ok ? b.x : 0.0f, ok ? b.y : 0.0f};
	r.q[3] = v2f{ok ? b.z : 0.0f, ok ? b.w : 0.0f};
	return r;
}

// sor_update() on two voxels at once; same operation order per element (Kernel.cu:621-622)
__device__ __forceinline__ v2f sor2(v2f pxp, v2f pxm, v2f pyp, v2f pym, v2f pzp, v2f pzm, v2f d, v2f pold, float dx2, float omega) {
	constexpr float inv6 = 0.166666667f;
	const v2f pGS = ((pxp + pxm + pyp + pym + pzp + pzm) - d * dx2) * inv6;
	return pold + omega * (pGS - pold);
}

// One colour of a whole z-row: candidates for all 8 voxels, then keep the even-z ones (take_even) or the odd-z ones.
// c = the row itself, below / above = its z=-1 / z=8 neighbours. `valid` false leaves the row untouched.
__device__ __forceinline__ RowP row_sweep(const RowP& xp, const RowP& xm, const RowP& yp, const RowP& ym, const RowP& c, float below, float above,
                                          const RowP& d, float dx2, float omega, bool take_even, bool valid) {
	RowP out;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const v2f zp = v2f{c.q[j].y, j < 3 ? c.q[j < 3 ? j + 1 : 3].x : above};
		const v2f zm = v2f{j > 0 ? c.q[j > 0 ? j - 1 : 0].y : below, c.q[j].x};
		const v2f cand = sor2(xp.q[j], xm.q[j], yp.q[j], ym.q[j], zp, zm, d.q[j], c.q[j], dx2, omega);
		out.q[j].x = (valid && take_even) ? cand.x : c.q[j].x;
		out.q[j].y = (valid && !take_even) ? cand.y : c.q[j].y;
	}
	return out;
}

// Per-lane constants of k_rbgs_pair (halo-row duty of lane l), generated at compile time so that the kernel reads them
// with two 16-byte loads instead of ~100 instructions of lane-dependent index arithmetic.
struct PairLane {
	int slotF, srcA, srcB, RA, RB, H_xm, H_xp, H_ym, H_yp, slotE, hpar, pad;
};
struct PairLaneTab {
	PairLane t[64];
};
constexpr PairLaneTab make_pair_lane_tab() {
	PairLaneTab T{};
	for (int l = 0; l < 64; ++l) {
		const int w = l >> 5, h = l & 31, f = h >> 3, i = h & 7;
		PairLane& e = T.t[l];
		e.slotF = f == 0 ? 4 : (f == 1 ? 22 : (f == 2 ? 10 : 16));
		e.srcA = f == 0 ? 56 + i : (f == 1 ? i : (f == 2 ? i * 8 + 7 : i * 8));
		e.srcB = f == 0 ? 48 + i : (f == 1 ? 8 + i : (f == 2 ? i * 8 + 6 : i * 8 + 1));
		e.RA = f == 0 ? i : (f == 1 ? 72 + i : (f == 2 ? 87 + 8 * i : 96 + 8 * i));
		e.RB = 89 + 8 * (h / 6) + (h % 6);
		e.H_xm = f == 0 ? e.RB : (f == 1 ? 64 + i : (i == 0 ? (f == 2 ? 81 : 82) : (f == 2 ? 87 + 8 * (i - 1) : 96 + 8 * (i - 1))));
		e.H_xp = f == 1 ? e.RB : (f == 0 ? 8 + i : (i == 7 ? (f == 2 ? 83 : 84) : (f == 2 ? 87 + 8 * (i + 1) : 96 + 8 * (i + 1))));
		e.H_ym = f == 2 ? e.RB : (f == 3 ? 8 * (i + 1) + 7 : (i == 0 ? (f == 0 ? 81 : 83) : (f == 0 ? i - 1 : 72 + i - 1)));
		e.H_yp = f == 3 ? e.RB : (f == 2 ? 8 * (i + 1) : (i == 7 ? (f == 0 ? 82 : 84) : (f == 0 ? i + 1 : 72 + i + 1)));
		e.slotE = e.slotF + (w ? 1 : -1);              // the face neighbour one leaf further along -z (lower leaf) / +z (upper leaf)
		e.hpar = (i + ((f & 1) ? 0 : 1)) & 1;          // parity of the halo row's x+y: faces -x,-y sit at coordinate -1
		e.pad = 0;
	}
	return T;
}
__device__ const PairLaneTab g_pair_lane_tab = make_pair_lane_tab();

__global__ __launch_bounds__(64) void k_rbgs_pair(const int* __restrict__ pairs, const float* __restrict__ div, const float* __restrict__ p_in,
                                                  float* __restrict__ p_out, const float dx2, const float omega) {
	__shared__ __attribute__((aligned(16))) PairTile S;
	const int l = threadIdx.x;
	// record: {leaf0, nbr27 of leaf0, leaf1, nbr27 of leaf1}; leaf1 is the +z neighbour of leaf0
	const int* __restrict__ rec = pairs + (size_t)blockIdx.x * 56;
	const int leaf0 = __builtin_amdgcn_readfirstlane(rec[0]), leaf1 = __builtin_amdgcn_readfirstlane(rec[28]);
	const int n_zm = __builtin_amdgcn_readfirstlane(rec[1 + 12]), n_zp = __builtin_amdgcn_readfirstlane(rec[28 + 1 + 14]);
	const int x = l >> 3, y = l & 7;
	const bool par = (x + y) & 1;  // false: even z red, true: odd z red (both leaves: their z origins differ by 8)

	// ---- every global load up front ----
	const RowP P0 = glb_rowp(p_in, leaf0, l), P1 = glb_rowp(p_in, leaf1, l);
	const RowP D0 = glb_rowp(div, leaf0, l), D1 = glb_rowp(div, leaf1, l);
	// p(x,y,-2..-1) below leaf0 and p(x,y,8..9) above leaf1; n_zm / n_zp are wave-uniform
	float2 zlo = *reinterpret_cast<const float2*>(p_in + (size_t)(n_zm < 0 ? 0 : n_zm) * 512 + l * 8 + 6);
	float2 zhi = *reinterpret_cast<const float2*>(p_in + (size_t)(n_zp < 0 ? 0 : n_zp) * 512 + l * 8);
	if (n_zm < 0) zlo = make_float2(0.0f, 0.0f);
	if (n_zp < 0) zhi = make_float2(0.0f, 0.0f);
	const int n_zh = par ? n_zm : n_zp;  // this lane's z-halo red voxel: below leaf0 if par, else above leaf1
	float d_zh = div[(size_t)(n_zh < 0 ? 0 : n_zh) * 512 + l * 8 + (par ? 7 : 0)];

	// halo-row duty: lanes 0..31 -> leaf0, 32..63 -> leaf1; 4 faces x 8 rows each (constants from g_pair_lane_tab)
	const int w = l >> 5;
	const int4* __restrict__ lt = reinterpret_cast<const int4*>(&g_pair_lane_tab.t[l]);
	const int4 t0 = lt[0], t1 = lt[1], t2 = lt[2];
	const int slotF = t0.x, srcA = t0.y, srcB = t0.z, RA = t0.w, RB = t1.x, H_xm = t1.y, H_xp = t1.z, H_ym = t1.w, H_yp = t2.x, slotE = t2.y;
	const bool hpar = t2.z;
	const int* __restrict__ nb = rec + 28 * w + 1;
	const int n_f = nb[slotF];
	const RowP HA = glb_rowp(p_in, n_f, srcA);
	const RowP HB = glb_rowp(p_in, n_f, srcB);
	const RowP HD = glb_rowp(div, n_f, srcA);
	// the halo row's own z-neighbour outside the pair: z=-1 for the lower leaf, z=8 for the upper leaf
	const int n_e = nb[slotE];
	float e_val = p_in[(size_t)(n_e < 0 ? 0 : n_e) * 512 + srcA * 8 + (w ? 0 : 7)];
	e_val = n_e < 0 ? 0.0f : e_val;
	// edge rows along z (lanes 0..7): tile rows (-1,-1), (-1,8), (8,-1), (8,8) of each leaf
	const int ew = (l >> 2) & 1, ea = (l >> 1) & 1, eb = l & 1;
	const int n_er = rec[28 * ew + 1 + (ea ? 2 : 0) * 9 + (eb ? 2 : 0) * 3 + 1];
	RowP ER;
	if (l < 8) ER = glb_rowp(p_in, n_er, (ea ? 0 : 7) * 8 + (eb ? 0 : 7));

	// ---- row numbers of the lane's own rows ----
	const int I = 8 * (x + 1) + y;
	const int R_xm = I - 8, R_xp = I + 8, R_ym = y == 0 ? 87 + 8 * x : I - 1, R_yp = y == 7 ? 96 + 8 * x : I + 1;

	// ---- stage ----
	pt_put(S, 0, I, P0);
	pt_put(S, 1, I, P1);
	S.ZM[I] = zlo.y;
	S.ZP[I] = zhi.x;
	pt_put(S, w, RA, HA);
	pt_put(S, w, RB, HB);
	(w ? S.ZP : S.ZM)[RA] = e_val;
	if (l < 8) pt_put(S, ew, 81 + ea * 2 + eb, ER);
	__syncthreads();

	// ---- phase R ----
	RowP hnew, c0, c1;
	float zc;
	{
		const RowP hxm = pt_row(S, w, H_xm), hxp = pt_row(S, w, H_xp), hym = pt_row(S, w, H_ym), hyp = pt_row(S, w, H_yp);
		const float other_lo = S.HI[0][RA].w, other_hi = S.LO[1][RA].x;
		const float below = w ? other_lo : e_val;  // z=-1 of this halo row
		const float above = w ? e_val : other_hi;  // z=8
		hnew = row_sweep(hxp, hxm, hyp, hym, HA, below, above, HD, dx2, omega, !hpar, n_f >= 0);
	}
	{
		const RowP xm = pt_row(S, 0, R_xm), xp = pt_row(S, 0, R_xp), ym = pt_row(S, 0, R_ym), yp = pt_row(S, 0, R_yp);
		c0 = row_sweep(xp, xm, yp, ym, P0, zlo.y, P1.q[0].x, D0, dx2, omega, !par, true);
	}
	{
		const RowP xm = pt_row(S, 1, R_xm), xp = pt_row(S, 1, R_xp), ym = pt_row(S, 1, R_ym), yp = pt_row(S, 1, R_yp);
		c1 = row_sweep(xp, xm, yp, ym, P1, P0.q[3].y, zhi.x, D1, dx2, omega, !par, true);
	}
	{
		// z-halo red voxel: (x,y,-1) under leaf0 when par, else (x,y,8) over leaf1
		const float* ZA = par ? S.ZM : S.ZP;
		zc = sor_update(ZA[R_xp], ZA[R_xm], ZA[R_yp], ZA[R_ym], par ? P0.q[0].x : zhi.y, par ? zlo.x : P1.q[3].y, d_zh, par ? zlo.y : zhi.x, dx2,
		                omega);
	}
	// values just outside each row after the red sweep
	const float below0 = (par && n_zh >= 0) ? zc : zlo.y;
	const float above1 = (!par && n_zh >= 0) ? zc : zhi.x;
	__syncthreads();  // phase-R reads complete before the rows are overwritten
	pt_put(S, 0, I, c0);
	pt_put(S, 1, I, c1);
	pt_put(S, w, RA, hnew);
	__syncthreads();

	// ---- phase B ----
	{
		const RowP xm = pt_row(S, 0, R_xm), xp = pt_row(S, 0, R_xp), ym = pt_row(S, 0, R_ym), yp = pt_row(S, 0, R_yp);
		const RowP o = row_sweep(xp, xm, yp, ym, c0, below0, c1.q[0].x, D0, dx2, omega, par, true);
		float4* q = reinterpret_cast<float4*>(p_out + (size_t)leaf0 * 512 + l * 8);
		q[0] = make_float4(o.q[0].x, o.q[0].y, o.q[1].x, o.q[1].y);
		q[1] = make_float4(o.q[2].x, o.q[2].y, o.q[3].x, o.q[3].y);
	}
	{
		const RowP xm = pt_row(S, 1, R_xm), xp = pt_row(S, 1, R_xp), ym = pt_row(S, 1, R_ym), yp = pt_row(S, 1, R_yp);
		const RowP o = row_sweep(xp, xm, yp, ym, c1, c0.q[3].y, above1, D1, dx2, omega, par, true);
		float4* q = reinterpret_cast<float4*>(p_out + (size_t)leaf1 * 512 + l * 8);
		q[0] = make_float4(o.q[0].x, o.q[0].y, o.q[1].x, o.q[1].y);
		q[1] = make_float4(o.q[2].x, o.q[2].y, o.q[3].x, o.q[3].y);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// subtractPressureGradient (reference Kernel.cu:765-829 / :694-762)
// ---------------------------------------------------------------------------------------------------------------

template <bool COLL>
__global__ __launch_bounds__(512) void k_subtract_gradient(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                           const float* __restrict__ uz, const float* __restrict__ p, float* ox, float* oy,
                                                           float* oz, const float* __restrict__ sdf, const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float gx = ((nbr_val<0, 1>(p, s_nbr, L.leaf, n) - nbr_val<0, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	const float gy = ((nbr_val<1, 1>(p, s_nbr, L.leaf, n) - nbr_val<1, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	const float gz = ((nbr_val<2, 1>(p, s_nbr, L.leaf, n) - nbr_val<2, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	f3 u = {ux[idx] - gx, uy[idx] - gy, uz[idx] - gz};
	if (COLL) {  // Kernel.cu:809-826
		const float sv = sdf[idx];
		if (sv < 0.0f) {
			u.x = u.y = u.z = 0.0f;
		} else if (sv < 0.1f) {
			const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
			const f3 nrm = sdf_normal(g, s_nbr, L.org, sdf, ci, cj, ck, inv_dx);
			u = no_slip_blend(u, nrm, 1.0f - (sv / 0.1f));
		}
	}
	ox[idx] = u.x;
	oy[idx] = u.y;
	oz[idx] = u.z;
}

// ---------------------------------------------------------------------------------------------------------------
// element-wise kernels
// ---------------------------------------------------------------------------------------------------------------

// combustion_oxygen (reference Kernel.cu:923-966)
__global__ __launch_bounds__(256) void k_combustion_oxygen(const float* __restrict__ fuelData, const float* __restrict__ wasteData,
                                                           const float* __restrict__ temperatureData, float* __restrict__ divergenceData,
                                                           const float* __restrict__ flameData, float* __restrict__ outFuel,
                                                           float* __restrict__ outWaste, float* __restrict__ outTemperature,
                                                           float* __restrict__ outFlame, const float temp_gain, const float expansion,
                                                           const uint64_t n) {
	for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
		float fuel = fuelData[idx];
		const float waste = wasteData[idx];
		const float temperature = temperatureData[idx];
		const float flame = flameData[idx];
		if (fuel < 0.001f) fuel = 0.0f;
		const float oxygen = 1.0f - fuel - waste;
		if (oxygen < 0.0f) {
			outFuel[idx] = fuel;
			outWaste[idx] = waste;
			outTemperature[idx] = temperature;
			outFlame[idx] = flame;
			continue;
		}
		const float burn = fminf(oxygen, fuel);
		outFuel[idx] = fuel - burn;
		outWaste[idx] = waste + burn * 2.0f;
		outTemperature[idx] = temperature + burn * temp_gain;
		divergenceData[idx] += burn * expansion;
		outFlame[idx] = fmaxf(flame, fminf(1.0f, burn * 10.0f));
	}
}

// temperature_buoyancy (reference Kernel.cu:831-847); x and z are vel + 0*dt == vel, so only uy is touched
__global__ __launch_bounds__(256) void k_temperature_buoyancy(const float* uy, const float* __restrict__ temp, float* out_uy, const float dt,
                                                              const float ambient, const float strength, const uint64_t n) {
	for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
		const float v = uy[idx];
		const float t = temp[idx];
		if (t <= ambient) {
			out_uy[idx] = v;
			continue;
		}
		const float tempDiff = t - ambient;
		out_uy[idx] = v + dt * fmaxf(0.0f, tempDiff * strength);
	}
}

__global__ __launch_bounds__(256) void k_aos_to_soa(const float* __restrict__ aos, float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ z, const uint64_t n) {
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		x[i] = aos[3 * i];
		y[i] = aos[3 * i + 1];
		z[i] = aos[3 * i + 2];
	}
}

__global__ __launch_bounds__(256) void k_soa_to_aos(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                    float* __restrict__ aos, const uint64_t n) {
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		aos[3 * i] = x[i];
		aos[3 * i + 1] = y[i];
		aos[3 * i + 2] = z[i];
	}
}

// whole-leaf gather/scatter for halo exchange: one float4 per thread, 128 threads per leaf
__global__ __launch_bounds__(128) void k_pack_leaves(const float* __restrict__ field, const int* __restrict__ ids, float* __restrict__ packed) {
	const int l = ids[blockIdx.x];
	reinterpret_cast<float4*>(packed + (size_t)blockIdx.x * 512)[threadIdx.x] = reinterpret_cast<const float4*>(field + (size_t)l * 512)[threadIdx.x];
}
__global__ __launch_bounds__(128) void k_unpack_leaves(const float* __restrict__ packed, const int* __restrict__ ids, float* __restrict__ field) {
	const int l = ids[blockIdx.x];
	reinterpret_cast<float4*>(field + (size_t)l * 512)[threadIdx.x] = reinterpret_cast<const float4*>(packed + (size_t)blockIdx.x * 512)[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------
// vorticityConfinement (reference Kernel.cu:970-1024 + Utils.cuh:226-243), out of place
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ f3 curl_at(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ ux,
                                      const float* __restrict__ uy, const float* __restrict__ uz, int i, int j, int k, float factor) {
	const int tpx = tap_index(g, s_nbr, org, i + 1, j, k), tmx = tap_index(g, s_nbr, org, i - 1, j, k);
	const int tpy = tap_index(g, s_nbr, org, i, j + 1, k), tmy = tap_index(g, s_nbr, org, i, j - 1, k);
	const int tpz = tap_index(g, s_nbr, org, i, j, k + 1), tmz = tap_index(g, s_nbr, org, i, j, k - 1);
	f3 w;
	w.x = ((ld0(uz, tpy) - ld0(uz, tmy)) - (ld0(uy, tpz) - ld0(uy, tmz))) * factor;
	w.y = ((ld0(ux, tpz) - ld0(ux, tmz)) - (ld0(uz, tpx) - ld0(uz, tmx))) * factor;
	w.z = ((ld0(uy, tpx) - ld0(uy, tmx)) - (ld0(ux, tpy) - ld0(ux, tmy))) * factor;
	return w;
}

__device__ __forceinline__ float curl_mag(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ ux,
                                          const float* __restrict__ uy, const float* __restrict__ uz, int i, int j, int k, float factor) {
	const f3 w = curl_at(g, s_nbr, org, ux, uy, uz, i, j, k, factor);
	return sqrtf(w.x * w.x + w.y * w.y + w.z * w.z);
}

__global__ __launch_bounds__(512) void k_vorticity(const GridDev g, const float* __restrict__ ux, const float* __restrict__ uy,
                                                   const float* __restrict__ uz, float* __restrict__ ox, float* __restrict__ oy,
                                                   float* __restrict__ oz, const float dt, const float inv_dx, const float scale, const int fs) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float factor = 0.5f * inv_dx;
	const f3 w = curl_at(g, s_nbr, L.org, ux, uy, uz, ci, cj, ck, factor);
	const float m_pX = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci + fs, cj, ck, factor), m_mX = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci - fs, cj, ck, factor);
	const float m_pY = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci, cj + fs, ck, factor), m_mY = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci, cj - fs, ck, factor);
	const float m_pZ = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci, cj, ck + fs, factor), m_mZ = curl_mag(g, s_nbr, L.org, ux, uy, uz, ci, cj, ck - fs, factor);
	const float grad_x = (m_pX - m_mX) * 0.5f * inv_dx;
	const float grad_y = (m_pY - m_mY) * 0.5f * inv_dx;
	const float grad_z = (m_pZ - m_mZ) * 0.5f * inv_dx;
	const float gradLen = sqrtf(grad_x * grad_x + grad_y * grad_y + grad_z * grad_z) + 1e-5f;
	const float Nx = grad_x / gradLen, Ny = grad_y / gradLen, Nz = grad_z / gradLen;
	ox[idx] = ux[idx] + dt * (scale * (Ny * w.z - Nz * w.y));
	oy[idx] = uy[idx] + dt * (scale * (Nz * w.x - Nx * w.z));
	oz[idx] = uz[idx] + dt * (scale * (Nx * w.y - Ny * w.x));
}

// enforceCollisionBoundaries (reference Kernel.cu:77-116)
__global__ __launch_bounds__(512) void k_enforce_collision(const GridDev g, float* ux, float* uy, float* uz, const float* __restrict__ sdf,
                                                           const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float sv = sdf[idx];
	if (sv < 0.0f) {
		ux[idx] = 0.0f;
		uy[idx] = 0.0f;
		uz[idx] = 0.0f;
		return;
	}
	const float margin = 0.1f;
	if (sv < margin) {
		const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
		const f3 nrm = sdf_normal(g, s_nbr, L.org, sdf, ci, cj, ck, inv_dx);
		const f3 v = {ux[idx], uy[idx], uz[idx]};
		const f3 r = no_slip_blend(v, nrm, 1.0f - (sv / margin));
		ux[idx] = r.x;
		uy[idx] = r.y;
		uz[idx] = r.z;
	}
}

}  // namespace hns

// ===============================================================================================================
// launchers (C ABI, include/hns.h "Kernel-level entry points")
// ===============================================================================================================

using namespace hns;

#define HNS_HIP(call)                                                                  \
	do {                                                                               \
		hipError_t e__ = (call);                                                       \
		if (e__ != hipSuccess) {                                                       \
			set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
			return HNS_ERR_HIP;                                                        \
		}                                                                              \
	} while (0)

static inline int check_grid(const hns_grid* g, const char* who) {
	if (!g) {
		set_error("%s: null grid", who);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	if (!g->on_device) {
		set_error("%s: grid has no device tables (host-only grid or no HIP device); there is no CPU fallback", who);
		return HNS_ERR_NO_DEVICE;
	}
	return HNS_OK;
}

static inline int launch_status(const char* who) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		set_error("%s: kernel launch failed: %s", who, hipGetErrorString(e));
		return HNS_ERR_HIP;
	}
	return HNS_OK;
}

static inline unsigned ew_blocks(uint64_t n) {
	uint64_t b = (n + 255) / 256;
	return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

#define NULLCHK(cond, who)                                \
	if (cond) {                                           \
		set_error("%s: null device pointer", who);        \
		return HNS_ERR_INVALID_ARGUMENT;                  \
	}

extern "C" {

int hns_dev_aos_to_soa(const float* aos3, float* x, float* y, float* z, uint64_t n, void* stream) {
	NULLCHK(!aos3 || !x || !y || !z, "hns_dev_aos_to_soa");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_aos_to_soa, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, aos3, x, y, z, n);
	return launch_status("hns_dev_aos_to_soa");
}

int hns_dev_soa_to_aos(const float* x, const float* y, const float* z, float* aos3, uint64_t n, void* stream) {
	NULLCHK(!aos3 || !x || !y || !z, "hns_dev_soa_to_aos");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_soa_to_aos, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, z, aos3, n);
	return launch_status("hns_dev_soa_to_aos");
}

int hns_dev_advect_vector(hns_grid* g, const float* ux, const float* uy, const float* uz, float* ox, float* oy, float* oz, const float* sdf,
                          int has_collision, float dt, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_vector")) return rc;
	NULLCHK(!ux || !uy || !uz || !ox || !oy || !oz, "hns_dev_advect_vector");
	if (g->n_active == 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;  // Kernel.cu:361
	const dim3 grid((unsigned)g->n_active), block(512);
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_advect_vector<true>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, ox, oy, oz, sdf, scaled_dt, inv_dx);
	else
		hipLaunchKernelGGL(k_advect_vector<false>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, ox, oy, oz, sdf, scaled_dt, inv_dx);
	return launch_status("hns_dev_advect_vector");
}

int hns_dev_advect_scalar(hns_grid* g, const float* ux, const float* uy, const float* uz, const float* in, float* out, const float* sdf,
                          int has_collision, float dt, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_scalar")) return rc;
	NULLCHK(!ux || !uy || !uz || !in || !out, "hns_dev_advect_scalar");
	if (g->n_active == 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;
	const dim3 grid((unsigned)g->n_active), block(512);
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_advect_scalar<true>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, in, out, sdf, scaled_dt);
	else
		hipLaunchKernelGGL(k_advect_scalar<false>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, in, out, sdf, scaled_dt);
	return launch_status("hns_dev_advect_scalar");
}

int hns_dev_advect_scalars(hns_grid* g, const float* ux, const float* uy, const float* uz, const float* const* in, float* const* out, int n,
                           const float* sdf, int has_collision, float dt, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_scalars")) return rc;
	NULLCHK(!ux || !uy || !uz || (n > 0 && (!in || !out)), "hns_dev_advect_scalars");
	if (g->n_active == 0 || n <= 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;
	const dim3 grid((unsigned)g->n_active), block(512);
	// the backtrace does not depend on the fields, so splitting S fields over several launches changes nothing numerically
	for (int base = 0; base < n; base += HNS_MAX_SCALARS) {
		ScalarPtrs P;
		P.n = n - base < HNS_MAX_SCALARS ? n - base : HNS_MAX_SCALARS;
		for (int s = 0; s < HNS_MAX_SCALARS; ++s) {
			P.in[s] = s < P.n ? in[base + s] : nullptr;
			P.out[s] = s < P.n ? out[base + s] : nullptr;
			if (s < P.n && (!P.in[s] || !P.out[s])) {
				set_error("hns_dev_advect_scalars: null device pointer for field %d", base + s);
				return HNS_ERR_INVALID_ARGUMENT;
			}
		}
		if (has_collision && sdf)
			hipLaunchKernelGGL(k_advect_scalars<true>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, P, sdf, scaled_dt);
		else
			hipLaunchKernelGGL(k_advect_scalars<false>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, P, sdf, scaled_dt);
	}
	return launch_status("hns_dev_advect_scalars");
}

int hns_dev_divergence(hns_grid* g, const float* ux, const float* uy, const float* uz, float* div, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_divergence")) return rc;
	NULLCHK(!ux || !uy || !uz || !div, "hns_dev_divergence");
	if (g->n_active == 0) return HNS_OK;
	hipLaunchKernelGGL(k_divergence, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), ux, uy, uz, div, inv_dx);
	return launch_status("hns_dev_divergence");
}

int hns_dev_rbgs_color(hns_grid* g, const float* div, float* p, float dx, float omega, int color, void* stream) {
	if (int rc = check_grid(g, "hns_dev_rbgs_color")) return rc;
	NULLCHK(!div || !p, "hns_dev_rbgs_color");
	if (color != 0 && color != 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_color: color must be 0 (red) or 1 (black)");
	if (g->n_active == 0) return HNS_OK;
	hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, (hipStream_t)stream, g->dev(), div, p, dx * dx, omega, color);
	return launch_status("hns_dev_rbgs_color");
}

int hns_dev_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                         void* stream) {
	if (int rc = check_grid(g, "hns_dev_rbgs_iterate")) return rc;
	NULLCHK(!div || !p_a || !p_b, "hns_dev_rbgs_iterate");
	if (p_a == p_b) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: p_a and p_b must be distinct buffers");
	if (iterations < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: negative iteration count");
	if (result_in_b) *result_in_b = iterations & 1;
	if (g->n_active == 0) return HNS_OK;
	const float dx2 = dx * dx;  // Kernel.cu:608
	const GridDev gd = g->dev();
	float* src = p_a;
	float* dst = p_b;
	static const bool use_lds_kernel = getenv("HNS_RBGS") && strcmp(getenv("HNS_RBGS"), "block") == 0;  // A/B switch: 256-thread LDS-tile form
	static const bool use_wave_kernel = getenv("HNS_RBGS") && strcmp(getenv("HNS_RBGS"), "wave") == 0;   // A/B switch: one leaf per wave everywhere
	for (int it = 0; it < iterations; ++it) {
		if (use_lds_kernel)
			hipLaunchKernelGGL(k_rbgs_fused, dim3((unsigned)g->n_active), dim3(256), 0, (hipStream_t)stream, gd, div, (const float*)src, dst, dx2, omega);
		else if (use_wave_kernel || !g->d_pairs)
			hipLaunchKernelGGL(k_rbgs_wave, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, div, (const float*)src, dst, dx2, omega);
		else {
			// paired leaves and the unpaired remainder are disjoint and both read src / write dst: two independent launches
			if (g->n_pairs)
				hipLaunchKernelGGL(k_rbgs_pair, dim3((unsigned)g->n_pairs), dim3(64), 0, (hipStream_t)stream, (const int*)g->d_pairs, div, (const float*)src,
				                   dst, dx2, omega);
			if (g->n_singles) {
				GridDev gs = gd;
				gs.blk = (const int*)g->d_singles;
				hipLaunchKernelGGL(k_rbgs_wave, dim3((unsigned)g->n_singles), dim3(64), 0, (hipStream_t)stream, gs, div, (const float*)src, dst, dx2, omega);
			}
		}
		float* tmp = src;
		src = dst;
		dst = tmp;
	}
	return launch_status("hns_dev_rbgs_iterate");
}

int hns_dev_time_rbgs(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int reps, float* ms_per_launch,
                      void* stream) {
	if (int rc = check_grid(g, "hns_dev_time_rbgs")) return rc;
	if (!ms_per_launch || iterations <= 0 || reps <= 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_time_rbgs: bad arguments");
	hipEvent_t e0, e1;
	HNS_HIP(hipEventCreate(&e0));
	HNS_HIP(hipEventCreate(&e1));
	double total = 0.0;
	int rc = HNS_OK;
	for (int r = 0; r < reps && rc == HNS_OK; ++r) {
		HNS_HIP(hipEventRecord(e0, (hipStream_t)stream));
		rc = hns_dev_rbgs_iterate(g, div, p_a, p_b, dx, omega, iterations, nullptr, stream);
		HNS_HIP(hipEventRecord(e1, (hipStream_t)stream));
		HNS_HIP(hipEventSynchronize(e1));
		float ms = 0.0f;
		HNS_HIP(hipEventElapsedTime(&ms, e0, e1));
		total += ms;
	}
	hipEventDestroy(e0);
	hipEventDestroy(e1);
	*ms_per_launch = (float)(total / ((double)reps * iterations));
	return rc;
}

int hns_dev_subtract_pressure_gradient(hns_grid* g, const float* ux, const float* uy, const float* uz, const float* p, float* ox, float* oy,
                                       float* oz, const float* sdf, int has_collision, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_subtract_pressure_gradient")) return rc;
	NULLCHK(!ux || !uy || !uz || !p || !ox || !oy || !oz, "hns_dev_subtract_pressure_gradient");
	if (g->n_active == 0) return HNS_OK;
	const dim3 grid((unsigned)g->n_active), block(512);
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_subtract_gradient<true>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, p, ox, oy, oz, sdf, inv_dx);
	else
		hipLaunchKernelGGL(k_subtract_gradient<false>, grid, block, 0, (hipStream_t)stream, g->dev(), ux, uy, uz, p, ox, oy, oz, sdf, inv_dx);
	return launch_status("hns_dev_subtract_pressure_gradient");
}

int hns_dev_combustion_oxygen(const float* fuel, const float* waste, const float* temperature, float* divergence, const float* flame,
                              float* out_fuel, float* out_waste, float* out_temperature, float* out_flame, float temp_gain, float expansion,
                              uint64_t n, void* stream) {
	NULLCHK(!fuel || !waste || !temperature || !divergence || !flame || !out_fuel || !out_waste || !out_temperature || !out_flame,
	        "hns_dev_combustion_oxygen");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_combustion_oxygen, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, fuel, waste, temperature, divergence, flame,
	                   out_fuel, out_waste, out_temperature, out_flame, temp_gain, expansion, n);
	return launch_status("hns_dev_combustion_oxygen");
}

int hns_dev_temperature_buoyancy(const float* uy, const float* temperature, float* out_uy, float dt, float ambient, float strength, uint64_t n,
                                 void* stream) {
	NULLCHK(!uy || !temperature || !out_uy, "hns_dev_temperature_buoyancy");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_temperature_buoyancy, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, uy, temperature, out_uy, dt, ambient, strength,
	                   n);
	return launch_status("hns_dev_temperature_buoyancy");
}

int hns_dev_vorticity_confinement(hns_grid* g, const float* ux, const float* uy, const float* uz, float* ox, float* oy, float* oz, float dt,
                                  float inv_dx, float confinement_scale, float factor_scale, void* stream) {
	if (int rc = check_grid(g, "hns_dev_vorticity_confinement")) return rc;
	NULLCHK(!ux || !uy || !uz || !ox || !oy || !oz, "hns_dev_vorticity_confinement");
	if (ux == ox || uy == oy || uz == oz) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_vorticity_confinement: output must not alias input");
	if (g->n_active == 0) return HNS_OK;
	const int fs = (int)factor_scale;  // nanovdb::Coord(factorScale,0,0) truncates (Kernel.cu:998)
	hipLaunchKernelGGL(k_vorticity, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), ux, uy, uz, ox, oy, oz, dt, inv_dx,
	                   confinement_scale, fs);
	return launch_status("hns_dev_vorticity_confinement");
}

int hns_dev_enforce_collision_boundaries(hns_grid* g, float* ux, float* uy, float* uz, const float* sdf, float voxel_size, void* stream) {
	if (int rc = check_grid(g, "hns_dev_enforce_collision_boundaries")) return rc;
	NULLCHK(!ux || !uy || !uz, "hns_dev_enforce_collision_boundaries");
	if (!sdf || g->n_active == 0) return HNS_OK;  // Kernel.cu:83
	hipLaunchKernelGGL(k_enforce_collision, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), ux, uy, uz, sdf,
	                   1.0f / voxel_size);
	return launch_status("hns_dev_enforce_collision_boundaries");
}

int hns_dev_pack_leaves(const float* field, const int32_t* leaf_ids, uint64_t n, float* packed, void* stream) {
	NULLCHK((!field || !leaf_ids || !packed) && n, "hns_dev_pack_leaves");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_pack_leaves, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, field, leaf_ids, packed);
	return launch_status("hns_dev_pack_leaves");
}

int hns_dev_unpack_leaves(const float* packed, const int32_t* leaf_ids, uint64_t n, float* field, void* stream) {
	NULLCHK((!field || !leaf_ids || !packed) && n, "hns_dev_unpack_leaves");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_unpack_leaves, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, packed, leaf_ids, field);
	return launch_status("hns_dev_unpack_leaves");
}

}  // extern "C"
