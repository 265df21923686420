// hns_advect.hip -- BFECC semi-Lagrangian advection kernels (reference src/Cuda/Kernel.cu:118-453).
//
// One 8^3 leaf per 512-thread workgroup, one voxel per thread (coordinate = leaf origin + thread id: no 12 B/voxel
// coordinate stream). The reference walks the NanoVDB tree for each of its 23-31 taps per voxel; here a tap is
//     slot = 9*dx + 3*dy + dz (leaf offset of the tap relative to the workgroup's leaf, each in {0,1,2})
//     index = s_base[slot] + local offset          (s_base = the 27 neighbour leaves' base indices, staged in LDS)
// computed once per axis for the two planes of a trilinear stencil and combined for its 8 corners, branch-free. Taps
// farther than one leaf away (|u| dt/dx > 8) take a generic path -- the neighbour's own neighbour table up to two leaves away, the origin hash beyond; wave-divergent but rare.
// The arithmetic (Floor, lerp order z->y->x, fused Vec3f lerps, unfused float lerps, weight-product form of
// advect_scalars, clamp set and order) is the reference's, so results stay bit-identical to the oracle.
#include <cstdlib>
#include <cstring>

#include "hns_device.hpp"

namespace hns {

struct Taps {
	int t[8];  // flat voxel index of corner (di,dj,dk) at t[di*4+dj*2+dk], -1 = outside the domain
	float fx, fy, fz;
};

// ---- 32-bit addressed field access ---------------------------------------------------------------------------------
// The advection kernels issue ~590 vector ALU instructions per voxel against 23 loads and are bound by VALU issue, not by
// memory; a large share of those instructions only guards and addresses the taps: a 64-bit multiply-add per address,
// an index clamp and three "value or 0" selects per out-of-domain-capable tap (IndexSampler<T,0>, Stencils.hpp:83,88).
// While a field is below 4 GiB all of that is a property of the load instead: a buffer descriptor over the whole field
// takes a 32-bit byte offset, and the hardware returns 0 for offsets past the end. Taps are therefore carried as the
// byte offset of the voxel in a float field (voxel * 4; a Vec3f tap is at three times that), and a tap outside the
// domain is any offset >= kOutside: absent neighbour leaves get kOutside as their base, so no select is needed at all.
typedef float v3f __attribute__((ext_vector_type(3)));
typedef int v4i __attribute__((ext_vector_type(4)));
// raw buffer loads through the LLVM intrinsics (this toolchain's __builtin_amdgcn_raw_buffer_load_b96 returns one dword)
__device__ v3f hns_buffer_load_v3f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
__device__ float hns_buffer_load_f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
typedef float v2f32 __attribute__((ext_vector_type(2)));
__device__ v2f32 hns_buffer_load_v2f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
typedef float v4f32 __attribute__((ext_vector_type(4)));
__device__ v4f32 hns_buffer_load_v4f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

constexpr unsigned kOutside = 0xFFFFE000u;        // float-field byte offsets at or above this read as 0 (and 3x it still lies past a Vec3f field)
constexpr uint64_t kNarrowBytes = 0xFFFF0000ull;  // largest Vec3f field the 32-bit path accepts (and largest 16-byte-per-voxel field of the q4 path)

__device__ __forceinline__ v4i field_rsrc(const float* p, unsigned bytes) {
	const unsigned long long a = (unsigned long long)p;
	v4i r;
	r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0: raw buffer
	r.z = __builtin_amdgcn_readfirstlane((int)bytes);                            // num_records in bytes
	r.w = 0x00020000;                                                            // 32-bit float data format
	return r;
}
__device__ __forceinline__ f3 ldv(const v4i& r, unsigned o4) {  // Vec3f of the voxel at float-offset o4; 0 outside
	const v3f v = hns_buffer_load_v3f32(r, (int)(o4 + (o4 << 1)), 0, 0);
	return f3{v.x, v.y, v.z};
}
__device__ __forceinline__ float lds1(const v4i& r, unsigned o4) { return hns_buffer_load_f32(r, (int)o4, 0, 0); }

// The two z-corners of a trilinear column, float field: when they are neighbours in memory (same leaf, k & 7 != 7) one
// 8-byte load fetches both; otherwise the second comes from its own load, which only the few lanes whose column
// straddles a leaf face execute. A load instruction costs the L1 the same 16 cycles per wave whatever its width, and
// the scalar gathers are what bounds advect_scalars, so halving them is the point. Reading 4 bytes past `lo` is safe:
// the descriptor's bounds check covers the end of the field.
__device__ __forceinline__ void ld_zpair(const v4i& r, unsigned lo, unsigned hi, float& a, float& b) {
	const v2f32 v = hns_buffer_load_v2f32(r, (int)lo, 0, 0);
	a = v.x;
	b = v.y;
	if (hi != lo + 4u) b = lds1(r, hi);
}

// stage nbr27 (for the generic path), the neighbours' base indices leaf*512 (-1 = absent) and, for the 32-bit path, their
// base byte offsets in a float field (kOutside = absent)
// s_b4p (32-bit kernels): the same byte bases again in a table padded to 4 x 4 x 4, entry (ax*4 + ay)*4 + az for the neighbour leaf
// (ax, ay, az) in 0..2 -- its byte index ax<<6 | ay<<4 | az<<2 is shifts and ORs of the tap's coordinates, where 9*ax + 3*ay + az cost
// four quarter-rate multiplies per trilinear sample (make_taps_b).
constexpr int kPadTab = 48;
__device__ __forceinline__ LeafCtx stage_leaf_base(const GridDev& g, int* s_nbr, int* s_base, int block, unsigned* s_b4 = nullptr, unsigned* s_b4p = nullptr, int leaf = -1) {
	LeafCtx c;
	c.leaf = leaf >= 0 ? leaf : launch_leaf(g, (unsigned)block);
	c.org = g.origins[c.leaf];
	if (threadIdx.x < 27) {
		const int nb = g.nbr27[c.leaf * 27 + threadIdx.x];
		s_nbr[threadIdx.x] = nb;
		s_base[threadIdx.x] = nb < 0 ? -1 : nb * 512;
		if (s_b4) s_b4[threadIdx.x] = nb < 0 ? kOutside : (unsigned)nb * 2048u;
		if (s_b4p) {
			const int t = threadIdx.x, ax = t / 9, ay = (t - 9 * ax) / 3, az = t - 9 * ax - 3 * ay;
			s_b4p[(ax * 4 + ay) * 4 + az] = nb < 0 ? kOutside : (unsigned)nb * 2048u;
		}
	}
	__syncthreads();
	return c;
}

struct TapsB {
	unsigned o[8];  // float-field byte offset of corner (di,dj,dk) at o[di*4+dj*2+dk]; >= kOutside: outside the domain
	float fx, fy, fz;
};

// Floor (Stencils.hpp:25-43) + the eight corner indices of TrilinearSampler::stencil (Stencils.hpp:104-114)
__device__ __forceinline__ Taps make_taps(const GridDev& g, const int* s_nbr, const int* s_base, const int4 org, float x, float y, float z) {
	Taps T;
	const int i = __float2int_rd(x), j = __float2int_rd(y), k = __float2int_rd(z);
	T.fx = x - (float)i;
	T.fy = y - (float)j;
	T.fz = z - (float)k;
	// leaf offsets (+1) of the two planes per axis; near <=> all in {0,1,2}
	const int ax0 = (i >> 3) - (org.x >> 3) + 1, ax1 = ((i + 1) >> 3) - (org.x >> 3) + 1;
	const int ay0 = (j >> 3) - (org.y >> 3) + 1, ay1 = ((j + 1) >> 3) - (org.y >> 3) + 1;
	const int az0 = (k >> 3) - (org.z >> 3) + 1, az1 = ((k + 1) >> 3) - (org.z >> 3) + 1;
	const bool near = ((unsigned)ax0 <= 2u) & ((unsigned)ax1 <= 2u) & ((unsigned)ay0 <= 2u) & ((unsigned)ay1 <= 2u) & ((unsigned)az0 <= 2u) &
	                  ((unsigned)az1 <= 2u);
	if (near) {
		const int sx[2] = {ax0 * 9, ax1 * 9}, sy[2] = {ay0 * 3, ay1 * 3}, sz[2] = {az0, az1};
		const int lx[2] = {(i & 7) << 6, ((i + 1) & 7) << 6}, ly[2] = {(j & 7) << 3, ((j + 1) & 7) << 3}, lz[2] = {k & 7, (k + 1) & 7};
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
			const int b = s_base[sx[di] + sy[dj] + sz[dk]];
			T.t[c] = b < 0 ? -1 : b + (lx[di] | ly[dj] | lz[dk]);
		}
	} else {
		int any = 0;
		far_cell_taps(g, s_nbr, org, i, j, k, T.t);
#pragma unroll
		for (int c = 0; c < 8; ++c) any |= T.t[c];
		// a multi-GPU rank: a tap beyond the 27-leaf neighbourhood whose leaf is not HERE may exist on another rank (hns_dist reports
		// it); one that resolves to a local leaf -- owned or ghost, both hold current values -- is answered as the single domain answers it
		if (any < 0 && g.far_flag) *g.far_flag = 1;
	}
	return T;
}

// the same for the 32-bit path: byte offsets, and an absent leaf needs no test (its base is kOutside).
// Round 4: the kernels that call this keep the VALU 86 % busy (4 cycles per wave instruction; SQ_ACTIVE_INST_VALU) next to a texture
// addresser at 84 %, and a good half of their instructions address taps. So: d = cell - (corner of the 3 x 3 x 3 leaves around the
// workgroup's leaf) per axis, "near" <=> every d in [0, 22] (cell and cell + 1 inside the 24 voxels: ONE max3 and ONE compare instead of
// six range tests), neighbour slot and voxel-in-leaf are bit fields of d, the table is padded so that its index is ORs (s_b4p), and
// the eight offsets are add3's of three precombined terms.
__device__ __forceinline__ TapsB make_taps_b(const GridDev& g, const int* s_nbr, const unsigned* s_b4p, const int4 org, float x, float y, float z) {
	TapsB T;
	const int i = __float2int_rd(x), j = __float2int_rd(y), k = __float2int_rd(z);
	T.fx = x - (float)i;
	T.fy = y - (float)j;
	T.fz = z - (float)k;
	const unsigned dx = (unsigned)(i - (org.x - 8)), dy = (unsigned)(j - (org.y - 8)), dz = (unsigned)(k - (org.z - 8));
	if (max(dx, max(dy, dz)) < 23u) {
		const unsigned ex = dx + 1u, ey = dy + 1u, ez = dz + 1u;
		// table byte index (d >> 3) << {6, 4, 2} and voxel-in-leaf byte offset (d & 7) << {8, 5, 2}, for the cell (0) and cell + 1 (1)
		const unsigned X[2] = {(dx & 24u) << 3, (ex & 24u) << 3}, Y[2] = {(dy & 24u) << 1, (ey & 24u) << 1}, Z[2] = {(dz & 24u) >> 1, (ez & 24u) >> 1};
		const unsigned lx[2] = {(dx & 7u) << 8, (ex & 7u) << 8}, ly[2] = {(dy & 7u) << 5, (ey & 7u) << 5}, lz[2] = {(dz & 7u) << 2, (ez & 7u) << 2};
		const unsigned XY[4] = {X[0] | Y[0], X[0] | Y[1], X[1] | Y[0], X[1] | Y[1]};
		const unsigned lxy[4] = {lx[0] | ly[0], lx[0] | ly[1], lx[1] | ly[0], lx[1] | ly[1]};
		const char* tab = reinterpret_cast<const char*>(s_b4p);
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const int dij = c >> 1, dk = c & 1;
			T.o[c] = *reinterpret_cast<const unsigned*>(tab + (XY[dij] | Z[dk])) + lxy[dij] + lz[dk];
		}
	} else {
		int any = 0, ft[8];
		far_cell_taps(g, s_nbr, org, i, j, k, ft);  // (round 6: through the neighbour tables up to two leaves away, the hash beyond)
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			any |= ft[c];
			T.o[c] = ft[c] < 0 ? kOutside : (unsigned)ft[c] << 2;
		}
		if (any < 0 && g.far_flag) *g.far_flag = 1;  // a multi-GPU rank: a far tap whose leaf is not here may exist on another rank (see make_taps)
	}
	return T;
}

// value or 0 outside the domain (IndexSampler<T,0>, Stencils.hpp:81-89), without a branch
__device__ __forceinline__ float ldz(const float* __restrict__ f, int idx) {
	const float v = f[idx < 0 ? 0 : idx];
	return idx < 0 ? 0.0f : v;
}

// IndexSampler<float,1>: unfused a + w*(b-a), z then y then x (Stencils.hpp:140-152)
__device__ __forceinline__ float tri_f_t(const float* __restrict__ f, const Taps& T) {
	const float z0 = lerp_f(ldz(f, T.t[0]), ldz(f, T.t[1]), T.fz);
	const float z1 = lerp_f(ldz(f, T.t[2]), ldz(f, T.t[3]), T.fz);
	const float z2 = lerp_f(ldz(f, T.t[4]), ldz(f, T.t[5]), T.fz);
	const float z3 = lerp_f(ldz(f, T.t[6]), ldz(f, T.t[7]), T.fz);
	const float y0 = lerp_f(z0, z1, T.fy);
	const float y1 = lerp_f(z2, z3, T.fy);
	return lerp_f(y0, y1, T.fx);
}

// IndexSampler<Vec3f,1> on the device branch: per component fmaf(w, b-a, a) (Stencils.hpp:131-135); eight 12-byte taps
template <class TapsT>
__device__ __forceinline__ float tri_c8(float c0, float c1, float c2, float c3, float c4, float c5, float c6, float c7, const TapsT& T) {
	const float z0 = lerp_c(c0, c1, T.fz);
	const float z1 = lerp_c(c2, c3, T.fz);
	const float z2 = lerp_c(c4, c5, T.fz);
	const float z3 = lerp_c(c6, c7, T.fz);
	const float y0 = lerp_c(z0, z1, T.fy);
	const float y1 = lerp_c(z2, z3, T.fy);
	return lerp_c(y0, y1, T.fx);
}

__device__ __forceinline__ f3 tri_v_t(const float* __restrict__ u, const Taps& T) {
	f3 c[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) c[q] = ld3z(u, T.t[q]);
	f3 r;
	r.x = tri_c8(c[0].x, c[1].x, c[2].x, c[3].x, c[4].x, c[5].x, c[6].x, c[7].x, T);
	r.y = tri_c8(c[0].y, c[1].y, c[2].y, c[3].y, c[4].y, c[5].y, c[6].y, c[7].y, T);
	r.z = tri_c8(c[0].z, c[1].z, c[2].z, c[3].z, c[4].z, c[5].z, c[6].z, c[7].z, T);
	return r;
}

// One Vec3f lerp of the device branch, fmaf(w, b - a, a) per component (Stencils.hpp:131-135), written on the (x, y) pair and on z: a
// 12-byte load lands x and y in an aligned register pair, so the pair goes through v_pk_add_f32 / v_pk_fma_f32 as it is (the same IEEE
// operations per component). Left to the SLP vectoriser the seven lerps of a sample paired components of DIFFERENT taps and paid 16
// register moves per sample for it.
struct V3 {
	v2f32 xy;
	float z;
};
__device__ __forceinline__ V3 lerp_v3(const V3& a, const V3& b, float w) {
	V3 r;
	r.xy = __builtin_elementwise_fma(v2f32{w, w}, b.xy - a.xy, a.xy);
	r.z = __fmaf_rn(w, b.z - a.z, a.z);
	return r;
}
__device__ __forceinline__ f3 tri_v_b(const v4i& ru, const TapsB& T) {
	V3 c[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const v3f v = hns_buffer_load_v3f32(ru, (int)(T.o[q] + (T.o[q] << 1)), 0, 0);
		c[q].xy = v2f32{v.x, v.y};
		c[q].z = v.z;
	}
	const V3 z0 = lerp_v3(c[0], c[1], T.fz), z1 = lerp_v3(c[2], c[3], T.fz), z2 = lerp_v3(c[4], c[5], T.fz), z3 = lerp_v3(c[6], c[7], T.fz);
	const V3 y0 = lerp_v3(z0, z1, T.fy), y1 = lerp_v3(z2, z3, T.fy);
	const V3 r = lerp_v3(y0, y1, T.fx);
	return f3{r.xy.x, r.xy.y, r.z};
}

__device__ __forceinline__ float tri_f_b(const v4i& rf, const TapsB& T) {
	const float z0 = lerp_f(lds1(rf, T.o[0]), lds1(rf, T.o[1]), T.fz);
	const float z1 = lerp_f(lds1(rf, T.o[2]), lds1(rf, T.o[3]), T.fz);
	const float z2 = lerp_f(lds1(rf, T.o[4]), lds1(rf, T.o[5]), T.fz);
	const float z3 = lerp_f(lds1(rf, T.o[6]), lds1(rf, T.o[7]), T.fz);
	const float y0 = lerp_f(z0, z1, T.fy);
	const float y1 = lerp_f(z2, z3, T.fy);
	return lerp_f(y0, y1, T.fx);
}

// float-field byte offset of the face neighbour of own voxel n along AXIS/DIR (>= kOutside where that leaf is absent)
template <int AXIS, int DIR>
__device__ __forceinline__ unsigned nbr_off(const unsigned* s_b4, unsigned own, int n) {
	constexpr int shift = AXIS == 0 ? 6 : (AXIS == 1 ? 3 : 0);
	constexpr int stride = 1 << shift;
	constexpr int dslot = AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1);
	const int c = (n >> shift) & 7;
	const bool inside = DIR > 0 ? c != 7 : c != 0;
	return inside ? own + (unsigned)(DIR * stride * 4) : s_b4[13 + DIR * dslot] + (unsigned)((n - DIR * 7 * stride) << 2);
}
__device__ __forceinline__ void nbr6_b(const unsigned* s_b4, unsigned own, int n, unsigned (&o)[6]) {
	o[0] = nbr_off<0, -1>(s_b4, own, n);
	o[1] = nbr_off<0, 1>(s_b4, own, n);
	o[2] = nbr_off<1, -1>(s_b4, own, n);
	o[3] = nbr_off<1, 1>(s_b4, own, n);
	o[4] = nbr_off<2, -1>(s_b4, own, n);
	o[5] = nbr_off<2, 1>(s_b4, own, n);
}

// flat index of the face neighbour of voxel n of the workgroup's leaf along AXIS in direction DIR, -1 = outside
template <int AXIS, int DIR>
__device__ __forceinline__ int nbr_idx(const int* s_base, int leaf, int n) {
	constexpr int shift = AXIS == 0 ? 6 : (AXIS == 1 ? 3 : 0);
	constexpr int stride = 1 << shift;
	constexpr int dslot = AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1);
	const int c = (n >> shift) & 7;
	const bool inside = DIR > 0 ? c != 7 : c != 0;
	const int b = s_base[13 + DIR * dslot];
	const int in_leaf = leaf * 512 + n + DIR * stride;
	const int out_leaf = b < 0 ? -1 : b + n - DIR * 7 * stride;
	return inside ? in_leaf : out_leaf;
}

// the six face neighbours in the reference's order -x,+x,-y,+y,-z,+z (Kernel.cu:219,334-342,410-421)
__device__ __forceinline__ void nbr6(const int* s_base, int leaf, int n, int (&t)[6]) {
	t[0] = nbr_idx<0, -1>(s_base, leaf, n);
	t[1] = nbr_idx<0, 1>(s_base, leaf, n);
	t[2] = nbr_idx<1, -1>(s_base, leaf, n);
	t[3] = nbr_idx<1, 1>(s_base, leaf, n);
	t[4] = nbr_idx<2, -1>(s_base, leaf, n);
	t[5] = nbr_idx<2, 1>(s_base, leaf, n);
}

// ---------------------------------------------------------------------------------------------------------------
// advect_vector (reference Kernel.cu:354-453): BFECC self-advection of the velocity, clamped
// ---------------------------------------------------------------------------------------------------------------

// float-field byte offset of LDS-tile halo entry h (hns_device.hpp); >= kOutside where that neighbour leaf is absent
__device__ __forceinline__ unsigned halo_off(const unsigned* s_b4, int h) {
	int slot, local;
	halo_entry(h, slot, local);
	return s_b4[slot] + ((unsigned)local << 2);
}

// ---- the leaf and ONE voxel around it as a 10 x 10 x 10 box in LDS (round 6) ------------------------------------------------------------------
// BFECC's second sample is taken at back + u(back) * s, which is the voxel's own position up to s * (u(back) - u(own)): a fraction of a voxel wherever the
// velocity is smooth. Its eight taps then lie among the voxel and its 26 neighbours -- values the workgroup holds anyway once the tile the clamp needs
// (own leaf + six face layers, 896 voxels) is completed by the twelve edges and eight corners (104 more). Where the sample lands inside the box the kernel reads
// the second sample's taps from LDS: 8 of the kernel's 17 gathers (26 L1 tag lookups per wave each, the unit that bounds it: profiles/floors.py) are gone
// for 12 % more staging. Outside (a steep velocity gradient) the taps are gathered as before. Same values, same arithmetic: bit-identical.
// Box cell of voxel (x, y, z) relative to the leaf origin, each in [-1, 8]: ((x + 1) * 10 + y + 1) * 10 + z + 1; three component planes of kBox floats.
constexpr int kBox = 1000, kBoxShell = kBox - 512;
// k_advect_vector_n pads its rows from 10 to 24 floats (kVY; a plane = kVP floats): a wave's taps are an 8 (y) x 8 (z) window of one x-slice, and with rows 24 apart the four rows of
// a 32-lane group fall on banks c, c + 24, c + 16, c + 8 (+ 0..7): every bank once. With rows 10 apart six lanes of every group collide (46 % of the kernel's LDS cycles were conflicts).
constexpr int kVY = 24, kVX = 10 * kVY, kVP = 10 * kVX;
// shell cell h in [0, 488): the two full x-slabs (2 x 100), the y = -1 / 8 rows of the inner x (2 x 80), the z = -1 / 8 ends of the inner rows (2 x 64):
// slot of its leaf in the 27-table, voxel inside that leaf, box cell
template <int XS = 100, int YS = 10>  // strides of the box the cell number is for
__device__ __forceinline__ void box_shell_entry(int h, int& slot, int& local, int& cell) {
	int bx, by, bz;
	if (h < 200) {
		const int side = h >= 100, r = h - 100 * side;
		bx = 9 * side, by = r / 10, bz = r - 10 * by;
	} else if (h < 360) {
		const int q = h - 200, side = q >= 80, r = q - 80 * side;
		bx = 1 + r / 10, by = 9 * side, bz = r - 10 * (bx - 1);
	} else {
		const int q = h - 360, side = q >> 6, r = q & 63;
		bx = 1 + (r >> 3), by = 1 + (r & 7), bz = 9 * side;
	}
	slot = ((bx + 7) >> 3) * 9 + ((by + 7) >> 3) * 3 + ((bz + 7) >> 3);
	local = (((bx + 7) & 7) << 6) | (((by + 7) & 7) << 3) | ((bz + 7) & 7);
	cell = bx * XS + by * YS + bz;
}
__device__ __forceinline__ V3 box_v3(const float* s_box, int a) {
	V3 r;
	r.xy = v2f32{s_box[a], s_box[a + kVP]};
	r.z = s_box[a + 2 * kVP];
	return r;
}
// TrilinearSampler over the box: a = cell of the lower corner
__device__ __forceinline__ f3 tri_v_box(const float* s_box, int a, float fx, float fy, float fz) {
	V3 c[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) c[q] = box_v3(s_box, a + (q >> 2) * kVX + ((q >> 1) & 1) * kVY + (q & 1));
	const V3 z0 = lerp_v3(c[0], c[1], fz), z1 = lerp_v3(c[2], c[3], fz), z2 = lerp_v3(c[4], c[5], fz), z3 = lerp_v3(c[6], c[7], fz);
	const V3 y0 = lerp_v3(z0, z1, fy), y1 = lerp_v3(z2, z3, fy);
	const V3 r = lerp_v3(y0, y1, fx);
	return f3{r.xy.x, r.xy.y, r.z};
}

// 32-bit addressed form (no collision field): same loads and arithmetic as the generic kernel below
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_advect_vector_n(const GridDev g, const float* __restrict__ u, float* __restrict__ out, const float scaled_dt) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	__shared__ unsigned s_b4[27];
	__shared__ unsigned s_b4p[kPadTab];
	// the voxel's own velocity needs the leaf number only: its load is issued before the neighbour table is fetched and staged (one memory
	// round trip less in front of the first gathers; the kernel is bound by the length of that chain)
	const int n = threadIdx.x;
	const int leaf = launch_leaf(g, blockIdx.x);
	const int idx = leaf * 512 + n;
	const v4i ru = field_rsrc(u, (unsigned)g.n_leaves * 6144u);
	const unsigned own = (unsigned)idx << 2;
	const f3 vo = ldv(ru, own);
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x, s_b4, s_b4p, leaf);
	const float px = (float)(L.org.x + (n >> 6)), py = (float)(L.org.y + ((n >> 3) & 7)), pz = (float)(L.org.z + (n & 7));

	__shared__ float s_box[3 * kVP];
	const int ob = ((n >> 6) + 1) * kVX + (((n >> 3) & 7) + 1) * kVY + (n & 7) + 1;
	s_box[ob] = vo.x, s_box[ob + kVP] = vo.y, s_box[ob + 2 * kVP] = vo.z;
	if (n < kBoxShell) {
		int slot, local, cell;
		box_shell_entry<kVX, kVY>(n, slot, local, cell);
		const f3 h = ldv(ru, s_b4[slot] + ((unsigned)local << 2));
		s_box[cell] = h.x, s_box[cell + kVP] = h.y, s_box[cell + 2 * kVP] = h.z;
	}
	float sx = px - scaled_dt * vo.x, sy = py - scaled_dt * vo.y, sz = pz - scaled_dt * vo.z;  // backPos (Kernel.cu:374)
	f3 vf = {0.0f, 0.0f, 0.0f}, vb = {0.0f, 0.0f, 0.0f};
	__syncthreads();  // box complete
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		const int i = __float2int_rd(sx), j = __float2int_rd(sy), k = __float2int_rd(sz);
		const unsigned rx = (unsigned)(i - (L.org.x - 1)), ry = (unsigned)(j - (L.org.y - 1)), rz = (unsigned)(k - (L.org.z - 1));  // cell and cell + 1 inside the box <=> each in [0, 8]
		f3 v;
		// (either sample: the FIRST one too lands in the box where the flow moves less than a voxel per step -- then the wave gathers nothing at all. Per WAVE here: with a lane
		// outside, all of them gather -- measured 2.4 % faster through the plume's transient than a per-lane split; advect_scalars splits per lane)
		if (__all(max(rx, max(ry, rz)) <= 8u)) {
			v = tri_v_box(s_box, (int)(rx * (unsigned)kVX + ry * (unsigned)kVY + rz), sx - (float)i, sy - (float)j, sz - (float)k);
		} else {
			const TapsB T = make_taps_b(g, s_nbr, s_b4p, L.org, sx, sy, sz);
			v = tri_v_b(ru, T);
		}
		if (pass == 0) {
			vf = v;
			sx = sx + scaled_dt * v.x, sy = sy + scaled_dt * v.y, sz = sz + scaled_dt * v.z;  // Kernel.cu:387
		} else {
			vb = v;
		}
	}
	f3 vc = {vf.x + 0.5f * (vo.x - vb.x), vf.y + 0.5f * (vo.y - vb.y), vf.z + 0.5f * (vo.z - vb.z)};
	const int e[6] = {ob - kVX, ob + kVX, ob - kVY, ob + kVY, ob - 1, ob + 1};
	f3 mn = vo, mx = vo;
#pragma unroll
	for (int d = 0; d < 6; ++d) {
		const V3 t = box_v3(s_box, e[d]);
		mn.x = fminf(mn.x, t.xy.x);
		mx.x = fmaxf(mx.x, t.xy.x);
		mn.y = fminf(mn.y, t.xy.y);
		mx.y = fmaxf(mx.y, t.xy.y);
		mn.z = fminf(mn.z, t.z);
		mx.z = fmaxf(mx.z, t.z);
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
	st3(out, idx, vc);
}

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_vector(const GridDev g, const float* __restrict__ u, float* __restrict__ out,
                                                       const float* __restrict__ sdf, const float scaled_dt, const float inv_dx) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const f3 vo = ld3(u, idx);
	// forward pass (backtrace) then backward check: the same sampling code twice, kept as a 2-trip loop so that the
	// far-tap path is emitted once
	float sx = px - scaled_dt * vo.x, sy = py - scaled_dt * vo.y, sz = pz - scaled_dt * vo.z;  // backPos (Kernel.cu:374)
	float rx = px, ry = py, rz = pz;                                                            // where a collision sends the trace back to
	f3 vf = {0.0f, 0.0f, 0.0f}, vb = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		Taps T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
		if (COLL) {
			if (tri_f_t(sdf, T) < 0.0f) {  // Kernel.cu:377-382 / :390-394
				sx = rx, sy = ry, sz = rz;
				T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
			}
		}
		const f3 v = tri_v_t(u, T);
		if (pass == 0) {
			vf = v;
			rx = sx, ry = sy, rz = sz;  // fwdPos2 falls back to backPos
			sx = sx + scaled_dt * v.x, sy = sy + scaled_dt * v.y, sz = sz + scaled_dt * v.z;  // Kernel.cu:387
		} else {
			vb = v;
		}
	}
	f3 vc = {vf.x + 0.5f * (vo.x - vb.x), vf.y + 0.5f * (vo.y - vb.y), vf.z + 0.5f * (vo.z - vb.z)};

	int nb[6];
	nbr6(s_base, L.leaf, n, nb);
	f3 mn = vo, mx = vo;
#pragma unroll
	for (int d = 0; d < 6; ++d) {
		const f3 nv = ld3z(u, nb[d]);
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
	st3(out, idx, vc);
}

// ---------------------------------------------------------------------------------------------------------------
// advect_scalar (reference Kernel.cu:269-352): single field, nested-lerp trilinear
// ---------------------------------------------------------------------------------------------------------------

// 32-bit addressed form (no collision field)
__global__ __launch_bounds__(512) void k_advect_scalar_n(const GridDev g, const float* __restrict__ u, const float* __restrict__ in, float* __restrict__ out,
                                                         const float scaled_dt) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	__shared__ unsigned s_b4[27];
	__shared__ unsigned s_b4p[kPadTab];
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x, s_b4, s_b4p);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float px = (float)(L.org.x + (n >> 6)), py = (float)(L.org.y + ((n >> 3) & 7)), pz = (float)(L.org.z + (n & 7));
	const unsigned bytes1 = (unsigned)g.n_leaves * 2048u, own = (unsigned)idx << 2;
	const v4i ru = field_rsrc(u, bytes1 * 3u), rf = field_rsrc(in, bytes1);

	__shared__ float s_tile[kTile];  // clamp neighbours through LDS (see k_advect_vector_n)
	const float phiOrig = lds1(rf, own);
	s_tile[n] = phiOrig;
	if (n < 384) s_tile[512 + n] = lds1(rf, halo_off(s_b4, n));
	const f3 vc = ldv(ru, own);
	float sx = px - scaled_dt * vc.x, sy = py - scaled_dt * vc.y, sz = pz - scaled_dt * vc.z;
	float phiForward = 0.0f, phiBackward = 0.0f;
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		const TapsB T = make_taps_b(g, s_nbr, s_b4p, L.org, sx, sy, sz);
		const float phi = tri_f_b(rf, T);
		if (pass == 0) {
			phiForward = phi;
			const f3 vf = tri_v_b(ru, T);  // same eight taps as phiForward
			sx = sx + scaled_dt * vf.x, sy = sy + scaled_dt * vf.y, sz = sz + scaled_dt * vf.z;
		} else {
			phiBackward = phi;
		}
	}
	const float error = phiOrig - phiBackward;
	const float phiCorr = phiForward + 0.5f * error;
	__syncthreads();
	const int e[6] = {tile_nbr<0, -1>(n), tile_nbr<0, 1>(n), tile_nbr<1, -1>(n), tile_nbr<1, 1>(n), tile_nbr<2, -1>(n), tile_nbr<2, 1>(n)};
	float mn = phiOrig, mx = phiOrig;
#pragma unroll
	for (int d = 0; d < 6; ++d) {
		const float nv = s_tile[e[d]];
		mn = fminf(mn, nv);
		mx = fmaxf(mx, nv);
	}
	mn = fminf(mn, phiForward);
	mx = fmaxf(mx, phiForward);
	out[idx] = fmaxf(mn, fminf(phiCorr, mx));
}

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_scalar(const GridDev g, const float* __restrict__ u, const float* __restrict__ in,
                                                       float* __restrict__ out, const float* __restrict__ sdf, const float scaled_dt) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const float phiOrig = in[idx];
	const f3 vc = ld3(u, idx);
	float sx = px - scaled_dt * vc.x, sy = py - scaled_dt * vc.y, sz = pz - scaled_dt * vc.z;
	float rx = px, ry = py, rz = pz;
	float phiForward = 0.0f, phiBackward = 0.0f;
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		Taps T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
		if (COLL) {
			if (tri_f_t(sdf, T) < 0.0f) {
				sx = rx, sy = ry, sz = rz;
				T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
			}
		}
		const float phi = tri_f_t(in, T);
		if (pass == 0) {
			phiForward = phi;
			const f3 vf = tri_v_t(u, T);  // same eight taps as phiForward
			rx = sx, ry = sy, rz = sz;
			sx = sx + scaled_dt * vf.x, sy = sy + scaled_dt * vf.y, sz = sz + scaled_dt * vf.z;
		} else {
			phiBackward = phi;
		}
	}
	const float error = phiOrig - phiBackward;
	const float phiCorr = phiForward + 0.5f * error;
	int nb[6];
	nbr6(s_base, L.leaf, n, nb);
	float mn = phiOrig, mx = phiOrig;
#pragma unroll
	for (int d = 0; d < 6; ++d) {
		const float nv = ldz(in, nb[d]);
		mn = fminf(mn, nv);
		mx = fmaxf(mx, nv);
	}
	mn = fminf(mn, phiForward);
	mx = fmaxf(mx, phiForward);
	out[idx] = fmaxf(mn, fminf(phiCorr, mx));
}

// ---------------------------------------------------------------------------------------------------------------
// advect_scalars (reference Kernel.cu:118-266): one backtrace shared by up to HNS_MAX_SCALARS fields,
// weight-product trilinear, out-of-domain taps read ELEMENT g.oob (0 in the reference: Kernel.cu:133,192,225)
// ---------------------------------------------------------------------------------------------------------------

#define HNS_MAX_SCALARS 8
struct ScalarPtrs {
	const float* in[HNS_MAX_SCALARS];
	float* out[HNS_MAX_SCALARS];
	int n;
	// k_advect_scalars_n<true> (round 6): four more fields that arrive as ONE 16-byte element per voxel -- {fuel, waste, temperature, flame} as the fused
	// divergence / combustion kernel leaves them (hns_pressure.hip: CombustFuse) -- and leave as four float arrays like every other field
	const float* q4;
	float* q4_out[4];
};

// setupInterpolation (Kernel.cu:163-196): indices and weights in the order 000,100,010,110,001,101,011,111 of (x,y,z)
__device__ __forceinline__ void interp_from_taps(const Taps& T, int oob, int (&ix)[8], float (&w)[8]) {
	const float tx = T.fx, ty = T.fy, tz = T.fz;
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
	const int perm[8] = {0, 4, 2, 6, 1, 5, 3, 7};  // (di,dj,dk) at t[di*4+dj*2+dk]
#pragma unroll
	for (int q = 0; q < 8; ++q) ix[q] = T.t[perm[q]] < 0 ? oob : T.t[perm[q]];
}

// 32-bit addressed form (no collision field). Out-of-domain taps read element g.oob, as in the generic kernel.
// Q4: besides the P.n float fields, the four fields of P.q4. A corner tap of those four is ONE 16-byte gather instead of four 4-byte (z-paired: 8-byte) ones: the
// kernel is bound by L1 accesses per gather instruction (profiles/r05_advect_notes.txt 2), and a quad of lanes costs an access whatever its width -- sixteen
// gathers for the four fields' two samples instead of thirty-two. Per field the arithmetic is the same chain of fused multiply-adds in the same order.
// (at least four waves per SIMD = two workgroups per CU: the Q4 form sits at the 128-register line, and one register over it is ONE workgroup per CU -- 585 -> 838 us at 256^3, measured)
template <bool Q4>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(Q4 ? 6 : 4))) void k_advect_scalars_n(const GridDev g, const float* __restrict__ u, const ScalarPtrs P, const float scaled_dt) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	__shared__ unsigned s_b4[27];
	__shared__ unsigned s_b4p[kPadTab];
	const int n = threadIdx.x;
	const int leaf = launch_leaf(g, blockIdx.x);
	const int idx = leaf * 512 + n;
	const unsigned bytes1 = (unsigned)g.n_leaves * 2048u;
	const v4i ru = field_rsrc(u, bytes1 * 3u);
	const unsigned own = (unsigned)idx << 2, oob4 = (unsigned)g.oob << 2;
	const f3 vc = ldv(ru, own);  // (issued before the neighbour table is staged: see k_advect_vector_n)
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x, s_b4, s_b4p, leaf);
	const float px = (float)(L.org.x + (n >> 6)), py = (float)(L.org.y + ((n >> 3) & 7)), pz = (float)(L.org.z + (n & 7));

	// the leaf and one voxel around it through LDS (see k_advect_vector_n): the velocity once, then per field; shell cell of this thread (the first 488) and where its value lies
	const int ob = (((n >> 6) + 1) * 10 + ((n >> 3) & 7) + 1) * 10 + (n & 7) + 1;
	const int e[6] = {ob - 100, ob + 100, ob - 10, ob + 10, ob - 1, ob + 1};
	unsigned ho = 0u;
	int hcell = 0;
	if (n < kBoxShell) {
		int slot, local;
		box_shell_entry(n, slot, local, hcell);
		ho = s_b4[slot] + ((unsigned)local << 2);
	}
	ho = ho >= kOutside ? oob4 : ho;  // out-of-domain neighbours read element g.oob here (Kernel.cu:225)
	// FIRST: the first sample's taps out of the boxes too (below). Measured (profiles/r06_advect_box_ab.txt): it pays in the q4 form, which then fits 80 registers = six waves per SIMD,
	// and costs the float-only form a fifth (one more 12-byte gather per thread for the velocity shell, and a barrier in front of its first gathers)
	constexpr bool FIRST = Q4;
	__shared__ float s_ubox[FIRST ? 3 * kBox : 1];
	if constexpr (FIRST) {
		s_ubox[ob] = vc.x, s_ubox[ob + kBox] = vc.y, s_ubox[ob + 2 * kBox] = vc.z;
		if (n < kBoxShell) {
			const f3 h = ldv(ru, ho);
			s_ubox[hcell] = h.x, s_ubox[hcell + kBox] = h.y, s_ubox[hcell + 2 * kBox] = h.z;
		}
	}

	const float bx = px - scaled_dt * vc.x, by = py - scaled_dt * vc.y, bz = pz - scaled_dt * vc.z;
	unsigned bo[8], fo[8];
	float bw[8], fw[8];
	const int perm[8] = {0, 4, 2, 6, 1, 5, 3, 7};  // setupInterpolation's order 000,100,010,110,001,... of (x,y,z) (Kernel.cu:163-196)
	// Where the flow moves less than a voxel per step the FIRST sample point, too, lies among the voxel's 26 neighbours: its taps (velocity here, the fields' below) come out of
	// the boxes, per lane. Corner q of the interpolation order (x fastest: perm) sits at box offset (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2).
	const int bi = __float2int_rd(bx), bj = __float2int_rd(by), bk = __float2int_rd(bz);
	const unsigned sx_ = (unsigned)(bi - (L.org.x - 1)), sy_ = (unsigned)(bj - (L.org.y - 1)), sz_ = (unsigned)(bk - (L.org.z - 1));
	const bool bboxed = FIRST && max(sx_, max(sy_, sz_)) <= 8u;
	const int ba = bboxed ? (int)((sx_ * 10u + sy_) * 10u + sz_) : 0;  // box cell of the back cell's lower corner
	{
		float tx, ty, tz;
		if (bboxed) {
			tx = bx - (float)bi, ty = by - (float)bj, tz = bz - (float)bk;  // (make_taps_b's fractions)
#pragma unroll
			for (int q = 0; q < 8; ++q) bo[q] = 0u;
		} else {
			const TapsB T = make_taps_b(g, s_nbr, s_b4p, L.org, bx, by, bz);
			tx = T.fx, ty = T.fy, tz = T.fz;
#pragma unroll
			for (int q = 0; q < 8; ++q) bo[q] = T.o[perm[q]] >= kOutside ? oob4 : T.o[perm[q]];
		}
		const float itx = 1.0f - tx, ity = 1.0f - ty, itz = 1.0f - tz;
		const float w00 = itx * ity, w10 = tx * ity, w01 = itx * ty, w11 = tx * ty;
		bw[0] = w00 * itz, bw[1] = w10 * itz, bw[2] = w01 * itz, bw[3] = w11 * itz, bw[4] = w00 * tz, bw[5] = w10 * tz, bw[6] = w01 * tz, bw[7] = w11 * tz;
	}
	f3 vt[8];
	if (!bboxed) {
#pragma unroll
		for (int q = 0; q < 8; ++q) vt[q] = ldv(ru, bo[q]);
	}
	if constexpr (FIRST) __syncthreads();  // velocity box complete
	if (bboxed) {
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const int a = ba + (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2);
			vt[q] = f3{s_ubox[a], s_ubox[a + kBox], s_ubox[a + 2 * kBox]};
		}
	}
	f3 vf = {0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int q = 0; q < 8; ++q) {  // velF = velF + v * w (Kernel.cu:201-206), unfused
		vf.x = vf.x + bw[q] * vt[q].x;
		vf.y = vf.y + bw[q] * vt[q].y;
		vf.z = vf.z + bw[q] * vt[q].z;
	}
	// The second sample point is the voxel's own position up to s * (u(back) - u(own)): where it lands inside the leaf's 10^3 box (k_advect_vector_n, which see)
	// the fields' forward taps are read from the LDS box that the clamp needs anyway, not gathered
	const float qx = bx + scaled_dt * vf.x, qy = by + scaled_dt * vf.y, qz = bz + scaled_dt * vf.z;
	const int qi = __float2int_rd(qx), qj = __float2int_rd(qy), qk = __float2int_rd(qz);
	const unsigned rx = (unsigned)(qi - (L.org.x - 1)), ry = (unsigned)(qj - (L.org.y - 1)), rz = (unsigned)(qk - (L.org.z - 1));
	const bool boxed = max(rx, max(ry, rz)) <= 8u;  // (per lane)
	const int fa = boxed ? (int)((rx * 10u + ry) * 10u + rz) : 0;  // box cell of the forward cell's lower corner
	{
		float tx, ty, tz;
		if (boxed) {
			tx = qx - (float)qi, ty = qy - (float)qj, tz = qz - (float)qk;  // (make_taps_b's fractions)
#pragma unroll
			for (int q = 0; q < 8; ++q) fo[q] = 0u;
		} else {
			const TapsB T = make_taps_b(g, s_nbr, s_b4p, L.org, qx, qy, qz);
			tx = T.fx, ty = T.fy, tz = T.fz;
#pragma unroll
			for (int q = 0; q < 8; ++q) fo[q] = T.o[perm[q]] >= kOutside ? oob4 : T.o[perm[q]];
		}
		const float itx = 1.0f - tx, ity = 1.0f - ty, itz = 1.0f - tz;
		const float w00 = itx * ity, w10 = tx * ity, w01 = itx * ty, w11 = tx * ty;
		fw[0] = w00 * itz, fw[1] = w10 * itz, fw[2] = w01 * itz, fw[3] = w11 * itz, fw[4] = w00 * tz, fw[5] = w10 * tz, fw[6] = w01 * tz, fw[7] = w11 * tz;
	}
	// per field one own value per thread and one shell value per thread of the first 488; two boxes alternate so that one barrier per field suffices
	__shared__ float s_box[2][kBox];
	if constexpr (Q4) {
		__shared__ v4f32 s_box4[kBox];
		const v4i rq = field_rsrc(P.q4, bytes1 * 4u);  // element = 16 bytes: byte offset = 4 x the float-field byte offset
		const v4f32 phiOrig = hns_buffer_load_v4f32(rq, (int)(own << 2), 0, 0);
		s_box4[ob] = phiOrig;
		if (n < kBoxShell) s_box4[hcell] = hns_buffer_load_v4f32(rq, (int)(ho << 2), 0, 0);
		v4f32 phiF = {0.0f, 0.0f, 0.0f, 0.0f}, phiB = {0.0f, 0.0f, 0.0f, 0.0f};
		if (!bboxed) {
			v4f32 c[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) c[q] = hns_buffer_load_v4f32(rq, (int)(bo[q] << 2), 0, 0);
#pragma unroll
			for (int q = 0; q < 8; ++q) phiF = __builtin_elementwise_fma(c[q], v4f32{bw[q], bw[q], bw[q], bw[q]}, phiF);
		}
		if (!boxed) {
			v4f32 c[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) c[q] = hns_buffer_load_v4f32(rq, (int)(fo[q] << 2), 0, 0);
#pragma unroll
			for (int q = 0; q < 8; ++q) phiB = __builtin_elementwise_fma(c[q], v4f32{fw[q], fw[q], fw[q], fw[q]}, phiB);
		}
		__syncthreads();
		if (bboxed) {
#pragma unroll
			for (int q = 0; q < 8; ++q) phiF = __builtin_elementwise_fma(s_box4[ba + (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2)], v4f32{bw[q], bw[q], bw[q], bw[q]}, phiF);
		}
		if (boxed) {
#pragma unroll
			for (int q = 0; q < 8; ++q) phiB = __builtin_elementwise_fma(s_box4[fa + (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2)], v4f32{fw[q], fw[q], fw[q], fw[q]}, phiB);
		}
		const v4f32 error = phiOrig - phiB;
		const v4f32 phiCorr = __builtin_elementwise_fma(v4f32{0.5f, 0.5f, 0.5f, 0.5f}, error, phiF);
		v4f32 mn = phiOrig, mx = phiOrig;
#pragma unroll
		for (int d = 0; d < 6; ++d) {
			const v4f32 v = s_box4[e[d]];
			mn = __builtin_elementwise_min(mn, v);
			mx = __builtin_elementwise_max(mx, v);
		}
		mn = __builtin_elementwise_min(mn, phiF);
		mx = __builtin_elementwise_max(mx, phiF);
		const v4f32 r = __builtin_elementwise_max(mn, __builtin_elementwise_min(phiCorr, mx));
		P.q4_out[0][idx] = r.x, P.q4_out[1][idx] = r.y, P.q4_out[2][idx] = r.z, P.q4_out[3][idx] = r.w;
	}
	for (int s = 0; s < P.n; ++s) {
		const v4i rf = field_rsrc(P.in[s], bytes1);
		float* box = s_box[s & 1];
		const float phiOrig = lds1(rf, own);
		box[ob] = phiOrig;
		if (n < kBoxShell) box[hcell] = lds1(rf, ho);
		float vb[8], vf8[8];  // corner q and q+4 of the interpolation order differ only in z
		if (!bboxed) {
#pragma unroll
			for (int q = 0; q < 4; ++q) ld_zpair(rf, bo[q], bo[q + 4], vb[q], vb[q + 4]);
		}
		if (!boxed) {
#pragma unroll
			for (int q = 0; q < 4; ++q) ld_zpair(rf, fo[q], fo[q + 4], vf8[q], vf8[q + 4]);
		}
		float phiF = 0.0f, phiB = 0.0f;
		__syncthreads();
		if (bboxed) {
#pragma unroll
			for (int q = 0; q < 8; ++q) vb[q] = box[ba + (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2)];
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) phiF = __fmaf_rn(vb[q], bw[q], phiF);
		if (boxed) {
#pragma unroll
			for (int q = 0; q < 8; ++q) vf8[q] = box[fa + (q & 1) * 100 + ((q >> 1) & 1) * 10 + (q >> 2)];
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) phiB = __fmaf_rn(vf8[q], fw[q], phiB);
		const float error = phiOrig - phiB;
		const float phiCorr = __fmaf_rn(0.5f, error, phiF);
		float mn = phiOrig, mx = phiOrig;
#pragma unroll
		for (int d = 0; d < 6; ++d) {
			const float v = box[e[d]];
			mn = fminf(mn, v);
			mx = fmaxf(mx, v);
		}
		mn = fminf(mn, phiF);
		mx = fmaxf(mx, phiF);
		P.out[s][idx] = fmaxf(mn, fminf(phiCorr, mx));
	}
}

template <bool COLL>
__global__ __launch_bounds__(512) void k_advect_scalars(const GridDev g, const float* __restrict__ u, const ScalarPtrs P,
                                                        const float* __restrict__ sdf, const float scaled_dt) {
	__shared__ int s_nbr[27];
	__shared__ int s_base[27];
	const LeafCtx L = stage_leaf_base(g, s_nbr, s_base, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float px = (float)ci, py = (float)cj, pz = (float)ck;

	const f3 vc = ld3(u, idx);
	float sx = px - scaled_dt * vc.x, sy = py - scaled_dt * vc.y, sz = pz - scaled_dt * vc.z;
	float rx = px, ry = py, rz = pz;
	int bi[8], fi[8];
	float bw[8], fw[8];
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		Taps T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
		if (COLL) {  // the back-position test is made twice in the reference (Kernel.cu:142-155); the repeat cannot change the outcome
			if (tri_f_t(sdf, T) < 0.0f) {
				sx = rx, sy = ry, sz = rz;
				T = make_taps(g, s_nbr, s_base, L.org, sx, sy, sz);
			}
		}
		if (pass == 0) {
			interp_from_taps(T, g.oob, bi, bw);
			f3 vf = {0.0f, 0.0f, 0.0f};
#pragma unroll
			for (int q = 0; q < 8; ++q) {  // velF = velF + v * w (Kernel.cu:201-206), unfused
				const f3 v = ld3(u, bi[q]);
				vf.x = vf.x + bw[q] * v.x;
				vf.y = vf.y + bw[q] * v.y;
				vf.z = vf.z + bw[q] * v.z;
			}
			rx = sx, ry = sy, rz = sz;
			sx = sx + scaled_dt * vf.x, sy = sy + scaled_dt * vf.y, sz = sz + scaled_dt * vf.z;
		} else {
			interp_from_taps(T, g.oob, fi, fw);
		}
	}
	int nb[6];
	nbr6(s_base, L.leaf, n, nb);
#pragma unroll
	for (int d = 0; d < 6; ++d) nb[d] = nb[d] < 0 ? g.oob : nb[d];
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

}  // namespace hns

using namespace hns;

// the 32-bit addressed kernels apply while a Vec3f field stays below kNarrowBytes; option "advect" = generic forces the 64-bit ones (A/B, tests)
static bool narrow_fields(const hns_grid* g) {
	return !options().advect_generic.load() && (uint64_t)g->topo.n_leaves * 6144u <= hns::kNarrowBytes;
}

extern "C" {

int hns_dev_advect_vector(hns_grid* g, const float* vel3, float* out3, const float* sdf, int has_collision, float dt, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_vector")) return rc;
	NULLCHK(!vel3 || !out3, "hns_dev_advect_vector");
	if (vel3 == out3) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_advect_vector: output must not alias input");
	if (g->n_active == 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;  // Kernel.cu:361
	const dim3 grid((unsigned)g->n_active), block(512);
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_advect_vector<true>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, out3, sdf, scaled_dt, inv_dx);
	else if (narrow_fields(g))
		hipLaunchKernelGGL(k_advect_vector_n, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, out3, scaled_dt);
	else
		hipLaunchKernelGGL(k_advect_vector<false>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, out3, sdf, scaled_dt, inv_dx);
	return launch_status("hns_dev_advect_vector");
}

int hns_dev_advect_scalar(hns_grid* g, const float* vel3, const float* in, float* out, const float* sdf, int has_collision, float dt, float inv_dx,
                          void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_scalar")) return rc;
	NULLCHK(!vel3 || !in || !out, "hns_dev_advect_scalar");
	if (g->n_active == 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;
	const dim3 grid((unsigned)g->n_active), block(512);
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_advect_scalar<true>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, in, out, sdf, scaled_dt);
	else if (narrow_fields(g))
		hipLaunchKernelGGL(k_advect_scalar_n, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, in, out, scaled_dt);
	else
		hipLaunchKernelGGL(k_advect_scalar<false>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, in, out, sdf, scaled_dt);
	return launch_status("hns_dev_advect_scalar");
}

int hns_dev_advect_scalars(hns_grid* g, const float* vel3, const float* const* in, float* const* out, int n, const float* sdf, int has_collision,
                           float dt, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_advect_scalars")) return rc;
	NULLCHK(!vel3 || (n > 0 && (!in || !out)), "hns_dev_advect_scalars");
	if (g->n_active == 0 || n <= 0) return HNS_OK;
	const float scaled_dt = dt * inv_dx;
	const dim3 grid((unsigned)g->n_active), block(512);
	// the backtrace does not depend on the fields, so splitting S fields over several launches changes nothing numerically
	for (int base = 0; base < n; base += HNS_MAX_SCALARS) {
		ScalarPtrs P;
		P.n = n - base < HNS_MAX_SCALARS ? n - base : HNS_MAX_SCALARS;
		for (int s = 0; s < HNS_MAX_SCALARS; ++s) {
			P.q4 = nullptr, P.q4_out[0] = P.q4_out[1] = P.q4_out[2] = P.q4_out[3] = nullptr;
			P.in[s] = s < P.n ? in[base + s] : nullptr;
			P.out[s] = s < P.n ? out[base + s] : nullptr;
			if (s < P.n && (!P.in[s] || !P.out[s])) {
				set_error("hns_dev_advect_scalars: null device pointer for field %d", base + s);
				return HNS_ERR_INVALID_ARGUMENT;
			}
		}
		if (has_collision && sdf)
			hipLaunchKernelGGL(k_advect_scalars<true>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, P, sdf, scaled_dt);
		else if (narrow_fields(g))
		{
			GridDev gd = g->dev();
			// backwards: the gradient kernel has just written the velocity front to back; starting on its cached tail also
			// leaves the head cached for the next substep's advect_vector (256^3: -1 % here, -4 % there).
			gd.rev = 1;
			hipLaunchKernelGGL(k_advect_scalars_n<false>, grid, block, 0, (hipStream_t)stream, gd, vel3, P, scaled_dt);
		}
		else
			hipLaunchKernelGGL(k_advect_scalars<false>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, P, sdf, scaled_dt);
	}
	return launch_status("hns_dev_advect_scalars");
}

// advect_scalars over the four fields of `q4` (one 16-byte element per voxel in, four float arrays out) and n more float fields, one launch (hns_sim_substep; no collision
// field). Applies where hns_advect_q4_ok(g).
int hns_advect_scalars_q4(hns_grid* g, const float* vel3, const float* q4, float* const* q4_out, const float* const* in, float* const* out, int n, float dt, float inv_dx,
                          void* stream) {
	if (int rc = check_grid(g, "hns_advect_scalars_q4")) return rc;
	NULLCHK(!vel3 || !q4 || !q4_out || (n > 0 && (!in || !out)), "hns_advect_scalars_q4");
	if (!hns_advect_q4_ok(g) || n > HNS_MAX_SCALARS) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_advect_scalars_q4: grid too large for 32-bit offsets, or too many fields");
	if (g->n_active == 0) return HNS_OK;
	ScalarPtrs P;
	P.n = n;
	P.q4 = q4;
	for (int c = 0; c < 4; ++c) {
		NULLCHK(!q4_out[c], "hns_advect_scalars_q4");
		P.q4_out[c] = q4_out[c];
	}
	for (int s = 0; s < HNS_MAX_SCALARS; ++s) {
		P.in[s] = s < n ? in[s] : nullptr;
		P.out[s] = s < n ? out[s] : nullptr;
		NULLCHK(s < n && (!P.in[s] || !P.out[s]), "hns_advect_scalars_q4");
	}
	GridDev gd = g->dev();
	gd.rev = 1;  // (as hns_dev_advect_scalars)
	hipLaunchKernelGGL(k_advect_scalars_n<true>, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, gd, vel3, P, dt * inv_dx);
	return launch_status("hns_advect_scalars_q4");
}

// can this grid's fields take the q4 path (32-bit byte offsets into a 16-byte-per-voxel array)?
bool hns_advect_q4_ok(const hns_grid* g) { return narrow_fields(g) && (uint64_t)g->topo.n_leaves * 8192u <= hns::kNarrowBytes; }

}  // extern "C"
