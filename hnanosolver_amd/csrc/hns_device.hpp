// hns_device.hpp -- device-side helpers shared by the kernel files: topology access (replaces the NanoVDB accessor walk
// of reference src/Utils/Stencils.hpp:51-71), the samplers of Stencils.hpp:74-173, and the collision helpers of
// reference src/Cuda/Kernel.cu:8-74. Arithmetic keeps the reference's association; build with -ffp-contract=off.
#pragma once

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

// The eight corners of a trilinear cell whose lower corner (i, j, k) lies beyond the 27-leaf neighbourhood of the workgroup's leaf (origin `org`, neighbours `s_nbr` in LDS): a
// back-trace of more than a leaf, |u| dt / dx > 8. Same leaf ids, same indices as eight calls of tap_index; t[di*4 + dj*2 + dk].
// Round 6: up to two leaves away (|u| dt / dx < 16: where a fast plume's back-traces land) the leaf comes out of the neighbour tables in TWO HOPS -- the workgroup's own neighbour
// towards it (LDS), then that neighbour's row of nbr27 (one 4-byte load) -- instead of a walk through the origin hash (hash slot, then the candidate's origin, then the next
// slot: dependent loads, one walk after the other for the 1, 2, 4 or 8 leaves under the cell). The eight table loads are independent and issued together: one round trip. nbr27 rows
// are built from the same hash, so the answer is the hash's; the hash is still asked where the intermediate leaf is absent or the cell lies farther out.
__device__ __forceinline__ void far_cell_taps(const GridDev& g, const int* s_nbr, const int4 org, int i, int j, int k, int (&t)[8]) {
	const int i0 = i & ~7, j0 = j & ~7, k0 = k & ~7;
	// leaf offsets, relative to the workgroup's leaf, of the leaf under the lower corner and (where the cell crosses a leaf face: lower corner on the last voxel) the one above it
	const int ax[2] = {(i0 - org.x) >> 3, ((i0 - org.x) >> 3) + ((i & 7) == 7)}, ay[2] = {(j0 - org.y) >> 3, ((j0 - org.y) >> 3) + ((j & 7) == 7)},
	          az[2] = {(k0 - org.z) >> 3, ((k0 - org.z) >> 3) + ((k & 7) == 7)};
	int L[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const int a = ax[c >> 2], b = ay[(c >> 1) & 1], d = az[c & 1];
		const int a1 = max(-1, min(1, a)), b1 = max(-1, min(1, b)), d1 = max(-1, min(1, d));
		const bool two = max(max(abs(a), abs(b)), abs(d)) <= 2;
		const int n1 = two ? s_nbr[(a1 + 1) * 9 + (b1 + 1) * 3 + (d1 + 1)] : -1;
		L[c] = n1 >= 0 ? g.nbr27[n1 * 27 + (a - a1 + 1) * 9 + (b - b1 + 1) * 3 + (d - d1 + 1)] : -2;  // (-2: not in the tables' reach)
	}
	bool hash = false;
#pragma unroll
	for (int c = 0; c < 8; ++c) hash |= L[c] == -2;
	if (hash) {  // one walk per DISTINCT leaf under the cell: in two cells out of three all eight corners share one leaf
		const bool cx = (i & 7) == 7, cy = (j & 7) == 7, cz = (k & 7) == 7;
		L[0] = d_find_leaf(g, i0, j0, k0);
		L[1] = cz ? d_find_leaf(g, i0, j0, k0 + 8) : L[0];
		L[2] = cy ? d_find_leaf(g, i0, j0 + 8, k0) : L[0];
		L[3] = cy ? (cz ? d_find_leaf(g, i0, j0 + 8, k0 + 8) : L[2]) : L[1];
		if (cx) {
			L[4] = d_find_leaf(g, i0 + 8, j0, k0);
			L[5] = cz ? d_find_leaf(g, i0 + 8, j0, k0 + 8) : L[4];
			L[6] = cy ? d_find_leaf(g, i0 + 8, j0 + 8, k0) : L[4];
			L[7] = cy ? (cz ? d_find_leaf(g, i0 + 8, j0 + 8, k0 + 8) : L[6]) : L[5];
		} else {
			L[4] = L[0], L[5] = L[1], L[6] = L[2], L[7] = L[3];
		}
	}
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const int di = c >> 2, dj = (c >> 1) & 1, dk = c & 1;
		const int leaf = L[c];  // (corner c lies in the leaf of the same index: where an axis does not cross, L[] repeats the lower leaf)
		t[c] = leaf < 0 ? -1 : leaf * 512 + ((((i + di) & 7) << 6) | (((j + dj) & 7) << 3) | ((k + dk) & 7));
	}
}

// IndexSampler<float,0> (Stencils.hpp:81-89): value, or 0 outside the domain
__device__ __forceinline__ float ld0(const float* __restrict__ f, int idx) { return idx < 0 ? 0.0f : f[idx]; }

struct f3 {
	float x, y, z;
};

// float lerp of TrilinearSampler (Stencils.hpp:140): a + w*(b-a), unfused
__device__ __forceinline__ float lerp_f(float a, float b, float w) { return a + w * (b - a); }
// Vec3f lerp on the device branch (Stencils.hpp:131-135): fmaf(w, b-a, a)
__device__ __forceinline__ float lerp_c(float a, float b, float w) { return __fmaf_rn(w, b - a, a); }


// ---- launch order ------------------------------------------------------------------------------------------------
// Block -> leaf order. The dispatcher places workgroup b on XCD b % 8 (observed, not contractual), each XCD has a
// private 4 MiB L2, and a leaf's halo is its neighbours' payload: every XCD gets one contiguous chunk of the leaf
// list (sizes differ by at most one leaf), so halo reads hit the L2 that already holds those leaves. Speed only; any
// order is correct. Option "schedule" = linear disables it.
// `seg`: leaves per contiguous segment; consecutive segments go to consecutive XCDs. seg = 1 is plain leaf order, seg <= 0
// (or >= n/8) one chunk per XCD; in between, the eight XCDs walk through neighbouring stretches of memory together (DRAM
// pages, Infinity Cache) while a leaf's z / y neighbours still sit in its own L2.
__host__ __device__ inline int sched_leaf(int b, int n, int seg) {
	if (seg == 1) return b;
	int body = 0;
	if (seg > 1) {
		const int rows = n / (8 * seg);
		body = rows * 8 * seg;
		if (b < body) {
			const int x = b & 7, i = b >> 3;
			return ((i / seg) * 8 + x) * seg + i % seg;
		}
		b -= body, n -= body;
	}
	const int base = n >> 3, rem = n & 7;
	const int x = b & 7, i = b >> 3;  // rows i < base hold all eight XCDs; the last row only x < rem, and b - 8*base == x there
	return body + x * base + (x < rem ? x : rem) + i;
}

// `pre` (a multiple of 8, 0 = none): the first `pre` leaves are dealt out first, an equal contiguous piece to every XCD, the rest
// as above behind them -- a multi-GPU rank's boundary leaves (first in its leaf order) then run on all eight XCDs instead of
// filling the head of XCD 0's chunk (hns_dist_*.hip: their waves poll and signal, and are slower than the others).
__host__ __device__ inline int sched_leaf(int b, int n, int seg, int pre) {
	if (pre <= 0) return sched_leaf(b, n, seg);
	if (b < pre) return (b & 7) * (pre >> 3) + (b >> 3);
	return pre + sched_leaf(b - pre, n - pre, seg);
}


// ---- face-neighbour values through LDS ------------------------------------------------------------------------------
// Kernels that need the six face neighbours of every voxel of a leaf (BFECC clamps, pressure gradient) would issue six
// wave-wide loads per thread, and on gfx950 the L1 charges a load instruction 16 cycles per wave whatever it touches. A
// 512-thread workgroup instead stages a tile: its own 512 values plus the six 8x8 face layers of the neighbouring
// leaves (384 values, one per thread of the first six waves), and every thread reads its neighbours from LDS, branch-free.
// Tile entry e: [0,512) own voxel n; 512 + 64*f + (a*8+b): face layer f (-x,+x,-y,+y,-z,+z) of the neighbouring leaf,
// (a,b) = the two other coordinates in x,y,z order.
constexpr int kTile = 512 + 6 * 64;

// halo entry h in [0,384): slot of the neighbour leaf in the 27-table and the voxel of that leaf to fetch
__device__ __forceinline__ void halo_entry(int h, int& slot, int& local) {
	const int f = h >> 6, a = (h >> 3) & 7, b = h & 7;
	const int axis = f >> 1, dir = (f & 1) ? 1 : -1;
	const int c = dir > 0 ? 0 : 7;  // the layer of the neighbour that touches our face
	slot = 13 + dir * (axis == 0 ? 9 : (axis == 1 ? 3 : 1));
	local = axis == 0 ? ((c << 6) | (a << 3) | b) : (axis == 1 ? ((a << 6) | (c << 3) | b) : ((a << 6) | (b << 3) | c));
}

template <int AXIS, int DIR>
__device__ __forceinline__ int tile_nbr(int n) {  // tile entry of the face neighbour of own voxel n
	constexpr int shift = AXIS == 0 ? 6 : (AXIS == 1 ? 3 : 0);
	constexpr int f = 2 * AXIS + (DIR > 0 ? 1 : 0);
	const int c = (n >> shift) & 7;
	const bool inside = DIR > 0 ? c != 7 : c != 0;
	const int x = n >> 6, y = (n >> 3) & 7, z = n & 7;
	const int ab = AXIS == 0 ? ((y << 3) | z) : (AXIS == 1 ? ((x << 3) | z) : ((x << 3) | y));
	return inside ? n + DIR * (1 << shift) : 512 + 64 * f + ab;
}

// Stage the workgroup's leaf id, origin and 27-neighbour table. Returns false for an out-of-range block.
struct LeafCtx {
	int leaf;
	int4 org;
};

__device__ __forceinline__ LeafCtx stage_leaf(const GridDev& g, int* s_nbr, int block) {
	LeafCtx c;
	c.leaf = g.sched ? g.sched[block] : g.first + block;
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


// Vec3f element idx of an AoS velocity array (12-byte stride, exactly the reference's nanovdb::Vec3f[]): one 12-byte access
__device__ __forceinline__ f3 ld3(const float* u, int idx) {
	const float3 v = *reinterpret_cast<const float3*>(u + 3 * (size_t)idx);
	f3 r = {v.x, v.y, v.z};
	return r;
}
__device__ __forceinline__ f3 ld3z(const float* u, int idx) {  // zero outside the domain (IndexSampler<Vec3f,0>)
	const f3 v = ld3(u, idx < 0 ? 0 : idx);
	f3 r = {idx < 0 ? 0.0f : v.x, idx < 0 ? 0.0f : v.y, idx < 0 ? 0.0f : v.z};
	return r;
}
__device__ __forceinline__ void st3(float* u, int idx, f3 v) { *reinterpret_cast<float3*>(u + 3 * (size_t)idx) = make_float3(v.x, v.y, v.z); }

// component COMP of the velocity at the face neighbour of voxel n along AXIS (0 outside the domain)
template <int AXIS, int DIR, int COMP>
__device__ __forceinline__ float nbr_val3(const float* __restrict__ u, const int* s_nbr, int leaf, int n) {
	constexpr int shift = AXIS == 0 ? 6 : (AXIS == 1 ? 3 : 0);
	constexpr int stride = 1 << shift;
	const int c = (n >> shift) & 7;
	if (DIR > 0) {
		if (c != 7) return u[3 * (size_t)(leaf * 512 + n + stride) + COMP];
		const int nl = s_nbr[13 + (AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1))];
		return nl < 0 ? 0.0f : u[3 * (size_t)(nl * 512 + n - 7 * stride) + COMP];
	} else {
		if (c != 0) return u[3 * (size_t)(leaf * 512 + n - stride) + COMP];
		const int nl = s_nbr[13 - (AXIS == 0 ? 9 : (AXIS == 1 ? 3 : 1))];
		return nl < 0 ? 0.0f : u[3 * (size_t)(nl * 512 + n + 7 * stride) + COMP];
	}
}

}  // namespace hns

// ---- launcher plumbing shared by the kernel files ----

#define HNS_HIP(call)                                                                  \
	do {                                                                               \
		hipError_t e__ = (call);                                                       \
		if (e__ != hipSuccess) {                                                       \
			hns::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
			return HNS_ERR_HIP;                                                        \
		}                                                                              \
	} while (0)

inline int check_grid(const hns_grid* g, const char* who) {
	if (!g) {
		hns::set_error("%s: null grid", who);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	if (!g->on_device) {
		hns::set_error("%s: grid has no device tables (host-only grid or no HIP device); there is no CPU fallback", who);
		return HNS_ERR_NO_DEVICE;
	}
	int cur = -1;
	if (hipGetDevice(&cur) != hipSuccess || cur != g->device) {  // launches go to the CURRENT device; the grid's tables live on g->device
		hns::set_error("%s: the grid lives on HIP device %d but device %d is current in this thread", who, g->device, cur);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	return HNS_OK;
}

inline int launch_status(const char* who) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		hns::set_error("%s: kernel launch failed: %s", who, hipGetErrorString(e));
		return HNS_ERR_HIP;
	}
	return HNS_OK;
}

inline unsigned ew_blocks(uint64_t n) {
	uint64_t b = (n + 255) / 256;
	return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

#define NULLCHK(cond, who)                                \
	if (cond) {                                           \
		hns::set_error("%s: null device pointer", who);        \
		return HNS_ERR_INVALID_ARGUMENT;                  \
	}

