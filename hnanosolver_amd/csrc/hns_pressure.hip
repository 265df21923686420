// hns_pressure.hip -- divergence, red-black SOR (all forms) and pressure-gradient subtraction: hand-written HIP for gfx950 / CDNA4.
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
//   * velocity is Vec3f AoS on the device (the host layout): one 12-byte access per tap, and every
//     2 KB leaf payload is one fully coalesced run;
//   * the red-black SOR loop runs ONE launch per iteration: a wave stages its leaves plus a two-voxel halo of
//     p in LDS, recomputes the red updates of the face-adjacent halo voxels itself (bit-identical to what the
//     neighbouring wave computes), then does black, ping-ponging p_in -> p_out. 12 B/voxel/iteration of HBM
//     traffic instead of the >=16 B of two in-place launches.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "hns_device.hpp"
#include "hns_flags.hpp"

namespace hns {

// ---------------------------------------------------------------------------------------------------------------
// divergence (reference Kernel.cu:499-519 and :455-496)
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(512) void k_divergence(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const f3 c = ld3(u, idx);
	const float xp = (c.x + nbr_val3<0, 1, 0>(u, s_nbr, L.leaf, n)) * 0.5f;
	const float xm = (c.x + nbr_val3<0, -1, 0>(u, s_nbr, L.leaf, n)) * 0.5f;
	const float yp = (c.y + nbr_val3<1, 1, 1>(u, s_nbr, L.leaf, n)) * 0.5f;
	const float ym = (c.y + nbr_val3<1, -1, 1>(u, s_nbr, L.leaf, n)) * 0.5f;
	const float zp = (c.z + nbr_val3<2, 1, 2>(u, s_nbr, L.leaf, n)) * 0.5f;
	const float zm = (c.z + nbr_val3<2, -1, 2>(u, s_nbr, L.leaf, n)) * 0.5f;
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
// subtractPressureGradient (reference Kernel.cu:765-829 / :694-762)
// ---------------------------------------------------------------------------------------------------------------

template <bool COLL>
__global__ __launch_bounds__(512) void k_subtract_gradient(const GridDev g, const float* u, const float* __restrict__ p, float* out,
                                                           const float* __restrict__ sdf, const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float gx = ((nbr_val<0, 1>(p, s_nbr, L.leaf, n) - nbr_val<0, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	const float gy = ((nbr_val<1, 1>(p, s_nbr, L.leaf, n) - nbr_val<1, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	const float gz = ((nbr_val<2, 1>(p, s_nbr, L.leaf, n) - nbr_val<2, -1>(p, s_nbr, L.leaf, n)) * 0.5f) * inv_dx;
	const f3 us = ld3(u, idx);  // `out` may alias `u`: each voxel reads only its own velocity (PressureProjection.cu:64)
	f3 r = {us.x - gx, us.y - gy, us.z - gz};
	if (COLL) {  // Kernel.cu:809-826
		const float sv = sdf[idx];
		if (sv < 0.0f) {
			r.x = r.y = r.z = 0.0f;
		} else if (sv < 0.1f) {
			const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
			const f3 nrm = sdf_normal(g, s_nbr, L.org, sdf, ci, cj, ck, inv_dx);
			r = no_slip_blend(r, nrm, 1.0f - (sv / 0.1f));
		}
	}
	st3(out, idx, r);
}

// Streaming form (no collision field): one wave per leaf. The pressure of the leaf and of the six touching face layers is
// staged in LDS (own values: two 16-byte loads per lane; a face layer: one value per lane); the velocity then streams
// through as 384 fully coalesced 16-byte chunks per leaf, each float subtracting the gradient component it belongs to
// (float f of the leaf: voxel f / 3, component f % 3, taps along that axis only). Same expression per component:
// ((p(+) - p(-)) * 0.5f) * inv_dx. The 512-thread form issues ~12 load/store instructions per wave of 64 voxels, mostly
// 12-byte and scattered 4-byte accesses; this one issues 20 per 512 voxels.
// M = NoMirror, or PhaseMirror for a chained multi-GPU rank (hns_flags.hpp): the boundary leaves' new velocity also goes, whole
// leaves, into the peers' ghost copies
template <class M>
__global__ __launch_bounds__(64) void k_subtract_gradient_s(const GridDev g, const float* u, const float* __restrict__ p, float* out, const float inv_dx, const M m) {
	__shared__ __attribute__((aligned(16))) float P[kTile];
	const int l = threadIdx.x;
	const int* __restrict__ rec = g.blk + (size_t)launch_pos(g, blockIdx.x) * 28;
	const int leaf = __builtin_amdgcn_readfirstlane(rec[0]);
	chain_begin(m, leaf);
	{
		const float4* q = reinterpret_cast<const float4*>(p + (size_t)leaf * 512 + l * 8);
		const float4 a = q[0], b = q[1];
		*reinterpret_cast<float4*>(&P[l * 8]) = a;
		*reinterpret_cast<float4*>(&P[l * 8 + 4]) = b;
	}
#pragma unroll
	for (int f = 0; f < 6; ++f) {  // face f: -x,+x,-y,+y,-z,+z; its 64 entries are one per lane
		int slot, local;
		halo_entry(f * 64 + l, slot, local);
		const int nb = __builtin_amdgcn_readfirstlane(rec[1 + slot]);
		const float v = p[(size_t)(nb < 0 ? 0 : nb) * 512 + local];
		P[512 + f * 64 + l] = nb < 0 ? 0.0f : v;
	}
	__syncthreads();
	const float4* src = reinterpret_cast<const float4*>(u + (size_t)leaf * 1536);
	float4* dst = reinterpret_cast<float4*>(out + (size_t)leaf * 1536);
	float4 a[6];
#pragma unroll
	for (int j = 0; j < 6; ++j) a[j] = src[l + 64 * j];
#pragma unroll
	for (int j = 0; j < 6; ++j) {
		float r[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			const int f = 4 * (l + 64 * j) + e;
			const int v = __mul24(f, 683) >> 11;  // f / 3 for f < 1536
			const int comp = f - 3 * v;
			const int shift = comp == 0 ? 6 : (comp == 1 ? 3 : 0);
			const int c = (v >> shift) & 7;
			const int x = v >> 6, y = (v >> 3) & 7, z = v & 7;
			const int ab = comp == 0 ? ((y << 3) | z) : (comp == 1 ? ((x << 3) | z) : ((x << 3) | y));
			const int plus = c != 7 ? v + (1 << shift) : 512 + 64 * (2 * comp + 1) + ab;
			const int minus = c != 0 ? v - (1 << shift) : 512 + 64 * (2 * comp) + ab;
			r[e] = r[e] - ((P[plus] - P[minus]) * 0.5f) * inv_dx;
		}
		dst[l + 64 * j] = make_float4(r[0], r[1], r[2], r[3]);
		chain_store_leaf16<3>(m, 0, leaf, 4 * (l + 64 * j), make_float4(r[0], r[1], r[2], r[3]));
	}
	chain_end(m, leaf);
}

// ---------------------------------------------------------------------------------------------------------------
// divergence, one wave per leaf (the production form)
// ---------------------------------------------------------------------------------------------------------------
//
// Same idea as the SOR kernels: lane = x*8+y owns a z-row, the row is read and written as 16-byte accesses (a Vec3f row
// is 96 contiguous bytes, a float row 32), lateral neighbours travel through a wave-private LDS tile, the +-z neighbours
// are in the lane's own registers. The 512-thread forms above issue 7-8 scalar taps per voxel and are bound by the
// texture addresser (one L1 access per 4 lanes per instruction); these issue ~1/5 of the accesses.

#define RT_ROW(xp, yp) (((xp) + 1) * 10 + ((yp) + 1))  // rows (x',y') in [-1,8]^2

struct RowTile {
	float4 lo[100], hi[100];
};
__device__ __forceinline__ void rt_put(RowTile& T, int R, const float (&v)[8]) {
	T.lo[R] = make_float4(v[0], v[1], v[2], v[3]);
	T.hi[R] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void rt_get(const RowTile& T, int R, float (&v)[8]) {
	const float4 a = T.lo[R], b = T.hi[R];
	v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}
// the 24 floats of Vec3f row `row` of leaf `leaf` (zeros when the leaf is absent)
__device__ __forceinline__ void glb_row3(const float* u, int leaf, int row, float (&r)[24]) {
	const float4* q = reinterpret_cast<const float4*>(u + ((size_t)(leaf < 0 ? 0 : leaf) * 512 + row * 8) * 3);
	const bool ok = leaf >= 0;
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		const float4 v = q[k];
		r[4 * k] = ok ? v.x : 0.0f, r[4 * k + 1] = ok ? v.y : 0.0f, r[4 * k + 2] = ok ? v.z : 0.0f, r[4 * k + 3] = ok ? v.w : 0.0f;
	}
}

// face-row duty of lane l < 32: face f = l>>3 (-x,+x,-y,+y), row i = l&7 -> neighbour slot, source row, tile row
__device__ __forceinline__ void face_duty(int l, int& slot, int& src, int& R) {
	const int f = (l >> 3) & 3, i = l & 7;
	slot = f == 0 ? 4 : (f == 1 ? 22 : (f == 2 ? 10 : 16));
	src = f == 0 ? 56 + i : (f == 1 ? i : (f == 2 ? i * 8 + 7 : i * 8));
	R = f == 0 ? RT_ROW(-1, i) : (f == 1 ? RT_ROW(8, i) : (f == 2 ? RT_ROW(i, -1) : RT_ROW(i, 8)));
}

// What hns_sim_substep fuses into the divergence launch (round 6; reference HNanoSolver.cu:181-234 runs three launches: divergence, combustion_oxygen, temperature_buoyancy).
// The lane that owns a z-row of the leaf has its eight divergences and its eight velocities in registers: it also burns the row's eight voxels
// (combustion_oxygen, Kernel.cu:923-966: div += burn * expansion), writes {fuel, waste, temperature, flame} as ONE 16-byte element per voxel (`q4`: what
// k_advect_scalars_n<true> gathers its taps from, hns_advect.hip) and the velocity with the buoyancy of the NEW temperature (temperature_buoyancy, Kernel.cu:831-847)
// into a second buffer -- the neighbours' divergences still read the un-buoyed u*. 60 B/voxel in one pass instead of 16 + 40 + 28 in three.
struct NoFuse {};
struct CombustFuse {
	const float* fuel;
	const float* waste;
	const float* temp;
	const float* flame;
	float4* q4;    // out: {fuel, waste, temperature, flame} after combustion, one element per voxel
	float* u_out;  // out: u* + buoyancy
	float temp_gain, expansion, dt, ambient, strength;
};
__device__ __forceinline__ void ld_row8(const float* f, size_t at, float (&v)[8]) {
	const float4* q = reinterpret_cast<const float4*>(f + at);
	const float4 a = q[0], b = q[1];
	v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}

// divergence (reference Kernel.cu:499-519): (xp - xm + yp - ym + zp - zm) * inv_dx with xp = (c.x + u(+x).x) * 0.5f, ...
// COAL (round 4; grids of 16k leaves and more, hns_dev_divergence): the leaf's own 6 KB are fetched in memory order -- lane l takes the
// 16-byte pieces l, 64 + l, ... so that an instruction touches 8 whole cache lines instead of a piece of each of the 48 -- and handed to
// the row owners through LDS (the wave's own 6 KB; one wave per workgroup, so only the wave's LDS order matters). Round 3 measured it
// (64.1 -> 62.9 us at 256^3, 133 -> 123 on the 66k-leaf plume, but 11.8 -> 13.4 at 128^3: 12.6 KB of LDS per wave halve the waves in
// flight where the grid is small) and dropped it; it is now switched by size instead.
// ZP (round 6, VERDICT r5 task 5): TWO waves per workgroup on the z-adjacent leaves 2q and 2q + 1 of the launch range (leaf order is z-fastest: on a box every such pair
// shares a z face); each hands the other the layer of u.z the other would gather -- 64 floats 96 bytes apart, every lane on a cache line of its own -- through LDS. The
// pair is checked against the neighbour table: leaves that are not z neighbours gather as before.
template <class M, bool COAL, class F, bool ZP = false>
__device__ __forceinline__ void divergence_row_body(const GridDev& g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx, const M& m, const F& fz) {
	constexpr bool FUSE = !std::is_same<F, NoFuse>::value;
	constexpr int NW = ZP ? 2 : 1;
	// ux rows (x faces), uy rows (y faces), and the hand-over buffer of the coalesced form. FUSE: all three carved out of one array, which the out-going q4 / velocity rows
	// are then staged in once the divergence has been read out of the tiles (576 float4: 64 rows x (8 + 1 pad))
	__shared__ __attribute__((aligned(16))) RowTile TXs[NW], TYs[NW];
	__shared__ __attribute__((aligned(16))) float4 s_owns[NW][COAL ? 384 : 1];
	__shared__ __attribute__((aligned(16))) float4 s_mems[NW][FUSE ? (COAL ? 784 : 576) : 1];
	__shared__ float s_z[NW][ZP ? 64 : 1];
	const int wv = ZP ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
	float4* const s_mem = s_mems[wv];
	RowTile& TX = FUSE ? *reinterpret_cast<RowTile*>(s_mem) : TXs[wv];
	RowTile& TY = FUSE ? *reinterpret_cast<RowTile*>(s_mem + 200) : TYs[wv];
	float4* const s_own = FUSE ? s_mem + 400 : s_owns[wv];
	const int l = threadIdx.x & 63, x = l >> 3, y = l & 7;
	const int* __restrict__ rec;
	int leaf;
	bool live = true, z_shared = false;
	if constexpr (ZP) {
		// pair q of the range in the chunked order of hns_device.hpp (one chunk of the pair list per XCD), walked backwards like the one-leaf form
		const int np = (g.n_active + 1) >> 1;
		const unsigned rows = (unsigned)np >> 3, b = blockIdx.x;
		const unsigned pos = (b >> 3) < rows ? (((rows - 1u - (b >> 3)) << 3) | (b & 7u)) : b;
		const int q = sched_leaf((int)pos, np, g.sched_seg > 1 ? g.sched_seg / 2 : 0);
		const int lf = g.first + 2 * q + wv;
		live = lf < g.first + g.n_active;
		leaf = live ? lf : g.first + 2 * q;  // (an odd range: the last pair's second wave shadows the first and stores nothing)
		rec = g.nbr27 + (size_t)leaf * 27 - 1;  // (rec[1 + j] = neighbour j, as in the launch-order records)
	} else {
		rec = g.blk + (size_t)launch_pos(g, blockIdx.x) * 28;
		leaf = __builtin_amdgcn_readfirstlane(rec[0]);
	}
	chain_begin(m, leaf);
	const int n_zm = __builtin_amdgcn_readfirstlane(rec[1 + 12]), n_zp = __builtin_amdgcn_readfirstlane(rec[1 + 14]);
	if constexpr (ZP) {
		const int last = g.first + g.n_active - 1;
		z_shared = wv == 0 ? (leaf + 1 <= last && n_zp == leaf + 1) : (live && n_zm == leaf - 1);
	}
	float r[24];
	float c_fu[8], c_wa[8], c_te[8], c_fl[8];  // FUSE: the row's four combustion fields (32 contiguous bytes each)
	if constexpr (FUSE) {
		const size_t at = (size_t)leaf * 512 + l * 8;
		ld_row8(fz.fuel, at, c_fu), ld_row8(fz.waste, at, c_wa), ld_row8(fz.temp, at, c_te), ld_row8(fz.flame, at, c_fl);
	}
	if constexpr (COAL) {
		const float4* q = reinterpret_cast<const float4*>(u + (size_t)leaf * 1536);
		float4 v[6];
#pragma unroll
		for (int k = 0; k < 6; ++k) v[k] = q[k * 64 + l];
#pragma unroll
		for (int k = 0; k < 6; ++k) s_own[k * 64 + l] = v[k];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const float4 w = s_own[l * 6 + k];
			r[4 * k] = w.x, r[4 * k + 1] = w.y, r[4 * k + 2] = w.z, r[4 * k + 3] = w.w;
		}
	} else {
		glb_row3(u, leaf, l, r);
	}
	float uz_m = (n_zm < 0 || (ZP && z_shared && wv == 1)) ? 0.0f : u[((size_t)n_zm * 512 + l * 8 + 7) * 3 + 2];
	float uz_p = (n_zp < 0 || (ZP && z_shared && wv == 0)) ? 0.0f : u[((size_t)n_zp * 512 + l * 8) * 3 + 2];
	if constexpr (ZP) s_z[wv][l] = wv == 0 ? r[23] : r[2];  // (u.z of the row's z = 7 / z = 0 voxel: what the partner wave would gather; read behind the workgroup barrier below)
	int slot, src, RF;
	face_duty(l, slot, src, RF);
	float h[24];
	if constexpr (COAL) {
		// (round 5) the four lateral face layers in memory order too: face f's eight Vec3f rows are 8 x 96 bytes -- one contiguous 768-byte run for the x faces, eight 96-byte runs for the y
		// faces -- fetched by 48 lanes, a 16-byte piece each, ONE instruction per face (12-16 L1 accesses) instead of one row per lane (32 lanes x six instructions, every lane on a cache
		// line of its own: 192 accesses); handed to the face-row lanes through the hand-over buffer, which the own rows have left by then. A timing-only build without these loads
		// ran 10 us faster at 256^3 (profiles/r05_divergence_face_bounds.txt).
		float4 fv[4];
		const int fr = l / 6, fk = l - fr * 6;  // lane -> (row of the face, piece of the row), l < 48
#pragma unroll
		for (int f = 0; f < 4; ++f) {
			const int nf = __builtin_amdgcn_readfirstlane(rec[1 + (f == 0 ? 4 : (f == 1 ? 22 : (f == 2 ? 10 : 16)))]);
			const int srow = f == 0 ? 56 + fr : (f == 1 ? fr : (f == 2 ? fr * 8 + 7 : fr * 8));
			const float4* q = reinterpret_cast<const float4*>(u + ((size_t)(nf < 0 ? 0 : nf) * 512 + srow * 8) * 3) + fk;
			fv[f] = (nf >= 0 && l < 48) ? *q : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();  // (every lane has read its own row out of s_own)
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (l < 48) {
#pragma unroll
			for (int f = 0; f < 4; ++f) s_own[f * 48 + l] = fv[f];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (l < 32) {
			const int f = l >> 3, i = l & 7;
#pragma unroll
			for (int k = 0; k < 6; ++k) {
				const float4 w = s_own[f * 48 + i * 6 + k];
				h[4 * k] = w.x, h[4 * k + 1] = w.y, h[4 * k + 2] = w.z, h[4 * k + 3] = w.w;
			}
		}
	} else {
		if (l < 32) glb_row3(u, rec[1 + slot], src, h);
	}

	float ux[8], uy[8];
#pragma unroll
	for (int z = 0; z < 8; ++z) ux[z] = r[3 * z], uy[z] = r[3 * z + 1];
	rt_put(TX, RT_ROW(x, y), ux);
	rt_put(TY, RT_ROW(x, y), uy);
	if (l < 32) {
		float hv[8];
		const int comp = l < 16 ? 0 : 1;  // x faces carry u.x, y faces u.y
#pragma unroll
		for (int z = 0; z < 8; ++z) hv[z] = comp ? h[3 * z + 1] : h[3 * z];
		rt_put(l < 16 ? TX : TY, RF, hv);
	}
	__syncthreads();
	if constexpr (ZP) {
		if (z_shared) {  // (wave-uniform)
			if (wv == 0) uz_p = s_z[1][l];
			else uz_m = s_z[0][l];
		}
	}
	float xp[8], xm[8], yp[8], ym[8];
	rt_get(TX, RT_ROW(x + 1, y), xp);
	rt_get(TX, RT_ROW(x - 1, y), xm);
	rt_get(TY, RT_ROW(x, y + 1), yp);
	rt_get(TY, RT_ROW(x, y - 1), ym);
	float d[8];
#pragma unroll
	for (int z = 0; z < 8; ++z) {
		const float cx = r[3 * z], cy = r[3 * z + 1], cz = r[3 * z + 2];
		const float zpv = z < 7 ? r[3 * (z < 7 ? z + 1 : 7) + 2] : uz_p;
		const float zmv = z > 0 ? r[3 * (z > 0 ? z - 1 : 0) + 2] : uz_m;
		const float a = (cx + xp[z]) * 0.5f, b = (cx + xm[z]) * 0.5f;
		const float c = (cy + yp[z]) * 0.5f, e = (cy + ym[z]) * 0.5f;
		const float f = (cz + zpv) * 0.5f, gg = (cz + zmv) * 0.5f;
		d[z] = (a - b + c - e + f - gg) * inv_dx;
	}
	if constexpr (FUSE) {
		// branch-free (selects between bit-exact alternatives): with branches the compiler carries r[] and d[] through them as whole vectors and spills.
		// The lane's eight q4 elements (128 bytes) and its velocity row (96 bytes) leave through LDS in MEMORY order -- lane l stores the 16-byte pieces l, 64 + l, ... --
		// instead of 16 bytes per lane at a 128 / 96-byte stride (measured at 256^3: 285 us for the launch, more than the three separate launches' 247).
		auto wave_sync = [] {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		};
		float ro[24];
		wave_sync();  // (every lane has read its neighbours' rows out of the tiles)
#pragma unroll
		for (int z = 0; z < 8; ++z) {
			// combustion_oxygen (Kernel.cu:923-966), the expressions of k_combustion_oxygen (hns_pointwise.hip)
			const float f0 = c_fu[z], wa = c_wa[z], fl = c_fl[z], t0 = c_te[z];
			const float fu = f0 < 0.001f ? 0.0f : f0;
			const float oxy = 1.0f - fu - wa;
			const bool burns = !(oxy < 0.0f);  // (no oxygen left: the voxel passes through)
			const float burn = fminf(oxy, fu);
			const float te = burns ? t0 + burn * fz.temp_gain : t0;
			s_mem[l * 9 + z] = make_float4(burns ? fu - burn : fu, burns ? wa + burn * 2.0f : wa, te, burns ? fmaxf(fl, fminf(1.0f, burn * 10.0f)) : fl);
			const float dd = d[z] + burn * fz.expansion;
			d[z] = burns ? dd : d[z];
			// temperature_buoyancy (Kernel.cu:831-847) with the temperature combustion has just written; x and z go through the same add as in
			// k_temperature_buoyancy (-0 + 0 = +0)
			const bool hot = !(te <= fz.ambient);
			const float tempDiff = te - fz.ambient;
			const float bx = r[3 * z] + fz.dt * 0.0f, by = r[3 * z + 1] + fz.dt * fmaxf(0.0f, tempDiff * fz.strength), bz = r[3 * z + 2] + fz.dt * 0.0f;
			ro[3 * z] = hot ? bx : r[3 * z], ro[3 * z + 1] = hot ? by : r[3 * z + 1], ro[3 * z + 2] = hot ? bz : r[3 * z + 2];
		}
		wave_sync();
		float4* q4 = fz.q4 + (size_t)leaf * 512;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int i = k * 64 + l;
			if (!ZP || live) q4[i] = s_mem[(i >> 3) * 9 + (i & 7)];
		}
		wave_sync();
#pragma unroll
		for (int k = 0; k < 6; ++k) s_mem[l * 7 + k] = make_float4(ro[4 * k], ro[4 * k + 1], ro[4 * k + 2], ro[4 * k + 3]);
		wave_sync();
		float4* uo = reinterpret_cast<float4*>(fz.u_out + (size_t)leaf * 1536);
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int i = k * 64 + l, row = __mul24(i, 10923) >> 16;  // i / 6 for i < 384
			if (!ZP || live) uo[i] = s_mem[row * 7 + (i - row * 6)];
		}
	}
	float4* q = reinterpret_cast<float4*>(div + (size_t)leaf * 512 + l * 8);
	if (!ZP || live) {
		q[0] = make_float4(d[0], d[1], d[2], d[3]);
		q[1] = make_float4(d[4], d[5], d[6], d[7]);
	}
	chain_store_row(m, 0, leaf, l, make_float4(d[0], d[1], d[2], d[3]), make_float4(d[4], d[5], d[6], d[7]));
	chain_end(m, leaf);
}
template <class M, bool COAL = false>
__global__ __launch_bounds__(64) void k_divergence_row(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx, const M m) {
	divergence_row_body<M, COAL, NoFuse>(g, u, div, inv_dx, m, NoFuse{});
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3))) void k_divergence_combust_buoyancy_zpair(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx,
                                                                                                                      const CombustFuse fz) {
	divergence_row_body<NoMirror, true, CombustFuse, true>(g, u, div, inv_dx, NoMirror{}, fz);
}
template <bool COAL>
__global__ __launch_bounds__(128) void k_divergence_zpair(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx) {
	divergence_row_body<NoMirror, COAL, NoFuse, true>(g, u, div, inv_dx, NoMirror{}, NoFuse{});
}
// the fused launch of hns_sim_substep (CombustFuse above). Four waves per SIMD: left alone the scheduler hoists every load and takes 256 registers (one wave per SIMD)
template <bool COAL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_divergence_combust_buoyancy(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx,
                                                                                                               const CombustFuse fz) {
	divergence_row_body<NoMirror, COAL, CombustFuse>(g, u, div, inv_dx, NoMirror{}, fz);
}

}  // namespace hns

using namespace hns;

extern "C" {

int hns_dev_divergence(hns_grid* g, const float* vel3, float* div, float inv_dx, void* stream) {
	if (int rc = check_grid(g, "hns_dev_divergence")) return rc;
	NULLCHK(!vel3 || !div, "hns_dev_divergence");
	if (g->n_active == 0) return HNS_OK;
	if (options().stencil_block.load() || !g->d_blk)  // option "stencil" = block: A/B switch
		hipLaunchKernelGGL(k_divergence, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), vel3, div, inv_dx);
	else
	{
		GridDev gd = g->dev();
		// backwards: advect_vector has just written the velocity front to back, so its tail is what the Infinity Cache holds
		// (256^3: 74 -> 66 us).
		gd.rev = 1;
		// option "divergence" = auto | row | coalesced: by size (k_divergence_row's COAL form from 16k leaves; loses below, see the kernel)
		const int form = options().divergence_form.load();
		// auto, from 16,384 leaves (round 6): z-adjacent leaves in pairs, the coalesced form (256^3 63 -> 58 us, 512^3 467 -> 431, 66k-leaf plume 118 -> 112;
		// below that size the pairing buys nothing and the coalesced form loses to the row form: profiles/r06_divergence_zpair_ab.txt). A range whose boundary leaves
		// are dealt out first (a chained multi-GPU rank) keeps one leaf per workgroup.
		if ((form == 3 || (form == 0 && g->n_active >= 16384)) && g->sched_pre == 0)
			hipLaunchKernelGGL(k_divergence_zpair<true>, dim3((unsigned)((g->n_active + 1) / 2)), dim3(128), 0, (hipStream_t)stream, gd, vel3, div, inv_dx);
		else if (form == 2 || (form == 0 && g->n_active >= 16384))
			hipLaunchKernelGGL((k_divergence_row<NoMirror, true>), dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, NoMirror{});
		else
			hipLaunchKernelGGL((k_divergence_row<NoMirror, false>), dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, NoMirror{});
	}
	return launch_status("hns_dev_divergence");
}

// divergence + combustion_oxygen + temperature_buoyancy as one launch (hns_sim_substep; see CombustFuse). Same form choice and launch order as hns_dev_divergence.
int hns_divergence_combust_buoyancy(hns_grid* g, const float* vel3, float* div, float inv_dx, const float* fuel, const float* waste, const float* temperature,
                                    const float* flame, float* q4, float* vel3_out, float temp_gain, float expansion, float dt, float ambient, float strength,
                                    void* stream) {
	if (int rc = check_grid(g, "hns_divergence_combust_buoyancy")) return rc;
	NULLCHK(!vel3 || !div || !fuel || !waste || !temperature || !flame || !q4 || !vel3_out, "hns_divergence_combust_buoyancy");
	if (vel3 == vel3_out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_divergence_combust_buoyancy: the buoyed velocity must not alias the velocity the divergence reads");
	if (g->n_active == 0) return HNS_OK;
	if (!g->d_blk) return fail(HNS_ERR_RUNTIME, "hns_divergence_combust_buoyancy: the grid has no launch tables");
	GridDev gd = g->dev();
	gd.rev = 1;
	const CombustFuse fz{fuel, waste, temperature, flame, reinterpret_cast<float4*>(q4), vel3_out, temp_gain, expansion, dt, ambient, strength};
	const int form = options().divergence_form.load();
	if ((form == 3 || (form == 0 && g->n_active >= 16384)) && g->sched_pre == 0)
		hipLaunchKernelGGL(k_divergence_combust_buoyancy_zpair, dim3((unsigned)((g->n_active + 1) / 2)), dim3(128), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, fz);
	else if (form == 2 || (form == 0 && g->n_active >= 16384))
		hipLaunchKernelGGL(k_divergence_combust_buoyancy<true>, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, fz);
	else
		hipLaunchKernelGGL(k_divergence_combust_buoyancy<false>, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, fz);
	return launch_status("hns_divergence_combust_buoyancy");
}

// the same kernels as ONE launch of a chained multi-GPU rank (hns_flags.hpp: PhaseMirror): boundary leaves first, their
// results also stored into the peers' ghost voxels
int hns_chain_divergence(hns_grid* g, const float* vel3, float* div, float inv_dx, const hns::PhaseMirror* m, void* stream) {
	if (int rc = check_grid(g, "hns_chain_divergence")) return rc;
	if (g->n_active == 0 || !g->d_blk) return fail(HNS_ERR_RUNTIME, "hns_chain_divergence: empty launch range");
	GridDev gd = g->dev();
	gd.rev = 1;  // (as hns_dev_divergence; walked backwards the boundary leaves come last, which the chain does not mind)
	hipLaunchKernelGGL(k_divergence_row<PhaseMirror>, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, gd, vel3, div, inv_dx, *m);
	return launch_status("hns_chain_divergence");
}
int hns_chain_subtract_pressure_gradient(hns_grid* g, const float* vel3, const float* p, float* out3, float inv_dx, const hns::PhaseMirror* m, void* stream) {
	if (int rc = check_grid(g, "hns_chain_subtract_pressure_gradient")) return rc;
	if (g->n_active == 0 || !g->d_blk) return fail(HNS_ERR_RUNTIME, "hns_chain_subtract_pressure_gradient: empty launch range");
	hipLaunchKernelGGL(k_subtract_gradient_s<PhaseMirror>, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, g->dev(), vel3, p, out3, inv_dx, *m);
	return launch_status("hns_chain_subtract_pressure_gradient");
}

int hns_dev_rbgs_color(hns_grid* g, const float* div, float* p, float dx, float omega, int color, void* stream) {
	if (int rc = check_grid(g, "hns_dev_rbgs_color")) return rc;
	NULLCHK(!div || !p, "hns_dev_rbgs_color");
	if (color != 0 && color != 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_color: color must be 0 (red) or 1 (black)");
	if (g->n_active == 0) return HNS_OK;
	hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, (hipStream_t)stream, g->dev(), div, p, dx * dx, omega, color);
	return launch_status("hns_dev_rbgs_color");
}

// One full (red, black) iteration src -> dst in the reference's own decomposition: copy, then one launch per colour in place (k_rbgs_color). The independent
// cross-check of the temporally blocked kernels (option "rbgs" = color), and what sweeps a grid they cannot (more than 2 M leaves).
static int launch_color_iteration(hns_grid* g, const GridDev& gd, const float* div, const float* src, float* dst, float dx2, float omega, hipStream_t st) {
	HNS_HIP(hipMemcpyAsync(dst, src, sizeof(float) * 512 * (size_t)g->topo.n_leaves, hipMemcpyDeviceToDevice, st));
	hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, st, gd, div, dst, dx2, omega, 0);
	hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, st, gd, div, dst, dx2, omega, 1);
	return HNS_OK;
}

int hns_dev_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                         void* stream) {
	return hns_rbgs_iterate(g, div, p_a, p_b, dx, omega, iterations, result_in_b, stream, false);
}

// from_zero: the solve starts from p = 0 (the reference never warm-starts, HNanoSolver.cu:113 / PressureProjection.cu:35) and
// p_a's content is irrelevant: the first blocked launch does not read it, so the caller need not clear it either.
int hns_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b, void* stream,
                     bool from_zero) {
	if (int rc = check_grid(g, "hns_dev_rbgs_iterate")) return rc;
	NULLCHK(!div || !p_a || !p_b, "hns_dev_rbgs_iterate");
	if (p_a == p_b) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: p_a and p_b must be distinct buffers");
	if (iterations < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: negative iteration count");
	if (result_in_b) *result_in_b = iterations & 1;
	hipStream_t st = (hipStream_t)stream;
	if (g->n_active == 0 || iterations == 0) {
		if (from_zero) HNS_HIP(hipMemsetAsync(p_a, 0, sizeof(float) * 512 * (size_t)g->topo.n_leaves, st));  // the result is p_a = 0
		return HNS_OK;
	}
	const float dx2 = dx * dx;  // Kernel.cu:608
	const GridDev gd = g->dev();
	// Temporally blocked (hns_sorblock.hip): k iterations per launch, p read and written once per launch -- 16^3 blocks two at a time, one-leaf blocks (small grids) two
	// or four; an odd iteration left over is one more launch of the same kernel with two colour sweeps. (A launch range -- a multi-GPU rank's boundary / interior leaves --
	// must never go through the two-launch form below: its copy of the whole field would undo what the launch over the neighbouring range has stored.)
	int k_max = 0;
	const int lb = options().rbgs.load() == kRbgsColor ? 0 : hns_rbgs_block_shape(g, &k_max);
	float* src = p_a;
	float* dst = p_b;
	int left = iterations, launches = 0;
	bool zero = from_zero;
	while (lb && left >= 2) {
		const int k = (k_max >= 4 && left >= 4) ? 4 : 2;
		if (int rc = hns_rbgs_block_launch(g, lb, k, zero, div, src, dst, dx2, omega, stream)) return rc;
		std::swap(src, dst);
		left -= k, ++launches, zero = false;
	}
	if (lb && left) {
		if (int rc = hns_rbgs_block_launch(g, lb, 1, zero, div, src, dst, dx2, omega, stream)) return rc;
		std::swap(src, dst);
		--left, ++launches, zero = false;
	}
	if (left && zero) {  // the two-launch form reads its input: give it the zeros
		HNS_HIP(hipMemsetAsync(src, 0, sizeof(float) * 512 * (size_t)g->topo.n_leaves, st));
		zero = false;
	}
	for (; left; --left, ++launches) {
		if (int rc = launch_color_iteration(g, gd, div, src, dst, dx2, omega, st)) return rc;
		std::swap(src, dst);
	}
	if (result_in_b) *result_in_b = launches & 1;
	return launch_status("hns_dev_rbgs_iterate");
}

int hns_grid_rbgs_plan(hns_grid* g, int iterations, char* description, uint64_t description_bytes, int* launches, int* iterations_per_launch) {
	if (int rc = check_grid(g, "hns_grid_rbgs_plan")) return rc;
	if (iterations < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_rbgs_plan: negative iteration count");
	int k_max = 0;
	const int lb = (iterations >= 1 && options().rbgs.load() != kRbgsColor) ? hns_rbgs_block_shape(g, &k_max) : 0;
	char buf[256];
	int n = 0, k = 1, left = iterations;
	if (lb) {
		while (left >= 2) left -= (k_max >= 4 && left >= 4) ? 4 : 2, ++n;
		if (left) --left, ++n;  // (an odd iteration left over: the same kernel, one iteration)
		k = iterations >= 2 ? k_max : 1;
		snprintf(buf, sizeof(buf), "k_rbgs_block%s<%d,%d>: %d red+black iterations per launch on %s blocks with a %d-voxel halo, p read and written once per launch%s", lb == 2 ? "_xy" : "", lb, k_max, k_max,
		         lb == 1 ? "one-leaf (8^3-voxel)" : "16^3-voxel", 2 * k_max, lb == 2 ? " (rows in LDS, three workgroups per CU, the sweep threads fetch their own rows)" : "");
	} else {
		snprintf(buf, sizeof(buf), "k_rbgs_color: two launches per iteration, in place (the reference's decomposition)");
	}
	n += 2 * left;  // (what is left goes through the two-launch form)
	if (description && description_bytes) snprintf(description, description_bytes, "%s", buf);
	if (launches) *launches = n;
	if (iterations_per_launch) *iterations_per_launch = k;
	return HNS_OK;
}

int hns_dev_time_rbgs(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int reps, float* ms_per_launch,
                      void* stream) {
	if (int rc = check_grid(g, "hns_dev_time_rbgs")) return rc;
	if (!ms_per_launch || iterations <= 0 || reps <= 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_time_rbgs: bad arguments");
	struct Events {  // destroyed on every return path
		hipEvent_t e0 = nullptr, e1 = nullptr;
		~Events() {
			if (e0) (void)hipEventDestroy(e0);
			if (e1) (void)hipEventDestroy(e1);
		}
	} ev;
	HNS_HIP(hipEventCreate(&ev.e0));
	HNS_HIP(hipEventCreate(&ev.e1));
	double total = 0.0;
	for (int r = 0; r < reps; ++r) {
		HNS_HIP(hipEventRecord(ev.e0, (hipStream_t)stream));
		if (int rc = hns_dev_rbgs_iterate(g, div, p_a, p_b, dx, omega, iterations, nullptr, stream)) return rc;
		HNS_HIP(hipEventRecord(ev.e1, (hipStream_t)stream));
		HNS_HIP(hipEventSynchronize(ev.e1));
		float ms = 0.0f;
		HNS_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
		total += ms;
	}
	*ms_per_launch = (float)(total / ((double)reps * iterations));
	return HNS_OK;
}

int hns_dev_subtract_pressure_gradient(hns_grid* g, const float* vel3, const float* p, float* out3, const float* sdf, int has_collision, float inv_dx,
                                       void* stream) {
	if (int rc = check_grid(g, "hns_dev_subtract_pressure_gradient")) return rc;
	NULLCHK(!vel3 || !p || !out3, "hns_dev_subtract_pressure_gradient");
	if (g->n_active == 0) return HNS_OK;
	const dim3 grid((unsigned)g->n_active), block(512);
	// (a wave-per-leaf row form like k_divergence_row was measured for this kernel too: 118 us vs 111 us at 256^3 -- not kept)
	// (also measured and not kept for this kernel: the six taps through an LDS tile as in the advection kernels, 125 vs 116 us --
	// at 470 MB per launch it streams from HBM and the extra barrier costs more than the loads it saves)
	const bool block_form = options().stencil_block.load() != 0;  // option "stencil" = block: A/B switch
	if (has_collision && sdf)
		hipLaunchKernelGGL(k_subtract_gradient<true>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, p, out3, sdf, inv_dx);
	else if (block_form || !g->d_blk)
		hipLaunchKernelGGL(k_subtract_gradient<false>, grid, block, 0, (hipStream_t)stream, g->dev(), vel3, p, out3, sdf, inv_dx);
	else
		hipLaunchKernelGGL(k_subtract_gradient_s<NoMirror>, grid, dim3(64), 0, (hipStream_t)stream, g->dev(), vel3, p, out3, inv_dx, NoMirror{});
	return launch_status("hns_dev_subtract_pressure_gradient");
}

}  // extern "C"
