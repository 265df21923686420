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
// red-black SOR, fused form (one launch = one full (red, black) iteration, p_in -> p_out), ONE WAVE PER LEAF
// ---------------------------------------------------------------------------------------------------------------
//
// The wave updates the 256 red voxels of its leaf AND the 6*32 red voxels directly across each face (where that neighbour
// leaf exists; outside the domain p stays 0) -- a halo red voxel (x=-1,y,z) needs p at (-2,y,z), (0,y,z), (-1,y+-1,z),
// (-1,y,z+-1): face layers two deep and edge lines, never corners -- then the 256 black voxels from the new reds. The
// recomputed halo reds see the same inputs as the neighbouring wave's own update, so the result is bit-identical to the
// two-launch form (k_rbgs_color). Organised for the CDNA4 wave: the 64 lanes
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

// M = NoMirror, or PhaseMirror for a chained multi-GPU rank (hns_flags.hpp): the form for small ragged ranks (see rbgs_form)
template <class M>
__global__ __launch_bounds__(64) void k_rbgs_wave(const GridDev g, const float* __restrict__ div, const float* __restrict__ p_in,
                                                  float* __restrict__ p_out, const float dx2, const float omega, const M m) {
	__shared__ __attribute__((aligned(16))) float T[W_FLOATS];
	const int l = threadIdx.x;
	// per-block record {leaf, nbr27[27]} in launch order: one dependent scalar fetch instead of sched -> nbr27
	const int* __restrict__ rec = g.blk + (size_t)blockIdx.x * 28;
	const int leaf = __builtin_amdgcn_readfirstlane(rec[0]);
	chain_begin(m, leaf);
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
		chain_store_row(m, 0, leaf, l, o0, o1);
	}
	chain_end(m, leaf);
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
//   (x', -1) -> 87 + 8*x'      (== row (x',0) - 1   mod 16), except x' = 7 -> 86 (see below)
//   (x',  8) -> 80 + 8*x'      (== row (x',7) + 1   mod 16)
//   edge rows (-1,-1) (-1,8) (8,-1) (8,8) -> 81..84; depth-2 rows of halo lane h -> 89 + 8*(h/6) + h%6
// ZM / ZP hold the z=-1 values of the lower tile and the z=8 values of the upper tile (core rows and face rows).
// 137 rows = 9,864 bytes per wave: 16 waves per CU fit in the 160 KB of LDS (a 153-row numbering that kept row (7,-1)
// conflict-free too needed 11,016 bytes = 14 waves, i.e. 4.6 rounds of waves per launch at 256^3 instead of 4.0).

#define PR_ROWS 137
__host__ __device__ constexpr int pr_row_ym(int x) { return x == 7 ? 86 : 87 + 8 * x; }  // tile row (x, -1)
__host__ __device__ constexpr int pr_row_yp(int x) { return 80 + 8 * x; }                 // tile row (x, 8)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

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
// LDS row numbers of lane l's halo-row duty, packed 8 bits each into two dwords (one 8-byte load per lane):
//   lo = RA | RB<<8 | H_xm<<16 | H_xp<<24,   hi = H_ym | H_yp<<8 | hpar<<16
struct PairLaneTab {
	unsigned t[64][2];
};
constexpr PairLaneTab make_pair_lane_tab() {
	PairLaneTab T{};
	for (int l = 0; l < 64; ++l) {
		const int h = l & 31, f = h >> 3, i = h & 7;
		const int RA = f == 0 ? i : (f == 1 ? 72 + i : (f == 2 ? pr_row_ym(i) : pr_row_yp(i)));
		const int RB = 89 + 8 * (h / 6) + (h % 6);
		const int H_xm = f == 0 ? RB : (f == 1 ? 64 + i : (i == 0 ? (f == 2 ? 81 : 82) : (f == 2 ? pr_row_ym(i - 1) : pr_row_yp(i - 1))));
		const int H_xp = f == 1 ? RB : (f == 0 ? 8 + i : (i == 7 ? (f == 2 ? 83 : 84) : (f == 2 ? pr_row_ym(i + 1) : pr_row_yp(i + 1))));
		const int H_ym = f == 2 ? RB : (f == 3 ? 8 * (i + 1) + 7 : (i == 0 ? (f == 0 ? 81 : 83) : (f == 0 ? i - 1 : 72 + i - 1)));
		const int H_yp = f == 3 ? RB : (f == 2 ? 8 * (i + 1) : (i == 7 ? (f == 0 ? 82 : 84) : (f == 0 ? i + 1 : 72 + i + 1)));
		const int hpar = (i + ((f & 1) ? 0 : 1)) & 1;  // parity of the halo row's x+y: faces -x,-y sit at coordinate -1
		T.t[l][0] = (unsigned)RA | ((unsigned)RB << 8) | ((unsigned)H_xm << 16) | ((unsigned)H_xp << 24);
		T.t[l][1] = (unsigned)H_ym | ((unsigned)H_yp << 8) | ((unsigned)hpar << 16);
	}
	return T;
}
__device__ const PairLaneTab g_pair_lane_tab = make_pair_lane_tab();

// everything one wave reads from global memory for one leaf pair
struct PairIn {
	int leaf0, leaf1;  // leaf1 < 0: an unpaired leaf travelling alone (its +z neighbour, if any, belongs to another wave)
	RowP P0, P1, D0, D1;  // own rows of p and div
	float2 zlo, zhi;      // p(x,y,-2..-1) below leaf0, p(x,y,8..9) above leaf1
	float d_zh;           // div at this lane's z-halo red voxel
	bool zh_ok;           // ... and whether that voxel's leaf exists
	RowP HA, HB, HD;      // halo-row duty: adjacent row, the row behind it, div of the adjacent row
	bool f_ok;            // the face neighbour leaf exists
	float e_val;          // the halo row's z-neighbour outside the pair
	RowP ER;              // edge row (lanes 0..7)
};

// per-lane constants (do not depend on the pair)
struct PairLaneCtx {
	int l, x, y, w, I, R_xm, R_xp, R_ym, R_yp;
	bool par;
	int RA, RB, H_xm, H_xp, H_ym, H_yp;
	bool hpar;
	int ew, ea, eb;
};

__device__ __forceinline__ PairLaneCtx pair_lane_ctx(int lane) {
	PairLaneCtx c;
	c.l = lane;
	c.x = c.l >> 3, c.y = c.l & 7;
	c.par = (c.x + c.y) & 1;  // false: even z red, true: odd z red (both leaves: their z origins differ by 8)
	c.w = c.l >> 5;
	const uint2 lt = *reinterpret_cast<const uint2*>(&g_pair_lane_tab.t[c.l][0]);
	c.RA = lt.x & 255, c.RB = (lt.x >> 8) & 255, c.H_xm = (lt.x >> 16) & 255, c.H_xp = lt.x >> 24;
	c.H_ym = lt.y & 255, c.H_yp = (lt.y >> 8) & 255;
	c.hpar = (lt.y >> 16) & 1;
	c.ew = (c.l >> 2) & 1, c.ea = (c.l >> 1) & 1, c.eb = c.l & 1;
	c.I = 8 * (c.x + 1) + c.y;
	c.R_xm = c.I - 8, c.R_xp = c.I + 8, c.R_ym = c.y == 0 ? pr_row_ym(c.x) : c.I - 1, c.R_yp = c.y == 7 ? pr_row_yp(c.x) : c.I + 1;
	return c;
}

// Blocked form (k_rbgs_tile): which faces of this wave's record belong to another wave of the same workgroup (its index in
// the workgroup, -1 = nobody: the halo comes from memory as in the one-wave form). Wave-uniform.
struct TileNbr {
	int ym, yp, zm, zp;
	bool zm_single;  // the wave below carries a lone leaf: its top leaf is its tile 0
};

// issue every global load of one pair (no waits here: the values are consumed in pair_compute).
// ZERO: p_in is known to be 0 everywhere (first iteration of a solve, which is never warm-started: HNanoSolver.cu:113),
// so none of it is read; only div is. TILED: faces listed in `nb` are not loaded at all.
template <bool ZERO, bool TILED>
__device__ __forceinline__ PairIn pair_load(const PairLaneCtx& c, const int* __restrict__ rec, const float* __restrict__ div,
                                            const float* __restrict__ p_in, const TileNbr& nb) {
	PairIn in;
	// record: {leaf0, nbr27 of leaf0, leaf1, nbr27 of leaf1}; leaf1 is the +z neighbour of leaf0
	in.leaf0 = __builtin_amdgcn_readfirstlane(rec[0]);
	in.leaf1 = __builtin_amdgcn_readfirstlane(rec[28]);
	const bool single = in.leaf1 < 0;
	// below the lower leaf / above the top leaf (the top leaf is leaf0 itself when it travels alone)
	// (TILED: a face shared inside the workgroup counts as absent here -- the loads stay branch-free, hit the always-hot
	// leaf 0 and are discarded; the real values arrive through LDS in pair_compute)
	const int n_zm = (TILED && nb.zm >= 0) ? -1 : __builtin_amdgcn_readfirstlane(rec[1 + 12]);
	const int n_zp = (TILED && nb.zp >= 0) ? -1 : __builtin_amdgcn_readfirstlane(single ? rec[1 + 14] : rec[28 + 1 + 14]);
	const int l = c.l;
	const RowP zero_row = {{v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}}};
	in.D0 = glb_rowp(div, in.leaf0, l), in.D1 = glb_rowp(div, in.leaf1, l);
	if (ZERO) {
		in.P0 = in.P1 = zero_row;
		in.zlo = in.zhi = make_float2(0.0f, 0.0f);
	} else {
		in.P0 = glb_rowp(p_in, in.leaf0, l), in.P1 = glb_rowp(p_in, in.leaf1, l);
		in.zlo = *reinterpret_cast<const float2*>(p_in + (size_t)(n_zm < 0 ? 0 : n_zm) * 512 + l * 8 + 6);
		in.zhi = *reinterpret_cast<const float2*>(p_in + (size_t)(n_zp < 0 ? 0 : n_zp) * 512 + l * 8);
		if (n_zm < 0) in.zlo = make_float2(0.0f, 0.0f);
		if (n_zp < 0) in.zhi = make_float2(0.0f, 0.0f);
	}
	const int n_zh = c.par ? n_zm : n_zp;  // this lane's z-halo red voxel: below leaf0 if par, else above leaf1
	in.zh_ok = n_zh >= 0;
	in.d_zh = div[(size_t)(n_zh < 0 ? 0 : n_zh) * 512 + l * 8 + (c.par ? 7 : 0)];
	// halo-row duty: lanes 0..31 -> leaf0, 32..63 -> leaf1; 4 faces x 8 rows each. For a leaf travelling alone lanes 32..63
	// have no rows to recompute; they only fetch the z=8 neighbours of leaf0's halo rows (the face neighbour one leaf up).
	// The neighbour ids are wave-uniform: they come from scalar loads of the record and are picked per lane with selects,
	// so that no vector load sits between the record and the halo loads (one dependent memory level instead of three).
	const int f = (l >> 3) & 3, i = l & 7;
	const int* __restrict__ r0 = rec + 1;                      // neighbour table of leaf0
	const int* __restrict__ r1 = rec + (single ? 1 : 29);      // ... of the leaf whose halo rows lanes 32..63 serve
	const int a0 = r0[4], a1 = r0[22], a2 = r0[10], a3 = r0[16];    // faces -x,+x,-y,+y of leaf0
	const int b0 = r1[4], b1 = r1[22], b2 = r1[10], b3 = r1[16];
	const int c0 = r0[3], c1 = r0[21], c2 = r0[9], c3 = r0[15];     // the same faces one leaf down (dz = -1)
	const int d0 = r1[5], d1 = r1[23], d2 = r1[11], d3 = r1[17];    // ... one leaf up (dz = +1)
	const int nf0 = f == 0 ? a0 : (f == 1 ? a1 : (f == 2 ? a2 : a3));
	const int nf1 = f == 0 ? b0 : (f == 1 ? b1 : (f == 2 ? b2 : b3));
	const int ne0 = f == 0 ? c0 : (f == 1 ? c1 : (f == 2 ? c2 : c3));
	const int ne1 = f == 0 ? d0 : (f == 1 ? d1 : (f == 2 ? d2 : d3));
	// a y-face row of a neighbour inside the workgroup is that wave's own row: nothing to load or recompute (absent, as above)
	const bool shared_duty = TILED && ((f == 2 && nb.ym >= 0) || (f == 3 && nb.yp >= 0));
	const int n_f = shared_duty ? -1 : (c.w ? (single ? -1 : nf1) : nf0);
	const int n_f_p = n_f, n_f_d = n_f;
	const int srcA = f == 0 ? 56 + i : (f == 1 ? i : (f == 2 ? i * 8 + 7 : i * 8));
	const int srcB = f == 0 ? 48 + i : (f == 1 ? 8 + i : (f == 2 ? i * 8 + 6 : i * 8 + 1));
	in.f_ok = n_f >= 0;
	in.HD = glb_rowp(div, n_f_d, srcA);
	if (ZERO) {
		in.HA = in.HB = in.ER = zero_row;
		in.e_val = 0.0f;
		return in;
	}
	in.HA = glb_rowp(p_in, n_f_p, srcA);
	in.HB = glb_rowp(p_in, n_f_p, srcB);
	// the halo row's own z-neighbour outside the pair: z=-1 for the lower leaf, z=8 for the upper leaf
	const int n_e = shared_duty ? -1 : (c.w ? ne1 : ne0);
	const float ev = p_in[(size_t)(n_e < 0 ? 0 : n_e) * 512 + srcA * 8 + (c.w ? 0 : 7)];
	in.e_val = n_e < 0 ? 0.0f : ev;
	// edge rows along z (lanes 0..7): tile rows (-1,-1), (-1,8), (8,-1), (8,8) of each leaf
	const int n_er = (single && c.ew) ? -1 : rec[28 * c.ew + 1 + (c.ea ? 2 : 0) * 9 + (c.eb ? 2 : 0) * 3 + 1];
	if (l < 8) in.ER = glb_rowp(p_in, n_er, (c.ea ? 0 : 7) * 8 + (c.eb ? 0 : 7));
	return in;
}

// stage, red sweep, black sweep, store: one full iteration for one pair from registers `in`.
// TILED (k_rbgs_tile): `tiles` holds one tile per wave of the workgroup, S = tiles[own]. A face listed in `nb` is read out
// of the neighbouring wave's tile -- its staged rows before the red sweep, its updated rows after it -- instead of being
// loaded and recomputed here: the same values (a wave's recomputed halo reds ARE the neighbour's own reds), so the same bits.
// Out: where the swept rows go -- StorePlain (16-byte stores to p_out), or StoreMirror for a multi-GPU rank (below).
struct StorePlain {
	float* __restrict__ p_out;
	__device__ __forceinline__ void operator()(int leaf, int k, int l, const RowP& o) const {
		(void)k;
		float4* q = reinterpret_cast<float4*>(p_out + (size_t)leaf * 512 + l * 8);
		q[0] = make_float4(o.q[0].x, o.q[0].y, o.q[1].x, o.q[1].y);
		q[1] = make_float4(o.q[2].x, o.q[2].y, o.q[3].x, o.q[3].y);
	}
};

template <bool TILED, class Out>
__device__ __forceinline__ void pair_compute(PairTile* tiles, PairTile& S, const PairLaneCtx& c, PairIn in, const TileNbr& nb, const Out& out,
                                             const float dx2, const float omega) {
	const int l = c.l, w = c.w, I = c.I;
	const bool par = c.par;
	const bool single = in.leaf1 < 0;  // wave-uniform
	// lateral -y / +y rows of the own rows: the neighbouring wave's rows (x,7) / (x,0) where that face is shared
	const bool red_ym = TILED && c.y == 0 && nb.ym >= 0, red_yp = TILED && c.y == 7 && nb.yp >= 0;
	const PairTile& Tym = red_ym ? tiles[nb.ym] : S;
	const PairTile& Typ = red_yp ? tiles[nb.yp] : S;
	const int R_ym = red_ym ? I + 7 : c.R_ym, R_yp = red_yp ? I - 7 : c.R_yp;
	// ---- stage ----
	pt_put(S, 0, I, in.P0);
	pt_put(S, 1, I, in.P1);
	// p just below / above the pair, per row: from memory, or written here by the wave below / above when that face is shared
	if (!TILED || nb.zm < 0) S.ZM[I] = in.zlo.y;
	if (!TILED || nb.zp < 0) S.ZP[I] = in.zhi.x;
	if (TILED && nb.zp >= 0) tiles[nb.zp].ZM[I] = single ? in.P0.q[3].y : in.P1.q[3].y;
	if (TILED && nb.zm >= 0) tiles[nb.zm].ZP[I] = in.P0.q[0].x;
	pt_put(S, w, c.RA, in.HA);
	pt_put(S, w, c.RB, in.HB);
	(w ? S.ZP : S.ZM)[c.RA] = in.e_val;
	if (l < 8) pt_put(S, c.ew, 81 + c.ea * 2 + c.eb, in.ER);
	__syncthreads();
	if (TILED) {  // p just outside the pair along z, from the waves below / above
		if (nb.zm >= 0) {
			const float4 t = tiles[nb.zm].HI[nb.zm_single ? 0 : 1][I];
			in.zlo = make_float2(t.z, t.w);
		}
		if (nb.zp >= 0) {
			const float4 t = tiles[nb.zp].LO[0][I];
			in.zhi = make_float2(t.x, t.y);
		}
	}

	// ---- phase R ----
	RowP hnew, c0, c1;
	float zc;
	{
		const RowP hxm = pt_row(S, w, c.H_xm), hxp = pt_row(S, w, c.H_xp), hym = pt_row(S, w, c.H_ym), hyp = pt_row(S, w, c.H_yp);
		const float other_lo = S.HI[0][c.RA].w, other_hi = single ? S.ZP[c.RA] : S.LO[1][c.RA].x;
		const float below = w ? other_lo : in.e_val;  // z=-1 of this halo row
		const float above = w ? in.e_val : other_hi;  // z=8
		hnew = row_sweep(hxp, hxm, hyp, hym, in.HA, below, above, in.HD, dx2, omega, !c.hpar, in.f_ok);
	}
	{
		const RowP xm = pt_row(S, 0, c.R_xm), xp = pt_row(S, 0, c.R_xp), ym = pt_row(Tym, 0, R_ym), yp = pt_row(Typ, 0, R_yp);
		c0 = row_sweep(xp, xm, yp, ym, in.P0, in.zlo.y, single ? in.zhi.x : in.P1.q[0].x, in.D0, dx2, omega, !par, true);
	}
	{
		const RowP xm = pt_row(S, 1, c.R_xm), xp = pt_row(S, 1, c.R_xp), ym = pt_row(Tym, 1, R_ym), yp = pt_row(Typ, 1, R_yp);
		c1 = row_sweep(xp, xm, yp, ym, in.P1, in.P0.q[3].y, in.zhi.x, in.D1, dx2, omega, !par, !single);
	}
	{
		// z-halo red voxel: (x,y,-1) under leaf0 when par, else (x,y,8) over leaf1
		const float* ZA = par ? S.ZM : S.ZP;
		const float zyp = (par ? Typ.ZM : Typ.ZP)[R_yp], zym = (par ? Tym.ZM : Tym.ZP)[R_ym];  // (the neighbouring wave's row across a shared y face)
		zc = sor_update(ZA[c.R_xp], ZA[c.R_xm], zyp, zym, par ? in.P0.q[0].x : in.zhi.y, par ? in.zlo.x : (single ? in.P0.q[3].y : in.P1.q[3].y), in.d_zh,
		                par ? in.zlo.y : in.zhi.x, dx2, omega);
	}
	// values just outside each row after the red sweep
	float below0 = (par && in.zh_ok) ? zc : in.zlo.y;
	float above1 = (!par && in.zh_ok) ? zc : in.zhi.x;
	__syncthreads();  // phase-R reads complete before the rows are overwritten
	pt_put(S, 0, I, c0);
	pt_put(S, 1, I, c1);
	pt_put(S, w, c.RA, hnew);
	__syncthreads();
	if (TILED) {  // ... which inside the workgroup are the neighbouring waves' freshly swept rows
		if (nb.zm >= 0) below0 = tiles[nb.zm].HI[nb.zm_single ? 0 : 1][I].w;
		if (nb.zp >= 0) above1 = tiles[nb.zp].LO[0][I].x;
	}

	// ---- phase B ----
	{
		const RowP xm = pt_row(S, 0, c.R_xm), xp = pt_row(S, 0, c.R_xp), ym = pt_row(Tym, 0, R_ym), yp = pt_row(Typ, 0, R_yp);
		const RowP o = row_sweep(xp, xm, yp, ym, c0, below0, single ? above1 : c1.q[0].x, in.D0, dx2, omega, par, true);
		out(in.leaf0, 0, l, o);
	}
	if (!single) {
		const RowP xm = pt_row(S, 1, c.R_xm), xp = pt_row(S, 1, c.R_xp), ym = pt_row(Tym, 1, R_ym), yp = pt_row(Typ, 1, R_yp);
		const RowP o = row_sweep(xp, xm, yp, ym, c1, c0.q[3].y, above1, in.D1, dx2, omega, par, true);
		out(in.leaf1, 1, l, o);
	}
}

// `last`: index of the last record when this sweep walks the list backwards (odd sweeps of a solve), else -1. Beyond the
// Infinity Cache a sweep ends with the tail of the arrays cached; the next one starts there.
// `list` (optional): the records to sweep, by index (the records the blocked kernel leaves over); null = all, in order.
template <bool ZERO>
__global__ __launch_bounds__(64) void k_rbgs_pair(const int* __restrict__ pairs, const int* __restrict__ list, const float* __restrict__ div,
                                                  const float* __restrict__ p_in, float* __restrict__ p_out, const float dx2, const float omega, const int last) {
	__shared__ __attribute__((aligned(16))) PairTile S;
	const PairLaneCtx c = pair_lane_ctx(threadIdx.x);
	// backwards = rows of eight records in reverse order, the position inside a row kept: workgroup b still lands on XCD
	// b % 8, whose L2 holds that chunk's leaves from the previous sweep when the grid is small enough (128^3: all of it)
	unsigned rec = blockIdx.x;
	if (last >= 0) {
		const unsigned rows = ((unsigned)last + 1u) >> 3;
		if ((rec >> 3) < rows) rec = ((rows - 1u - (rec >> 3)) << 3) | (rec & 7u);
	}
	if (list) rec = (unsigned)list[rec];
	const TileNbr nb = {-1, -1, -1, -1, false};
	const PairIn in = pair_load<ZERO, false>(c, pairs + (size_t)rec * 56, div, p_in, nb);
	pair_compute<false>(&S, S, c, in, nb, StorePlain{p_out}, dx2, omega);
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR of one rank of a multi-GPU run, the halo exchange inside the sweep (hns_dist.hip, "mirror" pressure loop)
// ---------------------------------------------------------------------------------------------------------------
//
// The boundary leaves are first in the local leaf order and their waves run first: a boundary wave waits until every peer
// has completed the boundary part of its previous sweep (flag; raised early in that sweep, so long true by now): the peer's
// boundary rows of that sweep are then in this rank's ghost voxels, and the peer no longer reads the ghost voxels of the
// buffer this sweep writes. It then sweeps like any other wave and stores its rows twice: into p_out, and the voxels a peer
// can read during ITS next sweep (reach 2) into that peer's ghost copy of the leaf, through the peer's memory mapped here,
// and waits for those stores before it ends. The rank's "sweep complete" flag goes up on every peer when its NEXT launch
// starts (first workgroup). No second stream, no pack / transfer / unpack kernels, no ghost sweeps; the arithmetic is
// k_rbgs_pair's.
struct StoreMirror {
	float* __restrict__ p_out;
	const PhaseMirror* m;
	__device__ __forceinline__ void operator()(int leaf, int k, int l, const RowP& o) const {
		(void)k;
		float4* q = reinterpret_cast<float4*>(p_out + (size_t)leaf * 512 + l * 8);
		const float4 lo = make_float4(o.q[0].x, o.q[0].y, o.q[1].x, o.q[1].y), hi = make_float4(o.q[2].x, o.q[2].y, o.q[3].x, o.q[3].y);
		q[0] = lo;
		q[1] = hi;
		chain_store_row(*m, 0, leaf, l, lo, hi);
	}
};

template <bool ZERO>
__global__ __launch_bounds__(64) void k_rbgs_pair_mirror(const int* __restrict__ pairs, const float* __restrict__ div, const float* __restrict__ p_in,
                                                         float* __restrict__ p_out, const float dx2, const float omega, const int last, const PhaseMirror m) {
	__shared__ __attribute__((aligned(16))) PairTile S;
	const PairLaneCtx c = pair_lane_ctx(threadIdx.x);
	unsigned rec = blockIdx.x;
	if (last >= 0 && rec >= m.head_records) {  // odd sweeps walk the records behind the boundary part backwards (see hns_rbgs_iterate: "alternate")
		const unsigned rows = ((unsigned)last + 1u - m.head_records) >> 3;
		unsigned t = rec - m.head_records;
		if ((t >> 3) < rows) t = ((rows - 1u - (t >> 3)) << 3) | (t & 7u);
		rec = m.head_records + t;
	}
	const int* __restrict__ r = pairs + (size_t)rec * 56;
	const int leaf0 = __builtin_amdgcn_readfirstlane(r[0]), leaf1 = __builtin_amdgcn_readfirstlane(r[28]);
	const int lowest = ((unsigned)leaf1 < (unsigned)leaf0) ? leaf1 : leaf0;  // a boundary record: either leaf below n_boundary (leaf1 may be -1)
	chain_begin(m, lowest);
	const TileNbr nb = {-1, -1, -1, -1, false};
	const PairIn in = pair_load<ZERO, false>(c, r, div, p_in, nb);
	pair_compute<false>(&S, S, c, in, nb, StoreMirror{p_out, &m}, dx2, omega);
	// (Counting the boundary waves and raising the flag from the last one -- inside the same launch, so that the peers never wait
	// -- cost 1.6 us per sweep: every boundary wave held its slot for an atomic round trip.)
	chain_end(m, lowest);
}

// out[0] = number of records that touch a boundary leaf, out[1] = 1 + index of the last of them
__global__ void k_count_boundary_records(const int* __restrict__ pairs, unsigned n_records, int n_boundary, unsigned* out) {
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_records) return;
	const int l0 = pairs[(size_t)i * 56], l1 = pairs[(size_t)i * 56 + 28];
	if (l0 < n_boundary || (unsigned)l1 < (unsigned)n_boundary) {
		atomicAdd(out, 1u);
		atomicMax(out + 1, i + 1u);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// red-black SOR, blocked form: kTileY x kTileZ wave records per workgroup, shared faces through LDS
// ---------------------------------------------------------------------------------------------------------------
//
// What bounds k_rbgs_pair once the sweep arrays leave the Infinity Cache is not the 12 B/voxel it must move but the halo: the
// y faces (64-byte pieces) and z faces (8-byte pieces at a 32-byte stride, i.e. every line of the neighbouring leaf) of
// leaves whose own wave runs elsewhere on the chip (measured with the halo sources switched off one by one,
// profiles/micro/exp: 512^3 433 us -> 377 without the y faces, 361 without the z faces, 275 without any halo). Here the
// wave records that are each other's y / z neighbours form one workgroup (groups built on the device, hns_gridbuild.hip:
// k_group_assign) and a face between two of them never touches memory: before the red sweep a wave reads the neighbour's
// staged rows out of its LDS tile, after it the neighbour's updated rows -- exactly the values it would otherwise load and
// recompute. Outer faces of the group, and all x faces (contiguous 256-byte rows, cheap), work as in the one-wave form.
template <bool ZERO>
__global__ __launch_bounds__(64 * kTileY * kTileZ) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rbgs_tile(const int* __restrict__ pairs, const int* __restrict__ groups, const float* __restrict__ div,
                                                                    const float* __restrict__ p_in, float* __restrict__ p_out, const float dx2, const float omega,
                                                                    const int last) {
	constexpr int W = kTileY * kTileZ;
	__shared__ __attribute__((aligned(16))) PairTile S[W];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // slot a * kTileZ + c of this wave in the group
	const PairLaneCtx c = pair_lane_ctx(threadIdx.x & 63);
	unsigned grp = blockIdx.x;
	if (last >= 0) {
		const unsigned rows = ((unsigned)last + 1u) >> 3;
		if ((grp >> 3) < rows) grp = ((rows - 1u - (grp >> 3)) << 3) | (grp & 7u);
	}
	const int* __restrict__ members = groups + (size_t)grp * W;
	const int* __restrict__ rec = pairs + (size_t)members[wv] * 56;
	const int a = wv / kTileZ, cz = wv % kTileZ;
	const int leaf1 = rec[28];
	const bool single = leaf1 < 0;
	// a face is shared only if the slot next door holds exactly the neighbouring leaves (the grouping goes by coordinates
	// alone: z-runs that pair up differently, or a lone leaf beside a pair, sit in the right slots without being neighbours)
	TileNbr nb = {-1, -1, -1, -1, false};
	if (a > 0) {
		const int* q = pairs + (size_t)members[wv - kTileZ] * 56;
		if (q[0] == rec[1 + 10] && (single ? q[28] < 0 : q[28] == rec[29 + 10])) nb.ym = wv - kTileZ;
	}
	if (a + 1 < kTileY) {
		const int* q = pairs + (size_t)members[wv + kTileZ] * 56;
		if (q[0] == rec[1 + 16] && (single ? q[28] < 0 : q[28] == rec[29 + 16])) nb.yp = wv + kTileZ;
	}
	if (cz > 0) {
		const int* q = pairs + (size_t)members[wv - 1] * 56;
		const int top = q[28] < 0 ? q[0] : q[28];
		if (top == rec[1 + 12]) nb.zm = wv - 1, nb.zm_single = q[28] < 0;
	}
	if (cz + 1 < kTileZ) {
		const int* q = pairs + (size_t)members[wv + 1] * 56;
		if (q[0] == (single ? rec[1 + 14] : rec[29 + 14])) nb.zp = wv + 1;
	}
	nb.ym = __builtin_amdgcn_readfirstlane(nb.ym), nb.yp = __builtin_amdgcn_readfirstlane(nb.yp);
	nb.zm = __builtin_amdgcn_readfirstlane(nb.zm), nb.zp = __builtin_amdgcn_readfirstlane(nb.zp);
	const PairIn in = pair_load<ZERO, true>(c, rec, div, p_in, nb);
	pair_compute<true>(S, S[wv], c, in, nb, StorePlain{p_out}, dx2, omega);
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
template <class M, bool COAL, class F>
__device__ __forceinline__ void divergence_row_body(const GridDev& g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx, const M& m, const F& fz) {
	constexpr bool FUSE = !std::is_same<F, NoFuse>::value;
	// ux rows (x faces), uy rows (y faces), and the hand-over buffer of the coalesced form. FUSE: all three carved out of one array, which the out-going q4 / velocity rows
	// are then staged in once the divergence has been read out of the tiles (576 float4: 64 rows x (8 + 1 pad))
	__shared__ __attribute__((aligned(16))) RowTile TXs, TYs;
	__shared__ __attribute__((aligned(16))) float4 s_owns[COAL ? 384 : 1];
	__shared__ __attribute__((aligned(16))) float4 s_mem[FUSE ? (COAL ? 784 : 576) : 1];
	RowTile& TX = FUSE ? *reinterpret_cast<RowTile*>(s_mem) : TXs;
	RowTile& TY = FUSE ? *reinterpret_cast<RowTile*>(s_mem + 200) : TYs;
	float4* const s_own = FUSE ? s_mem + 400 : s_owns;
	const int l = threadIdx.x, x = l >> 3, y = l & 7;
	const int* __restrict__ rec = g.blk + (size_t)launch_pos(g, blockIdx.x) * 28;
	const int leaf = __builtin_amdgcn_readfirstlane(rec[0]);
	chain_begin(m, leaf);
	const int n_zm = __builtin_amdgcn_readfirstlane(rec[1 + 12]), n_zp = __builtin_amdgcn_readfirstlane(rec[1 + 14]);
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
	const float uz_m = n_zm < 0 ? 0.0f : u[((size_t)n_zm * 512 + l * 8 + 7) * 3 + 2];
	const float uz_p = n_zp < 0 ? 0.0f : u[((size_t)n_zp * 512 + l * 8) * 3 + 2];
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
			q4[i] = s_mem[(i >> 3) * 9 + (i & 7)];
		}
		wave_sync();
#pragma unroll
		for (int k = 0; k < 6; ++k) s_mem[l * 7 + k] = make_float4(ro[4 * k], ro[4 * k + 1], ro[4 * k + 2], ro[4 * k + 3]);
		wave_sync();
		float4* uo = reinterpret_cast<float4*>(fz.u_out + (size_t)leaf * 1536);
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int i = k * 64 + l, row = __mul24(i, 10923) >> 16;  // i / 6 for i < 384
			uo[i] = s_mem[row * 7 + (i - row * 6)];
		}
	}
	float4* q = reinterpret_cast<float4*>(div + (size_t)leaf * 512 + l * 8);
	q[0] = make_float4(d[0], d[1], d[2], d[3]);
	q[1] = make_float4(d[4], d[5], d[6], d[7]);
	chain_store_row(m, 0, leaf, l, make_float4(d[0], d[1], d[2], d[3]), make_float4(d[4], d[5], d[6], d[7]));
	chain_end(m, leaf);
}
template <class M, bool COAL = false>
__global__ __launch_bounds__(64) void k_divergence_row(const GridDev g, const float* __restrict__ u, float* __restrict__ div, const float inv_dx, const M m) {
	divergence_row_body<M, COAL, NoFuse>(g, u, div, inv_dx, m, NoFuse{});
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
		// (256^3: 74 -> 66 us). Option "rev" = 0 walks every kernel forwards.
		gd.rev = options().rev.load();
		// option "divergence" = auto | row | coalesced: by size (k_divergence_row's COAL form from 16k leaves; loses below, see the kernel)
		const int form = options().divergence_form.load();
		if (form == 2 || (form == 0 && g->n_active >= 16384))
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
	gd.rev = options().rev.load();
	const CombustFuse fz{fuel, waste, temperature, flame, reinterpret_cast<float4*>(q4), vel3_out, temp_gain, expansion, dt, ambient, strength};
	const int form = options().divergence_form.load();
	if (form == 2 || (form == 0 && g->n_active >= 16384))
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
	gd.rev = options().rev.load();  // (as hns_dev_divergence; walked backwards the boundary leaves come last, which the chain does not mind)
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

// Which form sweeps this grid (option "rbgs"): the pair kernel, unless the option or the shape of the grid says otherwise.
// One leaf per wave is the better choice for SMALL IRREGULAR grids, which are latency-bound (twice as many, shorter waves;
// no half-empty pair waves): measured on the 3.9k-leaf plume 9.0 vs 11.9 us per iteration; and for grids of a few hundred
// leaves, which cannot fill 256 CUs with one wave per pair.
static int rbgs_form(hns_grid* g, int opt) {
	if (opt == kRbgsColor || opt == kRbgsWave) return opt;
	if (!g->d_pairs) return kRbgsWave;
	if (opt == kRbgsPair) return kRbgsPair;
	if (opt == kRbgsTile) return (hns_grid_build_tiles(g) == HNS_OK && g->d_tile_groups) ? kRbgsTile : kRbgsPair;
	// (<= 2048: also the boundary range of a multi-GPU rank, a few thousand leaves swept next to the interior launch)
	if (g->n_active <= 2048 || (g->n_active <= 16384 && g->n_singles * 20 > g->n_pairs)) return kRbgsWave;
	// far beyond the Infinity Cache the halo of leaves swept elsewhere on the chip is what costs: blocked form where most
	// records sit in complete groups (512^3: 428 -> 379 us per sweep; no gain on the ragged 66k-leaf plume, none in cache)
	if (g->n_active > 100000 && hns_grid_build_tiles(g) == HNS_OK && g->d_tile_groups && g->n_tile_groups * (uint64_t)(kTileY * kTileZ) * 10 >= g->n_pairs * 9)
		return kRbgsTile;
	return kRbgsPair;
}

// Grids the temporally blocked form sweeps by default (option "rbgs" = auto)
// (measured, profiles/r03_sor_forms.txt, us per iteration one-iteration form -> blocked: 64 leaves 3.7 -> 2.1, 512 4.0 -> 3.0, 4,096 7.9 -> 6.5,
// 13,824 19.0 -> 17.8, 32,768 39.2 -> 37.1, 64,000 98 -> 88, 262,144 394 -> 349: every size; the one-iteration forms remain for
// launch ranges -- the ranks of a multi-GPU run -- and for an odd iteration left over)
static bool rbgs_auto_block(const hns_grid* g) {
	(void)g;
	return true;
}

// one full (red, black) iteration src -> dst. src_is_zero (pair form only): the caller vouches that src is 0 on every
// leaf (first iteration of a solve) and the kernel skips reading it.
static int launch_rbgs_iteration(hns_grid* g, const GridDev& gd, const float* div, const float* src, float* dst, float dx2, float omega, int form,
                                 hipStream_t st, bool src_is_zero = false, bool backwards = false) {
	const int last = backwards ? (int)g->n_pairs - 1 : -1;
	if (form == kRbgsColor) {
		// the reference's own decomposition, kept as the independent cross-check: copy, then one launch per colour in place
		HNS_HIP(hipMemcpyAsync(dst, src, sizeof(float) * 512 * (size_t)g->topo.n_leaves, hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, st, gd, div, dst, dx2, omega, 0);
		hipLaunchKernelGGL(k_rbgs_color, dim3((unsigned)g->n_active), dim3(256), 0, st, gd, div, dst, dx2, omega, 1);
	} else if (form == kRbgsWave) {
		hipLaunchKernelGGL(k_rbgs_wave<NoMirror>, dim3((unsigned)g->n_active), dim3(64), 0, st, gd, div, src, dst, dx2, omega, NoMirror{});
	} else if (form == kRbgsTile) {
		// complete groups through the blocked kernel, the records outside them through the one-wave kernel (disjoint leaves, both
		// read src and write dst: any order)
		const dim3 tb(64 * kTileY * kTileZ);
		const int* recs = (const int*)g->d_pairs;
		const int glast = backwards ? (int)g->n_tile_groups - 1 : -1, rlast = backwards ? (int)g->n_tile_rest - 1 : -1;
		if (src_is_zero) {
			if (g->n_tile_groups) hipLaunchKernelGGL(k_rbgs_tile<true>, dim3((unsigned)g->n_tile_groups), tb, 0, st, recs, (const int*)g->d_tile_groups, div, src, dst, dx2, omega, glast);
			if (g->n_tile_rest) hipLaunchKernelGGL(k_rbgs_pair<true>, dim3((unsigned)g->n_tile_rest), dim3(64), 0, st, recs, (const int*)g->d_tile_rest, div, src, dst, dx2, omega, rlast);
		} else {
			if (g->n_tile_groups) hipLaunchKernelGGL(k_rbgs_tile<false>, dim3((unsigned)g->n_tile_groups), tb, 0, st, recs, (const int*)g->d_tile_groups, div, src, dst, dx2, omega, glast);
			if (g->n_tile_rest) hipLaunchKernelGGL(k_rbgs_pair<false>, dim3((unsigned)g->n_tile_rest), dim3(64), 0, st, recs, (const int*)g->d_tile_rest, div, src, dst, dx2, omega, rlast);
		}
	} else if (src_is_zero) {
		hipLaunchKernelGGL(k_rbgs_pair<true>, dim3((unsigned)g->n_pairs), dim3(64), 0, st, (const int*)g->d_pairs, (const int*)nullptr, div, src, dst, dx2, omega, last);
	} else {
		// one launch: the record list holds the z-adjacent pairs and, as {leaf, nbr27, -1, ...}, the leaves that found no partner
		hipLaunchKernelGGL(k_rbgs_pair<false>, dim3((unsigned)g->n_pairs), dim3(64), 0, st, (const int*)g->d_pairs, (const int*)nullptr, div, src, dst, dx2, omega, last);
	}
	return HNS_OK;
}

static inline float dx2_of(float dx) { return dx * dx; }  // Kernel.cu:608

int hns_dev_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                         void* stream) {
	return hns_rbgs_iterate(g, div, p_a, p_b, dx, omega, iterations, result_in_b, stream, false);
}

// from_zero: the solve starts from p = 0 (the reference never warm-starts, HNanoSolver.cu:113 / PressureProjection.cu:35) and
// p_a's content is irrelevant: the first sweep does not read it, so the caller need not clear it either.
int hns_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b, void* stream,
                     bool from_zero) {
	if (int rc = check_grid(g, "hns_dev_rbgs_iterate")) return rc;
	NULLCHK(!div || !p_a || !p_b, "hns_dev_rbgs_iterate");
	if (p_a == p_b) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: p_a and p_b must be distinct buffers");
	if (iterations < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_rbgs_iterate: negative iteration count");
	if (result_in_b) *result_in_b = iterations & 1;
	if (g->n_active == 0 || iterations == 0) {
		if (from_zero) HNS_HIP(hipMemsetAsync(p_a, 0, sizeof(float) * 512 * (size_t)g->topo.n_leaves, (hipStream_t)stream));  // the result is p_a = 0
		return HNS_OK;
	}
	const float dx2 = dx * dx;  // Kernel.cu:608
	const GridDev gd = g->dev();
	// Temporally blocked form (hns_sorblock.hip): k iterations per launch, p read and written once per launch. An odd iteration
	// left over goes through the one-iteration form below.
	{
		const int opt = options().rbgs.load();
		int k_max = 0;
		const int lb = (iterations >= 2 && (opt == kRbgsBlock || (opt == kRbgsAuto && rbgs_auto_block(g)))) ? hns_rbgs_block_shape(g, &k_max) : 0;
		if (lb) {
			float* src = p_a;
			float* dst = p_b;
			int left = iterations, launches = 0;
			bool zero = from_zero;
			while (left >= 2) {
				const int k = (k_max >= 4 && left >= 4) ? 4 : 2;
				if (int rc = hns_rbgs_block_launch(g, lb, k, zero, div, src, dst, dx2, omega, stream)) return rc;
				std::swap(src, dst);
				left -= k, ++launches, zero = false;
			}
			if (left) {
				if (int rc = launch_rbgs_iteration(g, gd, div, src, dst, dx2, omega, rbgs_form(g, kRbgsAuto), (hipStream_t)stream)) return rc;
				++launches;
			}
			if (result_in_b) *result_in_b = launches & 1;
			return launch_status("hns_dev_rbgs_iterate");
		}
	}
	const int form = rbgs_form(g, options().rbgs.load());
	if (from_zero && form != kRbgsPair && form != kRbgsTile) {  // the other forms read their input: give them the zeros
		HNS_HIP(hipMemsetAsync(p_a, 0, sizeof(float) * 512 * (size_t)g->topo.n_leaves, (hipStream_t)stream));
		from_zero = false;
	}
	// Odd sweeps walk the record list backwards: a sweep ends with the tail of p / div in the Infinity Cache and the next one
	// begins there. Nothing below the cache size, -11 % at 288^3 (287 MB of sweep arrays against 256 MB of cache), -6 % at
	// 320^3, -1.4 % at 512^3 (profiles/micro/sor_schedule.py). The result does not depend on the order. Option "alternate" = 0: off.
	// (Replaying the loop as a hipGraph was measured neutral at every size -- 64^3 4.11 vs 4.14 us per sweep, 256^3 39.2 vs 39.1: the
	// eager loop is never launch-bound -- and removed in round 3.)
	const bool alternate = options().alternate.load() != 0;
	float* src = p_a;
	float* dst = p_b;
	for (int it = 0; it < iterations; ++it) {
		if (int rc = launch_rbgs_iteration(g, gd, div, src, dst, dx2, omega, form, (hipStream_t)stream, from_zero && it == 0, alternate && (it & 1))) return rc;
		float* tmp = src;
		src = dst;
		dst = tmp;
	}
	return launch_status("hns_dev_rbgs_iterate");
}

int hns_grid_rbgs_plan(hns_grid* g, int iterations, char* description, uint64_t description_bytes, int* launches, int* iterations_per_launch) {
	if (int rc = check_grid(g, "hns_grid_rbgs_plan")) return rc;
	if (iterations < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_rbgs_plan: negative iteration count");
	const int opt = options().rbgs.load();
	int k_max = 0;
	const int lb = (iterations >= 2 && (opt == kRbgsBlock || (opt == kRbgsAuto && rbgs_auto_block(g)))) ? hns_rbgs_block_shape(g, &k_max) : 0;
	char buf[256];
	int n = iterations, k = 1;
	if (lb) {
		n = 0;
		int left = iterations;
		while (left >= 2) left -= (k_max >= 4 && left >= 4) ? 4 : 2, ++n;
		n += left;
		k = k_max;
		const bool lean = hns_rbgs_block_lean(g, lb, k_max), xy = lean && (options().sor_block_lean.load() & ~4) == 0;  // (auto | xy)
		snprintf(buf, sizeof(buf), "k_rbgs_block%s<%d,%d>: %d red+black iterations per launch on %s blocks with a %d-voxel halo, p read and written once per launch%s", xy ? "_xy" : "", lb, k_max, k_max,
		         lb == 1 ? "one-leaf (8^3-voxel)" : "16^3-voxel", 2 * k_max,
		         !lean ? "" : (xy ? " (XY form: rows in LDS, three workgroups per CU, the sweep threads fetch their own rows)" : " (lean form: rows in LDS, three workgroups per CU, waves sorted by parity)"));
	} else {
		const int form = rbgs_form(g, opt == kRbgsBlock ? kRbgsAuto : opt);
		const char* names[] = {"?", "k_rbgs_color: two launches per iteration, in place (the reference's decomposition)", "k_rbgs_wave: one launch = one red+black iteration, one wave per leaf",
		                       "k_rbgs_pair: one launch = one red+black iteration, one wave per z-adjacent leaf pair", "k_rbgs_tile: one launch = one red+black iteration, 2 x 2 wave records per workgroup"};
		snprintf(buf, sizeof(buf), "%s", names[form >= 0 && form <= 4 ? form : 0]);
		if (form == kRbgsColor) n = 2 * iterations;
	}
	if (description && description_bytes) snprintf(description, description_bytes, "%s", buf);
	if (launches) *launches = n;
	if (iterations_per_launch) *iterations_per_launch = k;
	return HNS_OK;
}

// ---- the mirroring sweep of a multi-GPU rank (see k_rbgs_pair_mirror) ----
int hns_rbgs_count_boundary_records(hns_grid* g, int n_boundary, unsigned* d_scratch, unsigned* out2, void* stream) {
	if (int rc = check_grid(g, "hns_rbgs_count_boundary_records")) return rc;
	if (!g->d_pairs) return fail(HNS_ERR_RUNTIME, "hns_rbgs_count_boundary_records: the grid has no wave records");
	hipStream_t st = (hipStream_t)stream;
	HNS_HIP(hipMemsetAsync(d_scratch, 0, 2 * sizeof(unsigned), st));
	if (g->n_pairs) hipLaunchKernelGGL(k_count_boundary_records, dim3((unsigned)((g->n_pairs + 255) / 256)), dim3(256), 0, st, (const int*)g->d_pairs, (unsigned)g->n_pairs, n_boundary, d_scratch);
	HNS_HIP(hipMemcpyAsync(out2, d_scratch, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
	HNS_HIP(hipStreamSynchronize(st));
	HNS_HIP(hipMemsetAsync(d_scratch, 0, 2 * sizeof(unsigned), st));
	return HNS_OK;
}

int hns_rbgs_mirror_sweep(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero, const hns::PhaseMirror* m, void* stream, bool backwards) {
	if (int rc = check_grid(g, "hns_rbgs_mirror_sweep")) return rc;
	if (!g->d_pairs || g->n_pairs == 0) return fail(HNS_ERR_RUNTIME, "hns_rbgs_mirror_sweep: the grid has no wave records");
	// small ragged ranks: one leaf per wave, as on a single GPU (rbgs_form; the 66k-leaf plume in 8 ranges: 22 -> 17 us per sweep)
	const int opt = options().rbgs.load();
	if (opt == kRbgsWave || (opt == kRbgsAuto && g->d_blk && (g->n_active <= 2048 || (g->n_active <= 16384 && g->n_singles * 20 > g->n_pairs)))) {
		if (src_is_zero) HNS_HIP(hipMemsetAsync(const_cast<float*>(src), 0, sizeof(float) * 512 * (size_t)g->topo.n_leaves, (hipStream_t)stream));  // this form reads its input
		hipLaunchKernelGGL(k_rbgs_wave<PhaseMirror>, dim3((unsigned)g->n_active), dim3(64), 0, (hipStream_t)stream, g->dev(), div, src, dst, dx2_of(dx), omega, *m);
		return launch_status("hns_rbgs_mirror_sweep");
	}
	// The boundary leaves come first in the local leaf order and their waves always run at the START of the launch, next to
	// everything else (walked backwards they were a tail of slow waves: 69 us per sweep instead of 38), so a rank's "sweep
	// complete" flag goes up long before its sweep ends and the peers' next sweep never waits for it. `backwards` reverses
	// only the records behind them (what "alternate" does for the single-GPU loop: the sweep starts where the last one ended).
	const int last = backwards ? (int)g->n_pairs - 1 : -1;
	if (src_is_zero)
		hipLaunchKernelGGL(k_rbgs_pair_mirror<true>, dim3((unsigned)g->n_pairs), dim3(64), 0, (hipStream_t)stream, (const int*)g->d_pairs, div, src, dst, dx2_of(dx), omega, last, *m);
	else
		hipLaunchKernelGGL(k_rbgs_pair_mirror<false>, dim3((unsigned)g->n_pairs), dim3(64), 0, (hipStream_t)stream, (const int*)g->d_pairs, div, src, dst, dx2_of(dx), omega, last, *m);
	return launch_status("hns_rbgs_mirror_sweep");
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
