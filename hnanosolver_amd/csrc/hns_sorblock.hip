// hns_sorblock.hip -- red-black SOR, temporally blocked: K whole (red, black) iterations per launch (gfx950 / CDNA4).
//
// Reference: redBlackGaussSeidelUpdate(_opt), src/Cuda/Kernel.cu:521-623, driven 2 launches per iteration by
// src/Cuda/HNanoSolver.cu:256-272 / src/Cuda/PressureProjection.cu:51-60. The arithmetic per voxel is exactly
// Kernel.cu:621-622 (same association, -ffp-contract=off), and every update reads exactly the values the reference's
// launch sequence would have produced, so the result is bit-identical to k_rbgs_color (the two-launch form).
//
// Why: the one-launch-per-iteration kernels (hns_pressure.hip) move the compulsory 12 B/voxel/iteration, which bounds
// them at the fabric rate on large grids (256^3: 38 us per iteration), and pay one kernel boundary plus one wave's load
// latency per iteration on small ones (128^3: 8 us per iteration for 1.5 us worth of traffic). A colour sweep moves
// information one voxel, so K iterations of a block of voxels depend on the block plus a halo of H = 2K voxels only:
// a workgroup loads that tile ONCE, runs all 2K colour sweeps in LDS / registers, and stores the block -- p is read and
// written once per K iterations, div is read once, there is one kernel boundary per K iterations.
//
// Shape: a block is LB^3 leaves (8*LB voxels on a side), the tile T = 8*LB + 2H voxels on a side. One THREAD per z-row
// of the tile (T*T rows, the waves of a workgroup sorted by the parity of x+y); a thread keeps its row of p and of div*dx^2 in registers for the whole launch, split by
// colour: R[] = its red voxels, B[] = its black ones (which z they are depends on the parity of x+y). The four lateral
// neighbours of a voxel have the other colour, at the same index of their rows' arrays: a sweep of one colour reads the
// four neighbouring rows' OTHER-colour arrays from LDS (ds_read_b128, 48-byte row stride: conflict-free for consecutive
// rows), takes the z neighbours from its own registers, and writes its updated array back for the next sweep. One
// workgroup barrier per sweep. After sweep s the values within s voxels of the tile's rim are stale (they lacked a
// neighbour); after 2K sweeps exactly the halo is, and the block inside it is what the reference computes.
// Voxels of absent leaves load as 0 (hardware bounds check of a buffer descriptor), are never updated, and their stores
// are dropped by the same check: "outside the domain p = 0" (Stencils.hpp:83) without a branch.
//
// Two kernels. The one described above (k_rbgs_block<1, K>: sb_body / sb_sweep, rows in registers) serves one-leaf blocks -- grids of up to
// 600 leaves, which cannot fill the chip with larger ones. 16^3 blocks are swept by k_rbgs_block_xy (further down): the row state stays
// in LDS, where the neighbours need it anyway, the thread keeps only div * dx^2; three workgroups fit a CU; the thread that sweeps a
// row fetches it; the block is stored in memory order; and only the part of the tile that can reach the block in 2K sweeps is fetched
// and computed. How it got there (the 16^3 rows-in-registers form, the parity-sorted "lean" form and its LDS-DMA variant, each with its
// measurements): DESIGN_HISTORY.md, profiles/r03_sorblock_notes.txt 10-12, profiles/r05_sorblock_notes.txt.
#include "hns_device.hpp"
#include "hns_flags.hpp"

namespace hns {

typedef float sb4f __attribute__((ext_vector_type(4)));
typedef int sb4i __attribute__((ext_vector_type(4)));
__device__ sb4f sb_load4(sb4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void sb_store4(sb4f data, sb4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ sb4i sb_rsrc(const float* p, unsigned bytes) {
	const unsigned long long a = (unsigned long long)p;
	sb4i r;
	r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
	r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0: raw buffer
	r.z = __builtin_amdgcn_readfirstlane((int)bytes);                            // bytes covered: offsets at or past it read 0 / are not stored
	r.w = 0x00020000;
	return r;
}

template <int LB, int K>
struct SbGeo {
	static constexpr int H = 2 * K;              // halo depth, voxels
	static constexpr int T = 8 * LB + 2 * H;     // tile edge, voxels
	static constexpr int C = LB + 2;             // leaf cells per axis under the tile
	static constexpr int HALF = T / 2;           // voxels of one colour in a row = rows of one parity per x
	static constexpr int ROWS = T * T / 2;       // rows of one parity (LDS row numbers: x * HALF + (y >> 1))
	static constexpr int TC = T - 2;             // rows per axis that are ever updated (the rim rows are only read)
	static constexpr int HC = TC / 2;            // ... of one parity per x
	static constexpr int CROWS = TC * HC;        // ... of one parity: one thread each
	static constexpr int RIM = 2 * TC;           // rim rows of one parity (without the tile's four corner rows, which nothing reads)
	static constexpr int SEC = (CROWS + 63) / 64 * 64;  // threads of one parity section (whole waves)
	static constexpr int NT = 2 * SEC;           // threads
	static constexpr int NQ = (HALF + 3) / 4;    // 16-byte pieces of a colour array
	static constexpr int HS4 = 3;                // LDS stride of a colour array in float4 (48 bytes)
	static constexpr int NCH = T / 4;            // 16-byte pieces of a row in memory
	static constexpr int REC = LB == 1 ? 28 : 64;  // ints per block record
	// 24-voxel tiles: the row state (p of both colours) stays in LDS, where the neighbours need it anyway, and only div * dx^2 is kept
	// in registers -- 24 instead of 48 registers of state, <= 80 in all, which with the overlapped LDS arrays below lets THREE workgroups
	// onto a CU instead of two (16-voxel tiles are small-grid, latency-bound launches: they keep the rows in registers)
	static_assert(H % 4 == 0 && H <= 8, "rows must start on a 16-byte piece and stay within the neighbouring leaf");
	static_assert(NQ <= HS4 && NT <= 1024 && RIM <= CROWS, "tile too large");
};

constexpr float kInv6 = 0.166666667f;  // Kernel.cu:609

// a thread's z-row for the length of the launch, split by colour (static indexing only: the arrays live in registers)
template <int HALF, int C>
struct SbRow {
	float R[HALF], B[HALF];    // p of the red / black voxels of the row
	float dR[HALF], dB[HALF];  // div * dx^2 of the same voxels
	unsigned ok[C];            // per leaf cell along z: all ones if the leaf exists, else 0
};

// LDS: [parity of x+y][colour][row][HS4 float4]. Rows of one parity are numbered x * HALF + (y >> 1): the lanes of a wave hold
// consecutive rows of ONE parity, so which z are red is the same for the whole wave (PAR is a template parameter: no
// per-lane selects anywhere) and ds_read_b128 at the 48-byte row stride is conflict-free.
// The red values of the rows x = 0 and x = T-1 are never read (a black update happens at distance >= 2 from the rim, its lateral
// neighbours at >= 1), so a red array is kept without its first and last HALF rows and the arrays overlap there: [black 0][red 0]
// [black 1][red 1], a red array's row 0 lying HALF rows inside the end of the black array in front of it. Nobody may WRITE red values
// of x = 0 / x = T-1 (the rim duty skips them). 52,992 instead of 55,296 bytes for 24-voxel tiles: three workgroups fit 160 KB.
template <int LB, int K>
struct SbLds {
	using G = SbGeo<LB, K>;
	static constexpr int NB = G::ROWS * G::HS4, NR = (G::ROWS - 2 * G::HALF) * G::HS4;
	float4 a[2 * (NB + NR)];
	__device__ __forceinline__ float4* arr(int par, int colour) { return a + par * (NB + NR) + (colour ? 0 : NB - G::HALF * G::HS4); }
};

// One colour sweep S (1-based; odd = red = colour 0, Kernel.cu:601) of the whole tile, for the rows with parity PAR of x+y.
// i = LDS number of the row among the rows of its parity, x * HALF + (y >> 1); b = y & 1.
template <int LB, int K, int S, int NS, bool PAR, bool MASKED, class Row>
__device__ __forceinline__ void sb_sweep(Row& r, SbLds<LB, K>& L, const int i, const int b, const int dist, const float omega) {
	using G = SbGeo<LB, K>;
	constexpr int H = G::H, HALF = G::HALF, HS4 = G::HS4;
	constexpr bool red = (S & 1) != 0;
	float(&X)[HALF] = red ? r.R : r.B;          // the colour being updated
	const float(&Y)[HALF] = red ? r.B : r.R;    // its neighbours' colour
	const float(&dX)[HALF] = red ? r.dR : r.dB;
	const float4* LY = L.arr(PAR ? 0 : 1, red ? 1 : 0);  // the lateral neighbours are rows of the other parity
	float4* LX = L.arr(PAR ? 1 : 0, red ? 0 : 1);
	// z neighbours of X[j] in the own row: red voxel j of a row with even x+y sits at z = 2j (between black j-1 and j), with odd
	// x+y at z = 2j+1 (between black j and j+1); black voxels the other way round
	constexpr bool up = red ? PAR : !PAR;
	// voxels further than S from the rim along z (a superset by one: the extra candidate is stale either way)
	// (in whole 16-byte pieces only: partial pieces make the compiler split the LDS accesses into narrower, conflicting ones)
	constexpr int qlo = (S / 2) / 4, qhi = (HALF - S / 2 + 3) / 4;
	constexpr int jlo = 4 * qlo, jhi = 4 * qhi < HALF ? 4 * qhi : HALF;
	if (dist >= S) {
		// rows (x+1, y), (x-1, y): same number +- HALF; (x, y+1): number + b; (x, y-1): number + b - 1
		const float4* pxp = LY + (i + HALF) * HS4;
		const float4* pxm = LY + (i - HALF) * HS4;
		const float4* pyp = LY + (i + b) * HS4;
		const float4* pym = LY + (i + b - 1) * HS4;
		// (software-pipelined: the four ds_read_b128 of the next 16-byte piece are in flight while this piece's candidates are
		// computed; the scheduling barriers keep the compiler from hoisting ALL of a sweep's reads, which costs 32 more registers)
		float4 nx[4] = {pxp[qlo], pxm[qlo], pyp[qlo], pym[qlo]};
#pragma unroll
		for (int q = qlo; q < qhi; ++q) {
			const float4 xp4 = nx[0], xm4 = nx[1], yp4 = nx[2], ym4 = nx[3];
			if (q + 1 < qhi) nx[0] = pxp[q + 1], nx[1] = pxm[q + 1], nx[2] = pyp[q + 1], nx[3] = pym[q + 1];
			__builtin_amdgcn_sched_barrier(0);
			const float xp[4] = {xp4.x, xp4.y, xp4.z, xp4.w}, xm[4] = {xm4.x, xm4.y, xm4.z, xm4.w};
			const float yp[4] = {yp4.x, yp4.y, yp4.z, yp4.w}, ym[4] = {ym4.x, ym4.y, ym4.z, ym4.w};
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int j = 4 * q + e;
				if (j < jlo || j >= jhi) continue;
				const float below = j > 0 ? Y[j > 0 ? j - 1 : 0] : 0.0f, above = j + 1 < HALF ? Y[j + 1 < HALF ? j + 1 : 0] : 0.0f;
				const float zm = up ? Y[j] : below, zp = up ? above : Y[j];
				const float pGS = ((xp[e] + xm[e] + yp[e] + ym[e] + zp + zm) - dX[j]) * kInv6;  // Kernel.cu:621 (dX = div * dx^2)
				const float cand = X[j] + omega * (pGS - X[j]);                                   // Kernel.cu:622
				const int cz = (2 * j - H + 8) >> 3;  // leaf cell of this voxel along z: the same for both parities
				// a voxel of an absent leaf stays +0 (as a bit mask, not a select: the compiler turns selects here into a branch per voxel)
				X[j] = MASKED ? __uint_as_float(__float_as_uint(cand) & r.ok[cz]) : cand;
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (S < NS) {
#pragma unroll
			for (int q = qlo; q < qhi; ++q) {
				const int j = 4 * q;
				LX[i * HS4 + q] = make_float4(X[j], j + 1 < HALF ? X[j + 1 < HALF ? j + 1 : 0] : 0.0f, j + 2 < HALF ? X[j + 2 < HALF ? j + 2 : 0] : 0.0f,
				                              j + 3 < HALF ? X[j + 3 < HALF ? j + 3 : 0] : 0.0f);
			}
		}
	}
	if (S < NS) __syncthreads();
}

// NS colour sweeps: 2K = K iterations; K = 2 with NS = 2 is ONE iteration on the tile of two (an odd iteration left over: SbSweepsXY, which see)
template <int LB, int K, int S, int NS, bool PAR, bool MASKED>
struct SbSweeps {
	template <class Row>
	static __device__ __forceinline__ void run(Row& r, SbLds<LB, K>& L, const int i, const int b, const int dist, const float omega) {
		sb_sweep<LB, K, S, NS, PAR, MASKED>(r, L, i, b, dist, omega);
		if constexpr (S < NS) SbSweeps<LB, K, S + 1, NS, PAR, MASKED>::run(r, L, i, b, dist, omega);
	}
};

// everything one thread does, for a row with parity PAR of x+y (wave-uniform)
template <int LB, int K, bool ZERO, bool PAR, int NS>
__device__ __forceinline__ void sb_body(SbLds<LB, K>& L, const int t, const int* __restrict__ recs, const float* __restrict__ div, const float* __restrict__ p_in,
                                        float* __restrict__ p_out, const unsigned field_bytes, const float dx2, const float omega) {
	using G = SbGeo<LB, K>;
	constexpr int H = G::H, T = G::T, C = G::C, HALF = G::HALF, NQ = G::NQ, NCH = G::NCH, HS4 = G::HS4;
	// thread t of the section: the t-th row of this parity among the rows inside the rim, x = 1.., y = 1..
	const bool valid = t < G::CROWS;  // (the last wave of a section has lanes without a row)
	const int xq = valid ? t / G::HC : 0;
	const int x = 1 + xq, y = valid ? 1 + 2 * (t - xq * G::HC) + ((x + 1 + (PAR ? 1 : 0)) & 1) : 1;
	const int b = y & 1;
	const int i = x * HALF + (y >> 1);
	const int dist = valid ? min(min(x, T - 1 - x), min(y, T - 1 - y)) : -1;  // rows at distance >= s from the rim take part in sweep s
	const int cx = (x - H + 8) >> 3, cy = (y - H + 8) >> 3;
	const unsigned row_bytes = (unsigned)(((((x - H) & 7) << 3) | ((y - H) & 7)) * 32);  // the row inside its leaf
	const int* __restrict__ rec = recs + (size_t)blockIdx.x * G::REC + (LB == 1 ? 1 : 0) + (cx * C + cy) * C;
	unsigned base[C];  // byte offset of this row in the leaf of each cell along z; an absent leaf (-1) lands beyond the field
	SbRow<HALF, C> r;
#pragma unroll
	for (int cz = 0; cz < C; ++cz) {
		const int id = valid ? rec[cz] : -1;
		r.ok[cz] = id >= 0 ? 0xFFFFFFFFu : 0u;
		base[cz] = (unsigned)id * 2048u + row_bytes;
	}
	const sb4i rp = sb_rsrc(p_in, field_bytes), rd = sb_rsrc(div, field_bytes), ro = sb_rsrc(p_out, field_bytes);
	sb4f pc[NCH], dc[NCH];
#pragma unroll
	for (int j = 0; j < NCH; ++j) {
		const int cz = (4 * j - H + 8) >> 3, zl = (4 * j - H) & 7;
		const unsigned off = base[cz] + (unsigned)(zl * 4);
		dc[j] = sb_load4(rd, (int)off, 0, 0);
		if (ZERO) pc[j] = sb4f{0.0f, 0.0f, 0.0f, 0.0f};
		else pc[j] = sb_load4(rp, (int)off, 0, 0);
	}
	// Rim duty of the first RIM threads of the section: one rim row of this parity each -- p only, straight into LDS (nobody
	// updates a rim row). Side 0: x = 0, 1: x = T-1, 2: y = 0, 3: y = T-1; the m-th row of the right parity along the side.
	if (t < G::RIM) {
		const int side = t / G::HC, m = t - side * G::HC;
		const int along = 2 * m + (((side & 1) != 0) == PAR ? 2 : 1);  // sides 0, 2: along + 0 has parity PAR; sides 1, 3: along + T-1 (odd)
		const int rx = side == 0 ? 0 : (side == 1 ? T - 1 : along), ry = side == 2 ? 0 : (side == 3 ? T - 1 : along);
		const int rcx = (rx - H + 8) >> 3, rcy = (ry - H + 8) >> 3;
		const unsigned rrow = (unsigned)(((((rx - H) & 7) << 3) | ((ry - H) & 7)) * 32);
		const int* __restrict__ rrec = recs + (size_t)blockIdx.x * G::REC + (LB == 1 ? 1 : 0) + (rcx * C + rcy) * C;
		unsigned rbase[C];
#pragma unroll
		for (int cz = 0; cz < C; ++cz) rbase[cz] = (unsigned)rrec[cz] * 2048u + rrow;
		sb4f rc[NCH];
#pragma unroll
		for (int j = 0; j < NCH; ++j) rc[j] = ZERO ? sb4f{0.0f, 0.0f, 0.0f, 0.0f} : sb_load4(rp, (int)(rbase[(4 * j - H + 8) >> 3] + (unsigned)(((4 * j - H) & 7) * 4)), 0, 0);
		float4* LR = L.arr(PAR ? 1 : 0, 0) + (rx * HALF + (ry >> 1)) * HS4;
		float4* LK = L.arr(PAR ? 1 : 0, 1) + (rx * HALF + (ry >> 1)) * HS4;
		float rr[HALF + 4], rb[HALF + 4];
#pragma unroll
		for (int j = 0; j < NCH; ++j) {
			rr[2 * j] = PAR ? rc[j].y : rc[j].x, rb[2 * j] = PAR ? rc[j].x : rc[j].y;
			rr[2 * j + 1] = PAR ? rc[j].w : rc[j].z, rb[2 * j + 1] = PAR ? rc[j].z : rc[j].w;
		}
#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			if (side >= 2) LR[q] = make_float4(rr[4 * q], rr[4 * q + 1], 4 * q + 2 < HALF ? rr[4 * q + 2] : 0.0f, 4 * q + 3 < HALF ? rr[4 * q + 3] : 0.0f);
			LK[q] = make_float4(rb[4 * q], rb[4 * q + 1], 4 * q + 2 < HALF ? rb[4 * q + 2] : 0.0f, 4 * q + 3 < HALF ? rb[4 * q + 3] : 0.0f);
		}
	}
	// split by colour: even z of a row with even x+y are red (colour = (x + y + z) & 1, Kernel.cu:599-601)
#pragma unroll
	for (int j = 0; j < NCH; ++j) {
		r.R[2 * j] = PAR ? pc[j].y : pc[j].x, r.B[2 * j] = PAR ? pc[j].x : pc[j].y;
		r.R[2 * j + 1] = PAR ? pc[j].w : pc[j].z, r.B[2 * j + 1] = PAR ? pc[j].z : pc[j].w;
		const float d0 = dc[j].x * dx2, d1 = dc[j].y * dx2, d2 = dc[j].z * dx2, d3 = dc[j].w * dx2;  // Kernel.cu:621: divVal * dx2
		r.dR[2 * j] = PAR ? d1 : d0, r.dB[2 * j] = PAR ? d0 : d1;
		r.dR[2 * j + 1] = PAR ? d3 : d2, r.dB[2 * j + 1] = PAR ? d2 : d3;
	}
	if (valid) {
		float4* LR = L.arr(PAR ? 1 : 0, 0) + i * HS4;
		float4* LK = L.arr(PAR ? 1 : 0, 1) + i * HS4;
#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			const int j = 4 * q;
			LR[q] = make_float4(r.R[j], r.R[j + 1], j + 2 < HALF ? r.R[j + 2 < HALF ? j + 2 : 0] : 0.0f, j + 3 < HALF ? r.R[j + 3 < HALF ? j + 3 : 0] : 0.0f);
			LK[q] = make_float4(r.B[j], r.B[j + 1], j + 2 < HALF ? r.B[j + 2 < HALF ? j + 2 : 0] : 0.0f, j + 3 < HALF ? r.B[j + 3 < HALF ? j + 3 : 0] : 0.0f);
		}
	}
	// (the barrier behind the staging; and: is every leaf under the tile present? Then no voxel needs masking)
	bool mine = true;
#pragma unroll
	for (int cz = 0; cz < C; ++cz) mine = mine && (r.ok[cz] != 0u || !valid);
	const bool all_present = __syncthreads_and(mine) != 0;
	if (all_present)
		SbSweeps<LB, K, 1, NS, PAR, false>::run(r, L, i, b, dist, omega);
	else
		SbSweeps<LB, K, 1, NS, PAR, true>::run(r, L, i, b, dist, omega);
	if (dist >= H) {  // the rows of the block itself
#pragma unroll
		for (int j = H / 4; j < NCH - H / 4; ++j) {
			const int cz = (4 * j - H + 8) >> 3, zl = (4 * j - H) & 7;
			sb4f v;
			v.x = PAR ? r.B[2 * j] : r.R[2 * j], v.y = PAR ? r.R[2 * j] : r.B[2 * j];
			v.z = PAR ? r.B[2 * j + 1] : r.R[2 * j + 1], v.w = PAR ? r.R[2 * j + 1] : r.B[2 * j + 1];
			sb_store4(v, ro, (int)(base[cz] + (unsigned)(zl * 4)), 0, 0);
		}
	}
}

// a thread's z-row in the 16^3-block (XY) form: the row's p lives in LDS, where the neighbours need it anyway; the thread keeps div * dx^2
template <int C, int HALF>
struct SbLeanRow {
	float dR[HALF], dB[HALF];  // div * dx^2 of the row's red / black voxels
	unsigned ok[C];            // per leaf cell along z: all ones if the leaf exists, else 0
};

// 16 bytes (z = 4 * half .. 4 * half + 3 of z-row `row`) of boundary leaf `leaf` into the ghost copies the peers keep of it: the chained
// blocked sweep of a multi-GPU rank (hns_flags.hpp). `mine` = this lane's piece belongs to `leaf`; the table walk is wave-uniform.
__device__ __forceinline__ void chain_store_piece(const PhaseMirror& m, int leaf, bool mine, int row, int half, sb4f v) {
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		if (!mine) continue;
		const unsigned bits = m.mask ? (m.mask[(size_t)e * 64 + row] >> (4 * half)) & 0xFu : 0xFu;
		float* r = chain_out(m, t.x, 0) + (size_t)t.y * 512 + row * 8 + 4 * half;
		if (bits == 0xFu) {
			store_through(r, chain_v4f{v.x, v.y, v.z, v.w});
		} else if (bits) {
			const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int z = 0; z < 4; ++z)
				if (bits >> z & 1) store_through(r + z, f[z]);
		}
	}
}

// the same piece into the peers' MESSAGES (PackMirror: the boundary sweep of an exchanged pressure loop packs what it stores)
__device__ __forceinline__ void chain_store_piece(const PackMirror& m, int leaf, bool mine, int row, int half, sb4f v) {
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		if (!mine) continue;
		const unsigned byte = m.mask[(size_t)e * 64 + row];
		const unsigned bits = (byte >> (4 * half)) & 0xFu;
		if (!bits) continue;
		float* q = m.msg[t.x] + (size_t)t.y + m.row_pre[(size_t)e * 64 + row] + __popc(byte & ((1u << (4 * half)) - 1u));
		const float f[4] = {v.x, v.y, v.z, v.w};
		int c = 0;
#pragma unroll
		for (int z = 0; z < 4; ++z)
			if (bits >> z & 1) q[c++] = f[z];
	}
}

// ---------------------------------------------------------------------------------------------------------------
// 16^3 blocks (round 5, "XY form"): the thread that SWEEPS a row is the thread that FETCHES it.
// Its predecessor sorted the rows by the parity of x+y into two sections of waves so that the parity was a template parameter of the sweep code; the price was that a wave's rows lay
// 64 bytes apart in memory (every other z-row of a leaf) -- ~40 L1 accesses per load instruction. Round 5 measured that instructions are not what this kernel waits for, and that div
// fetched at coalesced addresses WITHOUT a
// hand-over would be worth 5.7 % at 256^3 (r05_sorblock_notes.txt 11). So here thread t owns the interior row (x, y) = (1 + t / 22, 1 + t % 22): consecutive lanes are consecutive z-rows
// of a leaf (32 bytes apart: ~24 accesses per instruction for p AND div), the parity is a per-lane value (pointer selects at the staging, two selects per updated voxel for the z
// neighbours), there is no fetch mapping, no hand-over and no section. LDS: the same 50,880 bytes, the rows of BOTH parities in one (x, y) order -- a wave's 64 rows are 64 consecutive
// entries of either colour array, so the 16-byte accesses at the 48-byte row stride stay free of bank conflicts:
//   black array: planes x = 0 .. T-1, rows y = 1 .. T-2 (entry x * TC + y - 1), then the rim rows y = 0 / y = T-1 of every plane (entry T * TC + 2 x + (y != 0));
//   red array:   planes x = 1 .. T-2 only (entry (x - 1) * TC + y - 1 = the row's black entry - TC): nobody reads a red value of plane 0 / T-1 or of a rim row.
// A row's colour arrays hold its red / black values in ascending z (which z are red depends on the row's parity: even z if x+y is even); the four lateral neighbours of a voxel have the
// other colour at the same index of their rows' arrays. Voxels within S of the tile's z rim are stale at sweep S: the j ranges below are the UNION over both parities of
// what a parity-sorted sweep would compute (40 instead of 38 updates per row; a stale value is only ever read by stale voxels).
// ---------------------------------------------------------------------------------------------------------------
struct SbLdsXY {
	using G = SbGeo<2, 2>;
	static constexpr int T = G::T, TC = G::TC, HS4 = G::HS4;
	static constexpr int NB = (T * TC + 2 * T) * HS4, NR = TC * TC * HS4;  // float4
	float4 a[NB + NR];  // (padding between the two arrays -- 48 / 64 / 128 bytes -- was measured: nothing, runs/r05ar.sh)
	__device__ __forceinline__ float4* black() { return a; }
	__device__ __forceinline__ float4* red() { return a + NB; }
	static __device__ __forceinline__ int brow(int x, int y) { return x * TC + y - 1; }            // black entry of an interior-y row of any plane
	static __device__ __forceinline__ int brim(int x, int y) { return T * TC + 2 * x + (y != 0); }  // black entry of a rim row (y = 0 or T-1)
};
static_assert(sizeof(SbLdsXY) == 50880, "the XY form must fit three workgroups per CU (160 KB of LDS)");

// one colour sweep S of the row this thread owns: entry `bi` in the black array, bi - TC in the red one; par = (x + y) & 1; yp / ym = black entries of the rows (x, y +- 1)
template <int S, int NS, bool MASKED, class Row>
__device__ __forceinline__ void sb_sweep_xy(Row& r, SbLdsXY& L, const int bi, const int ypb, const int ymb, const int par, const int dist, const float omega) {
	using G = SbGeo<2, 2>;
	constexpr int H = G::H, HALF = G::HALF, HS4 = G::HS4, NQ = G::NQ, TC = G::TC;
	constexpr bool red = (S & 1) != 0;
	const float(&dX)[HALF] = red ? r.dR : r.dB;
	// the colour being updated at z' = 2j + zo: zo = 1 ("up") for the red voxels of an odd row and the black voxels of an even one
	const bool up = red ? par != 0 : par == 0;
	// updated: S <= z' <= T-1-S; union over zo = 0 / 1 (see above)
	constexpr int jlo = S / 2, jhi = (G::T - 1 - S) / 2 + 1;
	constexpr int qlo = jlo / 4, qhi = (jhi + 3) / 4;
	if (dist >= S) {
		// red sweep: own red at bi - TC (red array), own black at bi, lateral neighbours' black at bi +- TC, ypb, ymb.
		// black sweep (dist >= 2: every neighbour is an interior row of planes 1 .. T-2): own black at bi, own red at bi - TC, neighbours' red at bi - TC +- TC, bi - TC +- 1.
		sb4f* LXo4 = reinterpret_cast<sb4f*>((red ? L.red() + (bi - TC) * HS4 : L.black() + bi * HS4));
		const sb4f* LYo4 = reinterpret_cast<const sb4f*>((red ? L.black() + bi * HS4 : L.red() + (bi - TC) * HS4));
		const float4* LY = red ? L.black() : L.red() - TC * HS4;  // (indexed with BLACK entry numbers either way)
		float Y[HALF];
#pragma unroll
		for (int q = 0; q < NQ; ++q) {
			const sb4f y4 = LYo4[q];
			Y[4 * q] = y4.x, Y[4 * q + 1] = y4.y, Y[4 * q + 2] = y4.z, Y[4 * q + 3] = y4.w;
		}
		const sb4f* pxp = reinterpret_cast<const sb4f*>(LY + (bi + TC) * HS4);
		const sb4f* pxm = reinterpret_cast<const sb4f*>(LY + (bi - TC) * HS4);
		const sb4f* pyp = reinterpret_cast<const sb4f*>(LY + (red ? ypb : bi + 1) * HS4);
		const sb4f* pym = reinterpret_cast<const sb4f*>(LY + (red ? ymb : bi - 1) * HS4);
#pragma unroll
		for (int q = qlo; q < qhi; ++q) {
			sb4f x4 = LXo4[q], xp4 = pxp[q], xm4 = pxm[q], yp4 = pyp[q], ym4 = pym[q];
			if (4 * q < jlo || 4 * q + 4 > jhi) {  // a piece only part of which is updated: keep its accesses whole (partial pieces make the compiler split the LDS accesses into narrower, conflicting ones)
				asm volatile("" : "+v"(x4));
				asm volatile("" : "+v"(xp4));
				asm volatile("" : "+v"(xm4));
				asm volatile("" : "+v"(yp4));
				asm volatile("" : "+v"(ym4));
			}
			float X[4] = {x4.x, x4.y, x4.z, x4.w};
			const float xp[4] = {xp4.x, xp4.y, xp4.z, xp4.w}, xm[4] = {xm4.x, xm4.y, xm4.z, xm4.w}, yp[4] = {yp4.x, yp4.y, yp4.z, yp4.w}, ym[4] = {ym4.x, ym4.y, ym4.z, ym4.w};
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int j = 4 * q + e;
				if (j < jlo || j >= jhi) continue;
				const float below = j > 0 ? Y[j > 0 ? j - 1 : 0] : 0.0f, above = j + 1 < HALF ? Y[j + 1 < HALF ? j + 1 : 0] : 0.0f;
				const float zm = up ? Y[j] : below, zp = up ? above : Y[j];
				const float pGS = ((xp[e] + xm[e] + yp[e] + ym[e] + zp + zm) - dX[j]) * kInv6;  // Kernel.cu:621 (dX = div * dx^2)
				const float cand = X[e] + omega * (pGS - X[e]);                                   // Kernel.cu:622
				const int cz = (2 * j - H + 8) >> 3;
				X[e] = MASKED ? __uint_as_float(__float_as_uint(cand) & r.ok[cz]) : cand;
			}
			sb4f o4 = sb4f{X[0], X[1], X[2], X[3]};
			if (4 * q < jlo || 4 * q + 4 > jhi) asm volatile("" : "+v"(o4));
			LXo4[q] = o4;
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	if (S < NS) __syncthreads();
}

// NS colour sweeps: 4 = two (red, black) iterations, the form every solve runs; 2 = one iteration, for an odd one left over (round 6: this replaced the one-iteration
// kernels of rounds 1-2 -- one wave per leaf / per leaf pair -- everywhere: single-GPU leftovers, launch ranges, the chained and the packing sweeps of multi-GPU ranks).
// The tile, its 4-voxel halo and the fetch ranges are those of two iterations either way: after two sweeps the block is further inside the stale rim than it needs to be.
template <int S, int NS, bool MASKED>
struct SbSweepsXY {
	template <class Row>
	static __device__ __forceinline__ void run(Row& r, SbLdsXY& L, const int bi, const int ypb, const int ymb, const int par, const int dist, const float omega) {
		sb_sweep_xy<S, NS, MASKED>(r, L, bi, ypb, ymb, par, dist, omega);
		if constexpr (S < NS) SbSweepsXY<S + 1, NS, MASKED>::run(r, L, bi, ypb, ymb, par, dist, omega);
	}
};

// M = NoMirror, or PhaseMirror for the chained sweep of a multi-GPU rank (k_rbgs_block, which see: boundary workgroups wait for the peers' previous launch and store what the peers
// read of their boundary leaves into the peers' ghost copies as well)
template <bool ZERO, class M = NoMirror, int NS = 4>
__global__ __attribute__((amdgpu_waves_per_eu(6, 8))) __launch_bounds__((SbGeo<2, 2>::NT)) void k_rbgs_block_xy(const int* __restrict__ recs, const int* __restrict__ any_absent, const float* __restrict__ div, const float* __restrict__ p_in,
                                                                                                                float* __restrict__ p_out, const unsigned field_bytes, const float dx2, const float omega, const M m = M{}) {
	using G = SbGeo<2, 2>;
	constexpr int H = G::H, T = G::T, C = G::C, HALF = G::HALF, NCH = G::NCH, HS4 = G::HS4, TC = G::TC;
	__shared__ SbLdsXY L;
	__shared__ int s_rec[64];
	const int t = threadIdx.x;
	const bool valid = t < TC * TC;
	const int xq = valid ? t / TC : 0;
	const int x = 1 + xq, y = valid ? 1 + t - xq * TC : 1;
	const int par = (x + y) & 1;
	const int bi = SbLdsXY::brow(x, y);
	const int ypb = y == T - 2 ? SbLdsXY::brim(x, T - 1) : bi + 1, ymb = y == 1 ? SbLdsXY::brim(x, 0) : bi - 1;
	const int dist = valid ? min(min(x, T - 1 - x), min(y, T - 1 - y)) : -1;
	const int cx = (x - H + 8) >> 3, cy = (y - H + 8) >> 3;
	const unsigned row_bytes = (unsigned)(((((x - H) & 7) << 3) | ((y - H) & 7)) * 32);
	const int* brec = recs + (size_t)blockIdx.x * G::REC;
	// (id fetches: unconditional, each issued as soon as its address is known, nothing that uses one between them: dependent round trips at the head of the workgroup are what this kernel waits for)
	const int rec_word = brec[t & 63];
	const int4 q4 = *reinterpret_cast<const int4*>(brec + (valid ? (cx * C + cy) * C : 0));
	__builtin_amdgcn_sched_barrier(0);
	// rim duty: the black values of the rows of the planes x = 0 / T-1 and of the rim rows y = 0 / T-1 (nobody updates them), one 16-byte piece j = 1 .. NCH-2 per thread (the end
	// pieces cannot reach the block and are zeroed); a rim row off the block's own range in its other coordinate cannot either and reads as 0
	constexpr int RJ = NCH - 2, NRIM = 4 * TC;
	static_assert(NRIM * RJ <= G::NT && RJ == 4, "one rim piece per thread");
	const bool rim_on = t < NRIM * RJ;
	int rim_b2, rim_zero, rim_par;  // (float2 index of the piece in the black array)
	unsigned rim_off;
	bool rim_want;
	int rim_at;
	{
		const int mr = rim_on ? t >> 2 : 0, j = 1 + (t & 3);
		const int side = mr / TC, along = 1 + (mr - side * TC);
		const int rx = side == 0 ? 0 : (side == 1 ? T - 1 : along), ry = side == 2 ? 0 : (side == 3 ? T - 1 : along);
		const int rcx = (rx - H + 8) >> 3, rcy = (ry - H + 8) >> 3, rcz = (4 * j - H + 8) >> 3;
		rim_off = (unsigned)(((((rx - H) & 7) << 3) | ((ry - H) & 7)) * 32 + ((4 * j - H) & 7) * 4);
		rim_b2 = (side >= 2 ? SbLdsXY::brim(rx, ry) : SbLdsXY::brow(rx, ry)) * HS4 * 2 + j;
		rim_par = (rx + ry) & 1;
		rim_want = !ZERO && rim_on && along >= H && along < T - H;
		rim_at = rim_want ? (rcx * C + rcy) * C + rcz : 0;
		rim_zero = j == 1 ? -1 : (j == RJ ? 1 : 0);
	}
	const int rim_id = brec[rim_at];
	const int meta_word = any_absent[blockIdx.x];
	__builtin_amdgcn_sched_barrier(0);
	SbLeanRow<C, HALF> r;
	unsigned base[C], dbase[C];  // where the row's pieces lie in p / in div (what cannot reach the block in 2K sweeps -- div: 2K - 1 -- is not fetched: an offset beyond the field reads 0)
	constexpr unsigned kBeyond = 0xFFFFF000u;
	{
		const int ids[C] = {valid ? q4.x : -1, valid ? q4.y : -1, valid ? q4.z : -1, valid ? q4.w : -1};
		const int e = max(0, max(H - x, x - (T - 1 - H))) + max(0, max(H - y, y - (T - 1 - H)));
#pragma unroll
		for (int cz = 0; cz < C; ++cz) {
			r.ok[cz] = ids[cz] >= 0 ? 0xFFFFFFFFu : 0u;
			const unsigned b = (unsigned)ids[cz] * 2048u + row_bytes;
			const bool end = cz == 0 || cz == C - 1;  // (the end pieces are the only users of the first and last leaf cell along z)
			base[cz] = e > (end ? H - 1 : H) ? kBeyond : b;
			dbase[cz] = e > (end ? H - 2 : H - 1) ? kBeyond : b;
		}
	}
	// A chained rank's boundary workgroup waits for the peers' previous launch HERE -- behind the id fetches, which read nothing a peer writes, and in front of the first load that can touch
	// a ghost voxel: the flag's round trip overlaps the ids' instead of preceding it.
	int chain_leaf = 0x7fffffff;  // (chain_begin / chain_end take a leaf number: below n_boundary = "this workgroup waits and mirrors")
	if constexpr (!std::is_same<M, NoMirror>::value) {
		if (__builtin_amdgcn_readfirstlane(meta_word) & 2) chain_leaf = 0;
		chain_begin(m, chain_leaf);
		__syncthreads();
	}
	const sb4i rp = sb_rsrc(p_in, field_bytes), rd = sb_rsrc(div, field_bytes), ro = sb_rsrc(p_out, field_bytes);
	sb4f pc[NCH], dc[NCH], rimv;
#pragma unroll
	for (int j = 0; j < NCH; ++j) pc[j] = ZERO ? sb4f{0.0f, 0.0f, 0.0f, 0.0f} : sb_load4(rp, (int)(base[(4 * j - H + 8) >> 3] + (unsigned)(((4 * j - H) & 7) * 4)), 0, 0);
	__builtin_amdgcn_sched_barrier(0);
	rimv = ZERO ? sb4f{0.0f, 0.0f, 0.0f, 0.0f} : sb_load4(rp, (int)((unsigned)(rim_want ? rim_id : -1) * 2048u + rim_off), 0, 0);
#pragma unroll
	for (int j = 0; j < NCH; ++j) dc[j] = sb_load4(rd, (int)(dbase[(4 * j - H + 8) >> 3] + (unsigned)(((4 * j - H) & 7) * 4)), 0, 0);
	if (t < 64) s_rec[t] = rec_word;  // (the block record for the store phase; visible behind the staging barrier)
	if (rim_on) {
		float2* LK = reinterpret_cast<float2*>(L.black()) + rim_b2;  // (black only: even z of a row with odd x+y, odd z of one with even x+y)
		*LK = rim_par ? make_float2(rimv.x, rimv.z) : make_float2(rimv.y, rimv.w);
		if (rim_zero) LK[rim_zero] = make_float2(0.0f, 0.0f);
	}
	if (valid) {
		// the row's p split by colour: its even z are red if x+y is even, black if odd
		float4* LE = par ? L.black() + bi * HS4 : L.red() + (bi - TC) * HS4;
		float4* LO = par ? L.red() + (bi - TC) * HS4 : L.black() + bi * HS4;
#pragma unroll
		for (int q = 0; q < G::NQ; ++q) {
			const sb4f u = pc[2 * q], v = pc[2 * q + 1];
			LE[q] = make_float4(u.x, u.z, v.x, v.z);
			LO[q] = make_float4(u.y, u.w, v.y, v.w);
		}
	}
	const int meta = __builtin_amdgcn_readfirstlane(meta_word);
#pragma unroll
	for (int j = 0; j < NCH; ++j) {
		const float d0 = dc[j].x * dx2, d1 = dc[j].y * dx2, d2 = dc[j].z * dx2, d3 = dc[j].w * dx2;  // Kernel.cu:621: divVal * dx2
		r.dR[2 * j] = par ? d1 : d0, r.dB[2 * j] = par ? d0 : d1;
		r.dR[2 * j + 1] = par ? d3 : d2, r.dB[2 * j + 1] = par ? d2 : d3;
	}
	__syncthreads();
	if ((meta & 1) == 0)
		SbSweepsXY<1, NS, false>::run(r, L, bi, ypb, ymb, par, dist, omega);
	else
		SbSweepsXY<1, NS, true>::run(r, L, bi, ypb, ymb, par, dist, omega);
	// store phase: the block's 16 x 16 rows x four 16-byte pieces in memory order, values out of the rows' LDS entries, leaf ids out of the record's LDS copy
	__syncthreads();
	{
		int mirror_id[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // chained rank: the block's leaves this launch stores, as scalars (-1: not stored)
		if constexpr (!std::is_same<M, NoMirror>::value) {
			if (meta & 2) {
#pragma unroll
				for (int c = 0; c < 8; ++c)
					mirror_id[c] = (meta >> (8 + c)) & 1 ? __builtin_amdgcn_readfirstlane(s_rec[((1 + (c >> 2)) * C + 1 + ((c >> 1) & 1)) * C + 1 + (c & 1)]) : -1;
			}
		}
		const float2* B2 = reinterpret_cast<const float2*>(L.black());
		const float2* R2 = reinterpret_cast<const float2*>(L.red());
#pragma unroll
		for (int n = 0; n < 2; ++n) {
			const int pq = t + n * G::NT;
			const int jz = pq & 1, yy = (pq >> 1) & 15, czb = (pq >> 5) & 1, xx = pq >> 6;
			const int sx = xx + H, sy = yy + H, j = H / 4 + 2 * czb + jz;
			const int sp = (sx + sy) & 1;
			const int sb = SbLdsXY::brow(sx, sy);
			const float2 rr = R2[(sb - TC) * HS4 * 2 + j], bb = B2[sb * HS4 * 2 + j];
			const int cell = (((xx >> 3) << 1) | (yy >> 3)) << 1 | czb;
			const int id = (meta >> (8 + cell)) & 1 ? s_rec[((1 + (xx >> 3)) * C + 1 + (yy >> 3)) * C + 1 + czb] : -1;  // (a leaf outside the launch range is a source only)
			sb4f v;
			v.x = sp ? bb.x : rr.x, v.y = sp ? rr.x : bb.x, v.z = sp ? bb.y : rr.y, v.w = sp ? rr.y : bb.y;
			sb_store4(v, ro, (int)((unsigned)id * 2048u + (unsigned)((((xx & 7) << 3) | (yy & 7)) * 32 + jz * 16)), 0, 0);
			if constexpr (!std::is_same<M, NoMirror>::value) {
				// (piece n of every thread lies in the block's x half n, the walk over that half's four leaves and over a leaf's table entries is wave-uniform)
				if (meta & 2) {
#pragma unroll
					for (int cc = 0; cc < 4; ++cc) {
						const int c = 4 * n + cc;
						const int lc = mirror_id[c];
						if (lc < 0 || lc >= m.n_boundary) continue;
						chain_store_piece(m, lc, c == cell, ((xx & 7) << 3) | (yy & 7), jz, v);
					}
				}
			}
		}
	}
	if constexpr (!std::is_same<M, NoMirror>::value) chain_end(m, chain_leaf);
}

// recs: one record per workgroup, in launch order. LB = 1: {leaf, nbr27[27]} (the grid's d_blk); LB = 2: the 4 x 4 x 4 leaves
// under the tile, cell (cx, cy, cz) at (cx*4 + cy)*4 + cz, -1 = absent. ZERO: p_in is known to be 0 (first launch of a solve,
// HNanoSolver.cu:113) and is not read. The first half of the workgroup's waves takes the rows with even x+y, the second half
// those with odd x+y; both halves meet at the same number of barriers.
template <int LB, int K, bool ZERO, int NS = 2 * K>
__global__ __launch_bounds__((SbGeo<LB, K>::NT)) void k_rbgs_block(const int* __restrict__ recs, const float* __restrict__ div, const float* __restrict__ p_in, float* __restrict__ p_out,
                                                                   const unsigned field_bytes, const float dx2, const float omega) {
	using G = SbGeo<LB, K>;
	static_assert(LB == 1, "rows in registers: one-leaf blocks (16^3 blocks are swept by k_rbgs_block_xy)");
	__shared__ SbLds<LB, K> L;
	const int t = threadIdx.x;
	if (__builtin_amdgcn_readfirstlane(t >= G::SEC))
		sb_body<LB, K, ZERO, true, NS>(L, t - G::SEC, recs, div, p_in, p_out, field_bytes, dx2, omega);
	else
		sb_body<LB, K, ZERO, false, NS>(L, t, recs, div, p_in, p_out, field_bytes, dx2, omega);
}

// ---------------------------------------------------------------------------------------------------------------
// block tables for LB = 2: aligned 16^3-voxel blocks (2 x 2 x 2 leaves, any of them may be absent)
// ---------------------------------------------------------------------------------------------------------------

// flag[l] = 1 if leaf l is the lowest-numbered leaf OF THE LAUNCH RANGE [first, first + count) in its block. A launch
// range is what a multi-GPU rank sweeps (its boundary leaves, its interior leaves, ...: hns_grid_set_active_range); the leaves of the
// grid outside it -- ghost leaves among them -- are sources of the tiles and are never stored.
__global__ __launch_bounds__(256) void k_sb_leader(GridDev g, int first, int count, int* __restrict__ flag) {
	const int l = blockIdx.x * 256 + threadIdx.x;
	if (l >= g.n_leaves) return;
	int lead = (l >= first && l < first + count) ? 1 : 0;
	if (lead) {  // the leaf of the range with the LOWEST NUMBER in its block leads it: the blocks then come out ordered by that number, and a
		// multi-GPU rank's boundary leaves -- the first of its leaf order -- put the blocks that hold one in front of all others
		const int4 o = g.origins[l];
		const int bx = o.x & ~15, by = o.y & ~15, bz = o.z & ~15;
		for (int c = 0; c < 8; ++c) {
			const int q = d_find_leaf(g, bx + 8 * (c >> 2), by + 8 * ((c >> 1) & 1), bz + 8 * (c & 1));
			if (q >= first && q < l) lead = 0;
		}
	}
	flag[l] = lead;
}

// exclusive scan of flag[0..n) by ONE workgroup; leaders[pos] = l for flagged l; *total = number of flags
// total[1] = number of flagged l below `n_head` (the blocks led by a boundary leaf of a multi-GPU rank)
__global__ __launch_bounds__(1024) void k_sb_compact(const int* __restrict__ flag, int n, int* __restrict__ leaders, int* __restrict__ total, int n_head) {
	__shared__ int s_part[1024];
	const int per = (n + 1023) / 1024;
	const int lo = threadIdx.x * per, hi = min(n, lo + per);
	int sum = 0;
	for (int i = lo; i < hi; ++i) sum += flag[i];
	s_part[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const int v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
		__syncthreads();
		s_part[threadIdx.x] += v;
		__syncthreads();
	}
	int run = s_part[threadIdx.x] - sum;
	for (int i = lo; i < hi; ++i) {
		if (i == n_head) total[1] = run;
		if (flag[i]) leaders[run++] = i;
	}
	if (threadIdx.x == 1023) {
		total[0] = s_part[1023];
		if (n_head >= n) total[1] = s_part[1023];
	}
}

// record of launch position b: the 64 leaves under the tile of block order[b]
// Behind the records one word per block (zeroed by the host): bit 0 = "a leaf under this block's tile is absent" (its voxels must not be
// updated), bit 8 + c = leaf c of the block (c = (cx*2 + cy)*2 + cz) lies in the launch range [first, first + count) and is stored.
// Bit 1 = the block holds a leaf of the range below n_boundary: a boundary leaf of a multi-GPU rank (the chained sweep's workgroup waits / mirrors).
__global__ __launch_bounds__(256) void k_sb_table(GridDev g, const int* __restrict__ leaders, int n_blocks, int seg, int pre, int first, int count, int n_boundary, int* __restrict__ tab) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (int64_t)n_blocks * 64) return;
	const int b = (int)(i >> 6), c = (int)(i & 63);
	const int4 o = g.origins[leaders[sched_leaf(b, n_blocks, seg, pre)]];
	const int cx = c >> 4, cy = (c >> 2) & 3, cz = c & 3;
	const int64_t x = (int64_t)(o.x & ~15) + 8 * (cx - 1), y = (int64_t)(o.y & ~15) + 8 * (cy - 1), z = (int64_t)(o.z & ~15) + 8 * (cz - 1);
	int leaf = -1;
	if (x >= INT32_MIN && x <= INT32_MAX && y >= INT32_MIN && y <= INT32_MAX && z >= INT32_MIN && z <= INT32_MAX) leaf = d_find_leaf(g, (int)x, (int)y, (int)z);
	tab[i] = leaf;
	int* meta = tab + (int64_t)n_blocks * 64 + b;
	if (leaf < 0) atomicOr(meta, 1);
	const bool inner = ((cx - 1) | (cy - 1) | (cz - 1)) >= 0 && cx <= 2 && cy <= 2 && cz <= 2;
	if (inner && leaf >= first && leaf < first + count) atomicOr(meta, (1 << (8 + ((((cx - 1) << 1) | (cy - 1)) << 1 | (cz - 1)))) | (leaf < n_boundary ? 2 : 0));
}

}  // namespace hns

using namespace hns;

// Block records of the 16^3 form for the whole grid, built on first use (most small grids never ask). The table comes
// out of the arena pool and goes back with the grid.
//
// Round 4: (1) the records cover the grid's LAUNCH RANGE (hns_grid_set_active_range): blocks that hold a leaf of the range, every leaf
// of the grid under their tiles as a source, stores to the leaves of the range only -- what a multi-GPU rank's boundary / interior /
// owned ranges need (the ghost layer, 8 voxels deep, holds everything 2K = 4 sweeps can carry into an owned voxel). (2) The build runs
// on a stream of its own and waits for that stream only: it used to launch on the NULL stream and call hipDeviceSynchronize() from
// inside an asynchronous per-stream entry point, stalling every stream of the device on first use (concurrent cooks). (3) A table that
// was handed to launches is never returned to the pool while the grid lives (another host thread's launch may still read it): tables
// superseded by an option or range change are parked until the grid goes (hns_grid_retire_blocks).
void hns_grid_retire_blocks(hns_grid* g);

int hns_grid_build_blocks(hns_grid* g) {
	std::lock_guard<std::mutex> lock(g->build_mutex);
	const int seg = 0;  // (blocks per XCD segment of the launch order: one chunk per XCD; segments of N blocks were measured in round 4 and bought nothing)
	if (g->sb_built && g->sb_seg == seg && g->sb_first == g->first_active && g->sb_count == g->n_active) return HNS_OK;
	if (g->d_sb_tab) {
		//  a grid whose launch range or segment option changes every frame would park one table per change until it is destroyed: beyond
		// 64 parked tables wait for the device -- a launch that was handed one of them has finished then -- and give them all back to the pool
		// (that wait stalls every stream of the device from inside an asynchronous entry point, so it is the last resort of a pathological caller -- 64 rebuilds
		// of one grid's records, each a megabyte at 256^3 -- not something a cook meets: a launch range changes when hns_dist builds its four ranges, once each)
		if (g->sb_retired.size() >= 64) {
			HNS_HIP(hipDeviceSynchronize());
			hns_grid_retire_blocks(g);
		}
		g->sb_retired.emplace_back(g->d_sb_tab, g->sb_bytes);
	}
	g->d_sb_tab = nullptr;
	g->sb_bytes = 0;
	g->n_sb = 0;
	g->sb_built = true;
	g->sb_seg = seg;
	g->sb_first = g->first_active, g->sb_count = g->n_active;
	const int n = (int)g->topo.n_leaves;
	if (n == 0 || g->n_active == 0 || !g->d_scratch) return HNS_OK;
	int* flag = (int*)g->d_scratch;  // flag[n] | leaders[n] | total[2]
	int* leaders = flag + n;
	int* total = leaders + n;
	GridDev gd = g->dev();
	const int first = (int)g->first_active, count = (int)g->n_active;
	// (chain_boundary: the leading leaves of the range that are a multi-GPU rank's boundary leaves -- hns_dist_*.hip sets it on the range its
	// chained sweeps run over; 0 everywhere else. sched_prefix != 0: deal the blocks they lead out to all XCDs first)
	const int n_boundary = g->chain_boundary ? first + (int)std::min<uint64_t>(g->chain_boundary, g->n_active) : 0;
	hipStream_t st = nullptr;
	HNS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	struct Drop {
		hipStream_t s;
		~Drop() { (void)hipStreamDestroy(s); }
	} drop{st};
	hipLaunchKernelGGL(k_sb_leader, dim3((n + 255) / 256), dim3(256), 0, st, gd, first, count, flag);
	hipLaunchKernelGGL(k_sb_compact, dim3(1), dim3(1024), 0, st, (const int*)flag, n, leaders, total, n_boundary);
	int tot[2] = {0, 0};
	HNS_HIP(hipMemcpyAsync(tot, total, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
	HNS_HIP(hipStreamSynchronize(st));
	const int nb = tot[0];
	if (nb <= 0) return fail(HNS_ERR_RUNTIME, "hns_grid_build_blocks: no block leader found");
	if (int rc = hns_arena_get(sizeof(int) * 65 * (size_t)nb, g->device, &g->d_sb_tab, &g->sb_bytes)) return rc;
	HNS_HIP(hipMemsetAsync((int*)g->d_sb_tab + (size_t)nb * 64, 0, sizeof(int) * (size_t)nb, st));
	// the blocks led by a boundary leaf are dealt out to all XCDs first (as hns_grid_upload_schedule does with the boundary leaves themselves)
	const int pre = (n_boundary && g->sched_prefix) ? (std::min(tot[1], nb) & ~7) : 0;
	hipLaunchKernelGGL(k_sb_table, dim3((unsigned)(((int64_t)nb * 64 + 255) / 256)), dim3(256), 0, st, gd, (const int*)leaders, nb, seg, pre, first, count, n_boundary, (int*)g->d_sb_tab);
	HNS_HIP(hipStreamSynchronize(st));
	g->n_sb = (uint64_t)nb;
	return HNS_OK;
}

void hns_grid_retire_blocks(hns_grid* g) {  // (the grid is being destroyed or rebuilt: the device is idle for it)
	for (auto& t : g->sb_retired) hns_arena_put(t.first, t.second, g->device);
	g->sb_retired.clear();
}

// Can this grid be swept by the blocked form, and with which block edge (in leaves)? 0 = no.
int hns_rbgs_block_shape(hns_grid* g, int* k_max) {
	*k_max = 0;
	// fields addressable with 32-bit byte offsets. A launch range (a multi-GPU rank's boundary / interior / owned leaves) is swept like a
	// whole grid: blocks that hold a leaf of the range, the other leaves of the grid as sources only (hns_grid_build_blocks). The CALLER
	// vouches that p within 2K voxels of the range, and div within 2K - 1, are current in the leaves outside it (hns_dist_*.hip).
	if (g->n_active == 0 || g->topo.n_leaves > 2000000) return 0;
	int lb = options().sor_block_lb.load(), k = 0;
	// by size (profiles/r03_sor_forms.txt): one-leaf blocks while the grid cannot fill the chip with 16^3 blocks, four iterations per
	// launch while even those leave most of it idle
	if (lb == 0) lb = g->n_active <= 600 ? 1 : 2;  // (512 leaves: 2.97 against 3.36 us per iteration; 729: 3.65 against 3.49)
	if (lb != 1 && lb != 2) return 0;
	if (lb == 1 && !g->d_blk) return 0;
	if (lb == 2 && (hns_grid_build_blocks(g) != HNS_OK || g->n_sb == 0)) return 0;
	if (k == 0) k = (lb == 1 && g->n_active <= 300) ? 4 : 2;
	if (k != 2 && !(k == 4 && lb == 1)) k = 2;
	*k_max = k;
	return lb;
}

// 16^3 blocks are swept by the XY form (k_rbgs_block_xy: row state in LDS, three workgroups per CU, the thread that sweeps a row fetches it), one-leaf blocks by the
// rows-in-registers form (k_rbgs_block<1, K>). The forms this replaced -- 16^3 blocks with the rows in registers, the parity-sorted lean form and its LDS-DMA variant -- are
// history with their measurements: DESIGN_HISTORY.md, profiles/r03_sorblock_notes.txt, r05_sorblock_notes.txt.
bool hns_rbgs_block_lean(hns_grid*, int lb, int k) { return lb == 2 && k == 2; }

// one launch: k iterations src -> dst
int hns_rbgs_block_launch(hns_grid* g, int lb, int k, bool src_is_zero, const float* div, const float* src, float* dst, float dx2, float omega, void* stream) {
	hipStream_t st = (hipStream_t)stream;
	const unsigned bytes = (unsigned)((size_t)g->topo.n_leaves * 2048u);
	// (the records' address and count as ONE snapshot: another host thread sharing the grid may be rebuilding them -- an option changed --,
	// and a superseded table stays valid until the grid goes, but its count must be its own)
	const int* sb_tab;
	uint64_t n_sb;
	{
		std::lock_guard<std::mutex> lock(g->build_mutex);
		sb_tab = (const int*)g->d_sb_tab, n_sb = g->n_sb;
	}
	if (lb == 2 && (!sb_tab || !n_sb)) return fail(HNS_ERR_RUNTIME, "hns_rbgs_block_launch: no block records");
#define SB_LAUNCH(K_)                                                                                                                                                        \
	do {                                                                                                                                                                      \
		if (src_is_zero)                                                                                                                                                      \
			hipLaunchKernelGGL((k_rbgs_block<1, K_, true>), dim3((unsigned)g->n_active), dim3(SbGeo<1, K_>::NT), 0, st, (const int*)g->d_blk, div, src, dst, bytes, dx2, omega);  \
		else                                                                                                                                                                  \
			hipLaunchKernelGGL((k_rbgs_block<1, K_, false>), dim3((unsigned)g->n_active), dim3(SbGeo<1, K_>::NT), 0, st, (const int*)g->d_blk, div, src, dst, bytes, dx2, omega); \
	} while (0)
	if (lb == 1 && k == 2) SB_LAUNCH(2);
	else if (lb == 1 && k == 4) SB_LAUNCH(4);
	else if (lb == 1 && k == 1) {  // an odd iteration left over: the tile of two iterations, two colour sweeps
		if (src_is_zero)
			hipLaunchKernelGGL((k_rbgs_block<1, 2, true, 2>), dim3((unsigned)g->n_active), dim3(SbGeo<1, 2>::NT), 0, st, (const int*)g->d_blk, div, src, dst, bytes, dx2, omega);
		else
			hipLaunchKernelGGL((k_rbgs_block<1, 2, false, 2>), dim3((unsigned)g->n_active), dim3(SbGeo<1, 2>::NT), 0, st, (const int*)g->d_blk, div, src, dst, bytes, dx2, omega);
	}
	else if (lb == 2 && (k == 2 || k == 1)) {
#define XY_LAUNCH(ZERO_, NS_) hipLaunchKernelGGL((k_rbgs_block_xy<ZERO_, NoMirror, NS_>), dim3((unsigned)n_sb), dim3(SbGeo<2, 2>::NT), 0, st, sb_tab, sb_tab + (size_t)n_sb * 64, div, src, dst, bytes, dx2, omega, NoMirror{})
		if (k == 2 && src_is_zero) XY_LAUNCH(true, 4);
		else if (k == 2) XY_LAUNCH(false, 4);
		else if (src_is_zero) XY_LAUNCH(true, 2);
		else XY_LAUNCH(false, 2);
#undef XY_LAUNCH
	} else return fail(HNS_ERR_INVALID_ARGUMENT, "hns_rbgs_block_launch: unsupported block shape");
#undef SB_LAUNCH
	return HNS_OK;
}

// is the grid's launch range swept two iterations per launch in 16^3 blocks by the XY form (the one form that can pack a rank's messages as it stores)?
extern "C" __attribute__((visibility("hidden"))) bool hns_rbgs_block_packable(hns_grid* g) {
	int k = 0;
	return g && g->n_active && hns_rbgs_block_shape(g, &k) == 2 && k == 2;
}

// The sweep of an exchanged pressure loop that packs its own messages (hns_flags.hpp: PackMirror), two iterations per launch over the grid's launch range (the rank's
// boundary leaves, or -- round 6 -- all of its owned leaves); *done = false and nothing launched where that range is not swept in 16^3 blocks by the XY form.
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_pack_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                 const hns::PackMirror* m, void* stream, bool* done, int iterations) {
	*done = false;
	if (!hns_rbgs_block_packable(g)) return HNS_OK;
	if (int rc = hns_grid_build_blocks(g)) return rc;  // (the records of THIS launch range under the current options: ADVICE r5 -- not whatever table a previous launch left)
	const unsigned bytes = (unsigned)((size_t)g->topo.n_leaves * 2048u);
	const int* tab;
	uint64_t n_sb;
	{
		std::lock_guard<std::mutex> lock(g->build_mutex);
		tab = (const int*)g->d_sb_tab, n_sb = g->n_sb;
	}
	if (!tab || !n_sb) return HNS_OK;
	const float dx2 = dx * dx;  // Kernel.cu:608
#define XY_LAUNCH(ZERO_, NS_) hipLaunchKernelGGL((k_rbgs_block_xy<ZERO_, PackMirror, NS_>), dim3((unsigned)n_sb), dim3(SbGeo<2, 2>::NT), 0, (hipStream_t)stream, tab, tab + (size_t)n_sb * 64, div, src, dst, bytes, dx2, omega, *m)
	if (iterations != 1 && iterations != 2) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_rbgs_block_pack_launch: one or two iterations per launch");
	if (iterations == 2 && src_is_zero) XY_LAUNCH(true, 4);
	else if (iterations == 2) XY_LAUNCH(false, 4);
	else if (src_is_zero) XY_LAUNCH(true, 2);
	else XY_LAUNCH(false, 2);
#undef XY_LAUNCH
	*done = true;
	return HNS_OK;
}

// The chained sweep of a multi-GPU rank, two iterations per launch (hns_dist_*.hip; hns_flags.hpp): ONE launch over the rank's owned leaves
// that waits for the peers where it reads their values and delivers its own boundary values into their ghost voxels. 16^3 blocks only.
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_mirror_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                   const hns::PhaseMirror* m, void* stream, int iterations) {
	int k = 0;
	if (hns_rbgs_block_shape(g, &k) != 2 || k != 2) return fail(HNS_ERR_RUNTIME, "hns_rbgs_block_mirror_launch: this launch range is not swept in 16^3 blocks");
	const unsigned bytes = (unsigned)((size_t)g->topo.n_leaves * 2048u);
	const int* tab;
	uint64_t n_sb;
	{
		std::lock_guard<std::mutex> lock(g->build_mutex);
		tab = (const int*)g->d_sb_tab, n_sb = g->n_sb;
	}
	if (!tab || !n_sb) return fail(HNS_ERR_RUNTIME, "hns_rbgs_block_mirror_launch: no block records");
	const float dx2 = dx * dx;  // Kernel.cu:608
#define XY_LAUNCH(ZERO_, NS_) hipLaunchKernelGGL((k_rbgs_block_xy<ZERO_, PhaseMirror, NS_>), dim3((unsigned)n_sb), dim3(SbGeo<2, 2>::NT), 0, (hipStream_t)stream, tab, tab + (size_t)n_sb * 64, div, src, dst, bytes, dx2, omega, *m)
	if (iterations != 1 && iterations != 2) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_rbgs_block_mirror_launch: one or two iterations per launch");
	if (iterations == 2 && src_is_zero) XY_LAUNCH(true, 4);
	else if (iterations == 2) XY_LAUNCH(false, 4);
	else if (src_is_zero) XY_LAUNCH(true, 2);
	else XY_LAUNCH(false, 2);
#undef XY_LAUNCH
	return HNS_OK;
}

