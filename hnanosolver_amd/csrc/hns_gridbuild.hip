// hns_gridbuild.hip -- index-grid construction on the device.
//
// The reference rebuilds its NanoGrid<ValueOnIndex> on the GPU every cook (create_index_grid, reference
// src/Cuda/HNanoSolver.cu:375-384 -> externals/nanovdb/tools/cuda/PointsToGrid.cuh:511-1064: ~25 launches and 8 CUB
// sorts/scans over all N voxel coordinates). Here only the leaf origins (one coordinate in 512) cross PCIe; the origin
// hash, the 27-neighbour table and the launch-ordered wave records the kernels read are built by five small kernels
// over the leaves. Integer work only; tests/test_gridbuild_gpu.py checks every table against the host builder of
// hns_topology.cpp.
#include <cstdlib>
#include <cstring>

#include "hns_device.hpp"

#define HNS_TRY_RC(call)           \
	do {                            \
		int rc__ = (call);          \
		if (rc__ != HNS_OK) return rc__; \
	} while (0)

namespace hns {

// ---------------------------------------------------------------------------------------------------------------
// origin hash + neighbour table
// ---------------------------------------------------------------------------------------------------------------

// One thread per leaf: linear-probe insert with compare-and-swap. Entries are never removed, so a probe that meets
// an occupied slot can compare origins safely. status[0] receives the smallest leaf index that found its origin
// already present (INT_MAX when none).
__global__ __launch_bounds__(256) void k_hash_insert(const int4* __restrict__ origins, int n, int* __restrict__ hash, uint32_t mask, int* __restrict__ status) {
	const int l = blockIdx.x * 256 + threadIdx.x;
	if (l >= n) return;
	const int4 o = origins[l];
	uint32_t s = d_hash_origin(o.x, o.y, o.z) & mask;
	for (;;) {
		int cur = hash[s];
		if (cur < 0) {
			cur = atomicCAS(&hash[s], -1, l);
			if (cur < 0) return;  // slot was empty and is now ours
		}
		const int4 q = origins[cur];
		if (q.x == o.x && q.y == o.y && q.z == o.z) {
			atomicMin(status, l > cur ? l : cur);
			return;
		}
		s = (s + 1) & mask;
	}
}

// One thread per (leaf, neighbour slot). 64-bit arithmetic so that origins at the int32 edge cannot wrap into a
// valid neighbour (same guard as Topology::build_tables).
__global__ __launch_bounds__(256) void k_build_nbr27(GridDev g, int* __restrict__ nbr27) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= (int64_t)g.n_leaves * 27) return;
	const int l = (int)(t / 27), j = (int)(t % 27);
	const int4 o = g.origins[l];
	const int64_t nx = (int64_t)o.x + 8 * (j / 9 - 1), ny = (int64_t)o.y + 8 * ((j / 3) % 3 - 1), nz = (int64_t)o.z + 8 * (j % 3 - 1);
	int nb = -1;
	if (nx >= INT32_MIN && nx <= INT32_MAX && ny >= INT32_MIN && ny <= INT32_MAX && nz >= INT32_MIN && nz <= INT32_MAX)
		nb = d_find_leaf(g, (int)nx, (int)ny, (int)nz);
	nbr27[t] = nb;
}

// ---------------------------------------------------------------------------------------------------------------
// launch order
// ---------------------------------------------------------------------------------------------------------------

// (sched_leaf: the block -> leaf order, lives in hns_device.hpp: hns_sorblock.hip orders its leaf blocks with it too)

// {leaf, nbr27[27]} per block in launch order: the kernels that work one leaf per workgroup read their whole
// topology with one fetch.
__global__ __launch_bounds__(256) void k_build_blk(const int* __restrict__ nbr27, int first, int n_active, int seg, int pre, int* __restrict__ sched, int* __restrict__ blk) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t >= (int64_t)n_active * 28) return;
	const int b = (int)(t / 28), j = (int)(t % 28);
	const int l = first + sched_leaf(b, n_active, seg, pre);
	if (j == 0) {
		blk[t] = l;
		if (sched) sched[b] = l;
	} else {
		blk[t] = nbr27[(size_t)l * 27 + (j - 1)];
	}
}

// z-adjacent pairs for the one-wave-per-pair SOR kernel. A z-run is a maximal chain of active leaves linked through
// their -z/+z neighbours; counting from the bottom of its run, an even leaf heads a wave and takes the leaf above it
// as partner (none: the leaf travels alone), an odd leaf is that partner. The rule needs no ordering between leaves,
// leaves the fewest possible lone leaves, and reproduces the aligned (0,1),(2,3).. pairs on a dense grid.
__device__ __forceinline__ int pair_partner(const int* __restrict__ nbr27, int first, int n_active, int l) {
	int steps = 0;
	for (int m = l;;) {
		const int dn = nbr27[(size_t)m * 27 + 12];
		if (dn < first || dn >= first + n_active) break;
		m = dn;
		++steps;
	}
	if (steps & 1) return -2;  // not a head
	const int up = nbr27[(size_t)l * 27 + 14];
	return (up >= first && up < first + n_active) ? up : -1;
}

// pass 1: partner[b] for schedule position b, and the number of wave heads / lone leaves per 256-position block
__global__ __launch_bounds__(256) void k_pair_heads(const int* __restrict__ nbr27, int first, int n_active, int seg, int pre, int* __restrict__ partner, int* __restrict__ block_heads,
                                                    int* __restrict__ totals) {
	__shared__ int s_cnt[2];
	if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const int b = blockIdx.x * 256 + threadIdx.x;
	int p = -2;
	if (b < n_active) {
		p = pair_partner(nbr27, first, n_active, first + sched_leaf(b, n_active, seg, pre));
		partner[b] = p;
	}
	const unsigned long long heads = __ballot(p != -2), lone = __ballot(p == -1);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&s_cnt[0], __popcll(heads));
		atomicAdd(&s_cnt[1], __popcll(lone));
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		block_heads[blockIdx.x] = s_cnt[0];
		atomicAdd(&totals[0], s_cnt[0]);
		atomicAdd(&totals[1], s_cnt[1]);
	}
}

// pass 2: exclusive scan of the per-block head counts (one workgroup; at most 2^22 / 256 = 16384 entries)
__global__ __launch_bounds__(1024) void k_scan_blocks(int* __restrict__ block_heads, int n_blocks) {
	__shared__ int s_part[1024];
	const int per = (n_blocks + 1023) / 1024;
	const int lo = threadIdx.x * per, hi = min(n_blocks, lo + per);
	int sum = 0;
	for (int i = lo; i < hi; ++i) sum += block_heads[i];
	s_part[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
		const int v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
		__syncthreads();
		s_part[threadIdx.x] += v;
		__syncthreads();
	}
	int run = s_part[threadIdx.x] - sum;
	for (int i = lo; i < hi; ++i) {
		const int c = block_heads[i];
		block_heads[i] = run;
		run += c;
	}
}

// pass 3: wave records {leaf0, nbr27, leaf1 or -1, nbr27} (56 ints) in schedule order of their head leaf
__global__ __launch_bounds__(256) void k_write_pairs(const int* __restrict__ nbr27, int first, int n_active, int seg, int pre, const int* __restrict__ partner,
                                                     const int* __restrict__ block_base, int* __restrict__ recs) {
	__shared__ int s_wave[4];
	const int b = blockIdx.x * 256 + threadIdx.x;
	const int p = b < n_active ? partner[b] : -2;
	const unsigned long long heads = __ballot(p != -2);
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (lane == 0) s_wave[w] = __popcll(heads);
	__syncthreads();
	int pos = block_base[blockIdx.x] + __popcll(heads & ((1ull << lane) - 1ull));
	for (int i = 0; i < w; ++i) pos += s_wave[i];
	if (p == -2) return;
	const int l = first + sched_leaf(b, n_active, seg, pre);
	int* r = recs + (size_t)pos * 56;
	r[0] = l;
	r[28] = p;
	for (int j = 0; j < 27; ++j) {
		r[1 + j] = nbr27[(size_t)l * 27 + j];
		r[29 + j] = p < 0 ? -1 : nbr27[(size_t)p * 27 + j];
	}
}


// ---------------------------------------------------------------------------------------------------------------
// tile groups for the blocked SOR kernel (k_rbgs_tile): wy x wz wave records that are each other's y / z neighbours share
// one workgroup and hand their touching faces over through LDS instead of re-reading them from memory
// ---------------------------------------------------------------------------------------------------------------
//
// A record's slot is fixed by the coordinates of its first leaf: a = (y/8) mod wy, c = (z/16) mod wz (records of one column
// are at least 16 voxels apart, so c is unique inside a 16*wz window). Its group is led by the record in slot (0,0) of that
// window -- the head leaf at z = window start, or 8 above it when the z-runs pair up on odd leaf positions. Records whose
// window has no such leader, and groups with a missing member, are swept by the one-wave kernel instead.

__global__ __launch_bounds__(256) void k_head_index(const int* __restrict__ recs, int n_pairs, int* __restrict__ head_of_leaf) {
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	head_of_leaf[recs[(size_t)p * 56]] = p;
}

__global__ __launch_bounds__(256) void k_group_assign(GridDev g, const int* __restrict__ recs, int n_pairs, const int* __restrict__ head_of_leaf, int wy, int wz,
                                                      int* __restrict__ leader, int* __restrict__ members) {
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	const int4 o = g.origins[recs[(size_t)p * 56]];
	const int a = (o.y >> 3) & (wy - 1), c = (o.z >> 4) & (wz - 1);
	const int ay = o.y - 8 * a, az = ((o.z >> 4) - c) << 4;
	int lead = -1;
	for (int k = 0; k < 2 && lead < 0; ++k) {
		const int l = d_find_leaf(g, o.x, ay, az + 8 * k);
		if (l >= 0 && head_of_leaf[l] >= 0) lead = head_of_leaf[l];
	}
	if (lead < 0) lead = p;
	leader[p] = lead;
	members[(size_t)lead * (wy * wz) + a * wz + c] = p;
}

// flags[p] = 1 if p leads a complete group; rest[p] = 1 if p belongs to no complete group
__global__ __launch_bounds__(256) void k_group_flags(const int* __restrict__ leader, const int* __restrict__ members, int n_pairs, int w, int* __restrict__ lead_flag,
                                                     int* __restrict__ rest_flag) {
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pairs) return;
	const int lead = leader[p];
	bool full = true;
	for (int s = 0; s < w; ++s) full &= members[(size_t)lead * w + s] >= 0;
	lead_flag[p] = (full && lead == p) ? 1 : 0;
	rest_flag[p] = full ? 0 : 1;
}

// order-preserving compaction of the indices whose flag is set: pass 1 block counts, (k_scan_blocks), pass 2 write
__global__ __launch_bounds__(256) void k_flag_counts(const int* __restrict__ flag, int n, int* __restrict__ block_counts) {
	__shared__ int s_cnt;
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	const int i = blockIdx.x * 256 + threadIdx.x;
	const unsigned long long b = __ballot(i < n && flag[i]);
	if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, __popcll(b));
	__syncthreads();
	if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}
__global__ __launch_bounds__(256) void k_flag_write(const int* __restrict__ flag, int n, const int* __restrict__ block_base, int* __restrict__ out, int* __restrict__ total) {
	__shared__ int s_wave[4];
	const int i = blockIdx.x * 256 + threadIdx.x;
	const bool f = i < n && flag[i];
	const unsigned long long b = __ballot(f);
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (lane == 0) s_wave[w] = __popcll(b);
	__syncthreads();
	int pos = block_base[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
	for (int k = 0; k < w; ++k) pos += s_wave[k];
	if (f) out[pos] = i;
	if (i == n - 1) *total = pos + (f ? 1 : 0);
}

// group list entry g = the w member records of the g-th complete group
__global__ __launch_bounds__(256) void k_group_write(const int* __restrict__ lead_list, int n_groups, const int* __restrict__ members, int w, int* __restrict__ groups) {
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t >= n_groups * w) return;
	groups[t] = members[(size_t)lead_list[t / w] * w + t % w];
}

}  // namespace hns

using namespace hns;

// Launch order tables for the current n_active: d_sched, d_blk, d_pairs, n_pairs, n_singles. Called at build time and
// whenever hns_grid_set_active_leaves changes the active prefix. All of them are slices of the grid's one device
// allocation (hns_grid_upload), sized for n_active = n_leaves, so nothing is allocated here.
int hns_grid_upload_schedule(hns_grid* g) {
	g->n_pairs = g->n_singles = 0;
	const int n = (int)g->n_active, first = (int)g->first_active;
	if (n == 0) return HNS_OK;
	// One chunk per XCD wins by a wide margin while the sweep arrays fit the Infinity Cache and its neighbourhood (256^3: 40 vs
	// 51 us per sweep against plain leaf order). Beyond it the eight XCDs had better walk through neighbouring stretches of
	// memory together: segments of 128 leaves (1024^3-extent plume, 66k leaves: 127 -> 116 us; 512^3: 441 -> 428, 379 with the
	// blocked kernel; plain order, the rule up to round 1, gives 441 / 131). profiles/micro/sor_one.py schedule_segment=N.
	const int sched_opt = options().schedule.load(), seg_opt = options().schedule_segment.load();
	// option "schedule_segment" = N leaves per XCD segment (0: by size)
	const int seg = sched_opt == kScheduleLinear ? 1 : (sched_opt == kScheduleChunk ? 0 : (seg_opt > 0 ? seg_opt : (n > 40000 ? 128 : 0)));
	const int linear = seg;  // (name kept below: the kernels take the segment length)
	const int pre = (int)std::min<uint64_t>(g->sched_prefix, (uint64_t)n) & ~7;
	g->d_sched = (seg == 1 && pre == 0) ? nullptr : g->d_sched_mem;
	g->sched_seg = seg, g->sched_pre = pre;
	const int* nbr27 = (const int*)g->d_nbr27;
	const int n_blocks = (n + 255) / 256;
	int* partner = (int*)g->d_scratch;  // partner[n] | block_heads[n_blocks] | totals[2]
	int* block_heads = partner + n;
	int* totals = block_heads + n_blocks;
	int h_totals[2] = {0, 0};
	HNS_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(int), 0));
	k_build_blk<<<(unsigned)(((int64_t)n * 28 + 255) / 256), 256, 0, 0>>>(nbr27, first, n, linear, pre, (int*)g->d_sched, (int*)g->d_blk);
	k_pair_heads<<<n_blocks, 256, 0, 0>>>(nbr27, first, n, linear, pre, partner, block_heads, totals);
	k_scan_blocks<<<1, 1024, 0, 0>>>(block_heads, n_blocks);
	HNS_HIP(hipMemcpy(h_totals, totals, sizeof(h_totals), hipMemcpyDeviceToHost));  // also the sync point for the launches above
	g->n_pairs = (uint64_t)h_totals[0];
	g->n_singles = (uint64_t)h_totals[1];
	k_write_pairs<<<n_blocks, 256, 0, 0>>>(nbr27, first, n, linear, pre, partner, block_heads, (int*)g->d_pairs);
	HNS_HIP(hipDeviceSynchronize());
	g->tiles_built = false;  // built when a solve first asks for the blocked form (most grids never do)
	g->sb_built = false;     // (likewise the block records of hns_sorblock.hip)
	g->n_tile_groups = g->n_tile_rest = 0;
	return HNS_OK;
}

// Tile groups of the blocked SOR kernel for the current wave records (see k_group_assign). Tables live in the grid's arena:
// d_tile_groups = n_tile_groups x kTileWaves record indices, d_tile_rest = the n_tile_rest records outside complete groups.
int hns_grid_build_tiles(hns_grid* g) {
	std::lock_guard<std::mutex> lock(g->build_mutex);  // cooks from several host threads may share the grid
	if (g->tiles_built) return HNS_OK;
	g->tiles_built = true;
	g->n_tile_groups = g->n_tile_rest = 0;
	const int np = (int)g->n_pairs, nl = (int)g->topo.n_leaves, w = kTileY * kTileZ;
	if (np == 0 || !g->d_tile_mem) return HNS_OK;
	int* head_of_leaf = (int*)g->d_tile_mem;             // nl
	int* leader = head_of_leaf + nl;                     // np
	int* members = leader + np;                          // np * w
	int* lead_flag = members + (size_t)np * w;           // np
	int* rest_flag = lead_flag + np;                     // np
	int* lead_list = rest_flag + np;                     // np
	const int nb = (np + 255) / 256;
	int* counts = lead_list + np;                        // nb
	int* totals = counts + nb;                           // 2
	HNS_HIP(hipMemsetAsync(head_of_leaf, 0xFF, sizeof(int) * (size_t)nl, 0));
	HNS_HIP(hipMemsetAsync(members, 0xFF, sizeof(int) * (size_t)np * w, 0));
	HNS_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(int), 0));
	const int* recs = (const int*)g->d_pairs;
	k_head_index<<<nb, 256, 0, 0>>>(recs, np, head_of_leaf);
	k_group_assign<<<nb, 256, 0, 0>>>(g->dev(), recs, np, head_of_leaf, kTileY, kTileZ, leader, members);
	k_group_flags<<<nb, 256, 0, 0>>>(leader, members, np, w, lead_flag, rest_flag);
	k_flag_counts<<<nb, 256, 0, 0>>>(lead_flag, np, counts);
	k_scan_blocks<<<1, 1024, 0, 0>>>(counts, nb);
	k_flag_write<<<nb, 256, 0, 0>>>(lead_flag, np, counts, lead_list, totals);
	k_flag_counts<<<nb, 256, 0, 0>>>(rest_flag, np, counts);
	k_scan_blocks<<<1, 1024, 0, 0>>>(counts, nb);
	k_flag_write<<<nb, 256, 0, 0>>>(rest_flag, np, counts, (int*)g->d_tile_rest, totals + 1);
	int h[2] = {0, 0};
	HNS_HIP(hipMemcpy(h, totals, sizeof(h), hipMemcpyDeviceToHost));
	g->n_tile_groups = (uint64_t)h[0];
	g->n_tile_rest = (uint64_t)h[1];
	if (h[0]) k_group_write<<<(h[0] * w + 255) / 256, 256, 0, 0>>>(lead_list, h[0], members, w, (int*)g->d_tile_groups);
	HNS_HIP(hipDeviceSynchronize());
	return HNS_OK;
}

// Device tables from the host origin list (g->topo.origins / hash_mask prepared by Topology::prepare). One allocation
// from the arena pool (hns_api.hip) holds all of them: a cook that rebuilds the grid reuses the previous grid's memory.
int hns_grid_upload(hns_grid* g) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
		set_error("hns_grid: no HIP device available (libhns has no CPU fallback; pass HNS_GRID_HOST_ONLY for topology-only use)");
		return HNS_ERR_NO_DEVICE;
	}
	HNS_HIP(hipGetDevice(&g->device));
	Topology& t = g->topo;
	const size_t nl = (size_t)(t.n_leaves > 0 ? t.n_leaves : 1);
	const size_t hash_size = (size_t)t.hash_mask + 1;
	const size_t n_blocks = (nl + 255) / 256;
	auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
	const size_t tw = (size_t)kTileY * kTileZ;
	const size_t sz[10] = {pad(16 * nl),                        // origins (int4)
	                       pad(4 * 27 * nl),                    // nbr27
	                       pad(4 * (hash_size + 1)),            // hash + the duplicate-origin status word
	                       pad(4 * nl),                         // sched
	                       pad(4 * 28 * nl),                    // blk records
	                       pad(4 * 56 * nl),                    // wave records: at most one wave per leaf
	                       pad(4 * (nl + n_blocks + 2)),        // schedule-build scratch
	                       pad(4 * nl),                         // tile groups: every record in at most one group
	                       pad(4 * nl),                         // records outside complete groups
	                       pad(4 * (nl * (5 + tw) + n_blocks + 2))};  // scratch of the builds on first use (hns_grid_build_tiles, hns_grid_build_blocks)
	size_t total = 0;
	for (size_t s : sz) total += s;
	HNS_TRY_RC(hns_arena_get(total, g->device, &g->d_arena, &g->arena_bytes));
	char* q = (char*)g->d_arena;
	void** slot[10] = {&g->d_origins, &g->d_nbr27, &g->d_hash, &g->d_sched_mem, &g->d_blk, &g->d_pairs, &g->d_scratch, &g->d_tile_groups, &g->d_tile_rest, &g->d_tile_mem};
	for (int i = 0; i < 10; ++i) {
		*slot[i] = q;
		q += sz[i];
	}
	int* status = (int*)g->d_hash + hash_size;
	g->on_device = true;
	HNS_HIP(hipMemsetAsync(g->d_hash, 0xFF, sizeof(int32_t) * hash_size, 0));
	if (t.n_leaves > 0) {
		const int n = (int)t.n_leaves;
		const int int_max = INT32_MAX;
		HNS_HIP(hipMemcpy(g->d_origins, t.origins.data(), sizeof(int32_t) * 4 * nl, hipMemcpyHostToDevice));
		HNS_HIP(hipMemcpy(status, &int_max, sizeof(int), hipMemcpyHostToDevice));
		k_hash_insert<<<(n + 255) / 256, 256, 0, 0>>>((const int4*)g->d_origins, n, (int*)g->d_hash, t.hash_mask, status);
		int dup = 0;
		HNS_HIP(hipMemcpy(&dup, status, sizeof(int), hipMemcpyDeviceToHost));
		if (dup != INT32_MAX) {
			// error path only: the host builder walks the leaves in order and words the message (which two leaves collide)
			const int rc = t.build_tables();
			if (rc != HNS_OK) return rc;
			set_error("hns_grid: duplicate leaf origin reported by the device build near leaf %d", dup);
			return HNS_ERR_TOPOLOGY;
		}
		k_build_nbr27<<<(unsigned)(((int64_t)n * 27 + 255) / 256), 256, 0, 0>>>(g->dev(), (int*)g->d_nbr27);
		HNS_HIP(hipGetLastError());
	}
	return hns_grid_upload_schedule(g);
}

// Host copies of the device-built tables, fetched the first time a host query needs them.
int hns_grid_host_tables(const hns_grid* cg) {
	hns_grid* g = const_cast<hns_grid*>(cg);
	std::lock_guard<std::mutex> lock(g->host_mutex);
	Topology& t = g->topo;
	if (t.have_tables) return HNS_OK;
	if (!g->on_device) return t.build_tables();
	t.nbr27.resize((size_t)t.n_leaves * 27);
	t.hash.resize((size_t)t.hash_mask + 1);
	if (t.n_leaves > 0) HNS_HIP(hipMemcpy(t.nbr27.data(), g->d_nbr27, sizeof(int32_t) * t.nbr27.size(), hipMemcpyDeviceToHost));
	HNS_HIP(hipMemcpy(t.hash.data(), g->d_hash, sizeof(int32_t) * t.hash.size(), hipMemcpyDeviceToHost));
	t.have_tables = true;
	return HNS_OK;
}

void hns_grid_free_device(hns_grid* g) {
	if (!g) return;
	(void)hns_grid_release_cache(g);
	if (g->d_sb_tab) hns_arena_put(g->d_sb_tab, g->sb_bytes, g->device);
	hns_grid_retire_blocks(g);
	g->d_sb_tab = nullptr;
	g->sb_bytes = 0;
	g->n_sb = 0;
	g->sb_built = false;
	hns_arena_put(g->d_arena, g->arena_bytes, g->device);
	g->d_arena = nullptr;
	g->arena_bytes = 0;
	g->d_origins = g->d_nbr27 = g->d_hash = g->d_sched = g->d_sched_mem = g->d_blk = g->d_pairs = g->d_scratch = nullptr;
	g->d_tile_groups = g->d_tile_rest = g->d_tile_mem = nullptr;
	g->n_tile_groups = g->n_tile_rest = 0;
	g->on_device = false;
}

// Debug/test access: the tile groups of the blocked SOR kernel (record indices into the wave records)
extern "C" int hns_grid_tile_tables(const hns_grid* g, int32_t* groups, int32_t* rest, uint64_t* n_groups, uint64_t* n_rest, int* tile_y, int* tile_z) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_tile_tables: null grid");
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_grid_tile_tables: grid has no device tables (HNS_GRID_HOST_ONLY)");
	if (int rc = hns_grid_build_tiles(const_cast<hns_grid*>(g))) return rc;
	if (n_groups) *n_groups = g->n_tile_groups;
	if (n_rest) *n_rest = g->n_tile_rest;
	if (tile_y) *tile_y = kTileY;
	if (tile_z) *tile_z = kTileZ;
	if (groups && g->n_tile_groups) HNS_HIP(hipMemcpy(groups, g->d_tile_groups, sizeof(int32_t) * kTileY * kTileZ * g->n_tile_groups, hipMemcpyDeviceToHost));
	if (rest && g->n_tile_rest) HNS_HIP(hipMemcpy(rest, g->d_tile_rest, sizeof(int32_t) * g->n_tile_rest, hipMemcpyDeviceToHost));
	return HNS_OK;
}

// Debug/test access: copies of the launch-order tables (sched: n_active ints, may be null pointers to skip).
extern "C" int hns_grid_launch_tables(const hns_grid* g, int32_t* sched, int32_t* wave_records, uint64_t* n_waves, uint64_t* n_lone) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_launch_tables: null grid");
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_grid_launch_tables: grid has no device tables (HNS_GRID_HOST_ONLY)");
	if (n_waves) *n_waves = g->n_pairs;
	if (n_lone) *n_lone = g->n_singles;
	if (sched && g->n_active) {
		if (g->d_sched)
			HNS_HIP(hipMemcpy(sched, g->d_sched, sizeof(int32_t) * g->n_active, hipMemcpyDeviceToHost));
		else
			for (uint64_t b = 0; b < g->n_active; ++b) sched[b] = (int32_t)(g->first_active + b);
	}
	if (wave_records && g->n_pairs) HNS_HIP(hipMemcpy(wave_records, g->d_pairs, sizeof(int32_t) * 56 * g->n_pairs, hipMemcpyDeviceToHost));
	return HNS_OK;
}
