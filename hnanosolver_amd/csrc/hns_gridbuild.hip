// hns_gridbuild.hip -- index-grid construction on the device.
//
// The reference rebuilds its NanoGrid<ValueOnIndex> on the GPU every cook (create_index_grid, reference
// src/Cuda/HNanoSolver.cu:375-384 -> externals/nanovdb/tools/cuda/PointsToGrid.cuh:511-1064: ~25 launches and 8 CUB
// sorts/scans over all N voxel coordinates). Here only the leaf origins (one coordinate in 512) cross PCIe; the origin
// hash, the 27-neighbour table and the launch-ordered {leaf, nbr27} records the kernels read are built by three small kernels
// over the leaves (the block records of the SOR kernel: hns_sorblock.hip, on first use). Integer work only; tests/test_gridbuild_gpu.py checks every table against the host builder of
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

}  // namespace hns

using namespace hns;

// Launch order tables for the current n_active: d_sched, d_blk. Called at build time and whenever hns_grid_set_active_leaves / _range changes the
// launch range. Both are slices of the grid's one device allocation (hns_grid_upload), sized for n_active = n_leaves, so nothing is allocated here.
int hns_grid_upload_schedule(hns_grid* g) {
	const int n = (int)g->n_active, first = (int)g->first_active;
	g->sb_built = false;  // (the block records of hns_sorblock.hip follow the launch range: rebuilt on next use)
	if (n == 0) return HNS_OK;
	// One chunk per XCD wins by a wide margin while the sweep arrays fit the Infinity Cache and its neighbourhood (256^3: 40 vs
	// 51 us per sweep against plain leaf order). Beyond it the eight XCDs had better walk through neighbouring stretches of
	// memory together: segments of 128 leaves (1024^3-extent plume, 66k leaves: 127 -> 116 us; 512^3: 441 -> 428; plain order
	// gives 441 / 131). Option "schedule" = linear: plain leaf order (A/B, tests).
	const int seg = options().schedule.load() == kScheduleLinear ? 1 : (n > 40000 ? 128 : 0);
	const int pre = (int)std::min<uint64_t>(g->sched_prefix, (uint64_t)n) & ~7;
	g->d_sched = (seg == 1 && pre == 0) ? nullptr : g->d_sched_mem;
	g->sched_seg = seg, g->sched_pre = pre;
	k_build_blk<<<(unsigned)(((int64_t)n * 28 + 255) / 256), 256, 0, 0>>>((const int*)g->d_nbr27, first, n, seg, pre, (int*)g->d_sched, (int*)g->d_blk);
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
	auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
	const size_t sz[6] = {pad(16 * nl),              // origins (int4)
	                      pad(4 * 27 * nl),          // nbr27
	                      pad(4 * (hash_size + 1)),  // hash + the duplicate-origin status word
	                      pad(4 * nl),               // sched
	                      pad(4 * 28 * nl),          // blk records
	                      pad(4 * (2 * nl + 2))};    // scratch of the block-record build on first use (hns_grid_build_blocks: flag[n] | leaders[n] | total[2])
	size_t total = 0;
	for (size_t s : sz) total += s;
	HNS_TRY_RC(hns_arena_get(total, g->device, &g->d_arena, &g->arena_bytes));
	char* q = (char*)g->d_arena;
	void** slot[6] = {&g->d_origins, &g->d_nbr27, &g->d_hash, &g->d_sched_mem, &g->d_blk, &g->d_scratch};
	for (int i = 0; i < 6; ++i) {
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
	g->d_origins = g->d_nbr27 = g->d_hash = g->d_sched = g->d_sched_mem = g->d_blk = g->d_scratch = nullptr;
	g->on_device = false;
}

// Debug/test access: copy of the launch order (sched: n_active leaf ids in workgroup order).
extern "C" int hns_grid_launch_tables(const hns_grid* g, int32_t* sched) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_launch_tables: null grid");
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_grid_launch_tables: grid has no device tables (HNS_GRID_HOST_ONLY)");
	if (sched && g->n_active) {
		if (g->d_sched)
			HNS_HIP(hipMemcpy(sched, g->d_sched, sizeof(int32_t) * g->n_active, hipMemcpyDeviceToHost));
		else
			for (uint64_t b = 0; b < g->n_active; ++b) sched[b] = (int32_t)(g->first_active + b);
	}
	return HNS_OK;
}
