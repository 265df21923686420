// hns_internal.hpp -- shared declarations of libhns.so (not part of the public ABI; see include/hns.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "hns.h"

namespace hns {

// ---- error plumbing -------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
inline int fail(int code, const char* msg) {
	set_error("%s", msg);
	return code;
}

// ---- options (hns_set_option) -------------------------------------------------------------------------------------
// Alternative kernel forms and data-movement strategies kept for A/B measurement and as cross-checks of the default one.
// Every entry point reads the current value when it is called, so a test or benchmark can switch forms between calls.
enum { kRbgsAuto = 0, kRbgsColor = 1 };
enum { kScheduleAuto = 0, kScheduleLinear = 1 };
struct Options {
	std::atomic<int> rbgs{kRbgsAuto};          // "rbgs": auto | color (the reference's two launches per iteration: the independent cross-check)
	std::atomic<int> advect_generic{0};        // "advect": auto | generic (64-bit addressed kernels)
	std::atomic<int> stencil_block{0};         // "stencil": auto | block (512-thread divergence / gradient)
	std::atomic<int> schedule{kScheduleAuto};  // "schedule": auto | linear (read when launch tables are built)
	std::atomic<int> cook_cache{1};            // "cook_cache": operator calls keep their device buffers with the grid
	std::atomic<int> cook_pipeline{1};         // "cook_pipeline": hns_compute_sim overlaps transfers with the substep
	std::atomic<int> divergence_form{0};       // "divergence": 0 auto (by size) | 1 row | 2 coalesced (own leaf fetched in memory order, handed to the row owners through LDS) | 3 zpair (coalesced, two z-adjacent leaves per workgroup)
	std::atomic<int> fuse_pointwise{1};        // "fuse": hns_sim_substep / hns_compute_sim without a collision field run divergence + combustion + buoyancy as one launch and advect the four combustion fields out of one 16-byte-per-voxel array
	std::atomic<int> sor_block_lb{0};          // "sor_block_lb": block edge of the temporally blocked SOR in leaves, 0 = by size | 1 | 2 (the tests' way to every kernel on every grid)
	std::atomic<int> dist_wire_us{0};          // "dist_wire_us": loopback transport only, emulated time on the wire per exchange
	std::atomic<int> dist_mirror{1};           // "dist_mirror": 1 | 0 | guarded -- over the ipc / local transports a rank of 16^3 blocks with sweeps_per_exchange = 2 runs the CHAINED substep (every kernel delivers its own halo); 0 = the exchanged substep (what RCCL ranks run)
	std::atomic<int> dist_unsplit{1};          // "dist_unsplit": small ranks of the exchanged substep run their short phases as ONE launch over the owned leaves with the exchange behind it on the compute stream; 0 = boundary / interior split on two streams at every size
};
Options& options();

// ---- host topology ----------------------------------------------------------------------------------------------
// Replaces the NanoGrid<ValueOnIndex> tree walk (reference Stencils.hpp:51-71 -> NanoVDB.h:5549) with flat tables:
//   origins[l]      leaf origin (8-aligned), l = position of the leaf in the caller's coordinate array
//   nbr27[l][27]    leaf index of the 27 surrounding leaves ((dx+1)*9+(dy+1)*3+(dz+1)), -1 = absent
//   hash            open-addressing map origin -> leaf for taps further than one leaf away
struct Topology {
	std::vector<int32_t> origins;  // n_leaves * 4 (x, y, z, pad) -- int4 on the device
	std::vector<int32_t> nbr27;    // n_leaves * 27
	std::vector<int32_t> hash;     // table_size entries, -1 = empty
	uint32_t hash_mask = 0;
	int64_t n_leaves = 0;
	bool have_tables = false;  // nbr27 / hash present on the host (device-built grids fetch them on first host query)

	// Both return HNS_OK or HNS_ERR_TOPOLOGY (message set).
	int prepare(const int32_t* leaf_origins_xyz, int64_t n_leaves);  // origins + hash size; checks alignment and the leaf limit
	int build_tables();                                             // host build of hash + nbr27; detects duplicate origins
	int64_t find_leaf(int32_t ox, int32_t oy, int32_t oz) const;
	uint64_t offset(int32_t i, int32_t j, int32_t k) const;  // 1-based, 0 = outside
};

uint32_t hash_origin(int32_t x, int32_t y, int32_t z);

// Device view handed to every kernel by value.
struct GridDev {
	const int4* origins;
	const int* nbr27;
	const int* hash;
	const int* sched;  // block -> leaf order (XCD-chunked), n_active entries
	int sched_seg;     // >= 0: the order is the closed form sched_leaf(pos, n_active, sched_seg, sched_pre) (hns_device.hpp) -- one chunk per XCD (0) or
	int sched_pre;     // plain order (1): a few scalar instructions instead of a dependent load at the head of every workgroup; -1: look `sched` up
	const int* blk;    // per block, in launch order: {leaf, nbr27[27]} (28 ints)
	uint32_t hash_mask;
	int n_leaves;
	int n_active;
	int first;  // first active leaf: kernels update leaves [first, first + n_active)
	int oob;  // element read by advect_scalars for out-of-domain taps (0 on an unpartitioned grid)
	int rev;  // 1: walk the launch order backwards (rows of eight workgroups reversed: a kernel starts on the cached tail of what its predecessor wrote)
	int* far_flag;  // null, or (the local grid of a multi-GPU rank) a word the advection kernels raise when a tap leaves the 27-leaf neighbourhood of its
	                // leaf: the rank holds one layer of ghost leaves, further away it cannot tell "outside the domain" from "on another rank"
};

// workgroup -> position in the launch-order tables, honouring GridDev::rev
__device__ __forceinline__ unsigned launch_pos(const GridDev& g, unsigned b) {
	if (!g.rev) return b;
	const unsigned rows = (unsigned)g.n_active >> 3;
	return (b >> 3) < rows ? (((rows - 1u - (b >> 3)) << 3) | (b & 7u)) : b;
}

// the leaf workgroup b works on (kernels with one workgroup per leaf). One chunk per XCD (sched_seg 0: every grid up to 40k leaves) is
// closed form -- block b = 8 i + x is leaf x * (n / 8) + min(x, n % 8) + i of the range -- and costs a few scalar instructions; reading
// it from the table put a dependent load in front of everything a workgroup does (the advection kernels are bound by the length of
// that chain: profiles/r04_advect_notes.txt).
__device__ __forceinline__ int launch_leaf(const GridDev& g, unsigned b) {
	const int pos = (int)launch_pos(g, b);
	if (g.sched_seg == 0 && g.sched_pre == 0) {
		const int base = g.n_active >> 3, rem = g.n_active & 7, x = pos & 7, i = pos >> 3;
		return g.first + x * base + (x < rem ? x : rem) + i;
	}
	return g.sched ? g.sched[pos] : g.first + pos;
}

}  // namespace hns

struct hns_sim;

struct hns_grid {
	hns::Topology topo;
	float voxel_size = 1.0f;
	uint64_t n_active = 0;      // kernels update leaves [first_active, first_active + n_active) and only read the others
	uint64_t first_active = 0;
	uint64_t outside_element = 0;
	int* far_flag = nullptr;    // see GridDev::far_flag (set by hns_dist; not owned)
	bool on_device = false;
	int device = -1;
	// device copies
	void* d_origins = nullptr;
	void* d_nbr27 = nullptr;
	void* d_hash = nullptr;
	void* d_sched = nullptr;
	void* d_blk = nullptr;
	int sched_seg = -1, sched_pre = 0;  // parameters of the current launch order (hns_grid_upload_schedule), for GridDev
	uint64_t chain_boundary = 0;  // leaves at the head of the active range that are a multi-GPU rank's BOUNDARY leaves (hns_dist: the range its chained sweeps run over); correctness, not speed
	uint64_t sched_prefix = 0;    // leaves at the head of the active range that the launch order deals out to all XCDs first (hns_dist: boundary leaves)
	void* d_sched_mem = nullptr;  // storage of d_sched (d_sched itself is null under the linear schedule)
	void* d_scratch = nullptr;    // scratch of the block-record build (hns_grid_build_blocks)
	void* d_arena = nullptr;      // the one device allocation all of the above are slices of (arena pool, hns_api.hip)
	size_t arena_bytes = 0;
	// temporally blocked SOR kernel (hns_sorblock.hip): records of the 16^3-voxel blocks in launch order (64 leaves under each tile),
	// built on first use into an arena allocation of their own
	void* d_sb_tab = nullptr;
	size_t sb_bytes = 0;
	uint64_t n_sb = 0;
	bool sb_built = false;
	int sb_seg = 0;
	uint64_t sb_first = 0, sb_count = 0;                  // the launch range the records were built for
	std::vector<std::pair<void*, size_t>> sb_retired;     // superseded tables: back to the pool when the grid goes, or -- beyond four of them -- behind a device synchronise
	std::mutex build_mutex;              // guards the tables built on first use (block records): cooks from several host threads may share a grid
	std::mutex host_mutex;               // guards the lazy host copy of the device-built tables and sim_cache
	std::vector<hns_sim*> sim_cache;     // device-resident state kept between operator calls (hns_api.hip: make_sim)
	hns::GridDev dev() const;
};

// implemented in hns_pressure.hip: hns_dev_rbgs_iterate with the option of starting from p = 0 without reading (or clearing) p_a
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                                void* stream, bool from_zero);

// the sweeps of a multi-GPU rank (hns_flags.hpp: PhaseMirror, PackMirror)
namespace hns { struct PhaseMirror; struct PackMirror; }
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_pack_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                 const hns::PackMirror* m, void* stream, bool* done, int iterations);  // hns_sorblock.hip: the sweep (one or two iterations) that packs its own messages
extern "C" __attribute__((visibility("hidden"))) bool hns_rbgs_block_packable(hns_grid* g);  // hns_sorblock.hip: would hns_rbgs_block_pack_launch launch on this grid's range?
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_mirror_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                   const hns::PhaseMirror* m, void* stream, int iterations);  // hns_sorblock.hip: one or two iterations per chained launch
extern "C" __attribute__((visibility("hidden"))) int hns_chain_divergence(hns_grid* g, const float* vel3, float* div, float inv_dx, const hns::PhaseMirror* m, void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_chain_subtract_pressure_gradient(hns_grid* g, const float* vel3, const float* p, float* out3, float inv_dx,
                                                                                          const hns::PhaseMirror* m, void* stream);

// the fused middle of hns_sim_substep (round 6): divergence + combustion_oxygen + temperature_buoyancy in one launch (hns_pressure.hip), leaving {fuel, waste, temperature,
// flame} as one 16-byte element per voxel, and the advect_scalars launch that gathers its taps from that array (hns_advect.hip)
extern "C" __attribute__((visibility("hidden"))) int hns_divergence_combust_buoyancy(hns_grid* g, const float* vel3, float* div, float inv_dx, const float* fuel, const float* waste,
                                                                                      const float* temperature, const float* flame, float* q4, float* vel3_out, float temp_gain,
                                                                                      float expansion, float dt, float ambient, float strength, void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_advect_scalars_q4(hns_grid* g, const float* vel3, const float* q4, float* const* q4_out, const float* const* in,
                                                                            float* const* out, int n, float dt, float inv_dx, void* stream);
extern "C" __attribute__((visibility("hidden"))) bool hns_advect_q4_ok(const hns_grid* g);

// implemented in hns_pointwise.hip: combustion_oxygen split into its divergence update (needs fuel, waste) and the rest (hns_dist_*.hip: the boundary leaves' divergence first)
extern "C" __attribute__((visibility("hidden"))) int hns_combustion_div(const float* fuel, const float* waste, float* divergence, float expansion, uint64_t n,
                                                                        void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_combustion_fields(const float* fuel, const float* waste, const float* temperature, const float* flame,
                                                                           float* out_fuel, float* out_waste, float* out_temperature, float* out_flame,
                                                                           float temp_gain, uint64_t n, void* stream);

// implemented in hns_api.hip: process-wide pool of device allocations (simulation state and grid tables)
extern "C" __attribute__((visibility("hidden"))) int hns_arena_get(size_t need, int device, void** p, size_t* bytes);
extern "C" __attribute__((visibility("hidden"))) void hns_arena_put(void* p, size_t bytes, int device);

// implemented in hns_gridbuild.hip
int hns_grid_upload(hns_grid* g);           // device build of every table from topo.origins
void hns_grid_free_device(hns_grid* g);
int hns_grid_upload_schedule(hns_grid* g);  // launch-order tables for the current n_active
// implemented in hns_sorblock.hip: the temporally blocked SOR form (k iterations per launch)
int hns_grid_build_blocks(hns_grid* g);     // records of the 16^3-voxel blocks of the grid's launch range
void hns_grid_retire_blocks(hns_grid* g);   // superseded records back to the pool (grid destruction / rebuild only)
int hns_rbgs_block_shape(hns_grid* g, int* k_max);  // block edge in leaves this grid is swept with (0: not by this form) and the iterations per launch
bool hns_rbgs_block_lean(hns_grid* g, int lb, int k);  // is that launch the lean form of the kernel (row state in LDS, three workgroups per CU)?
int hns_rbgs_block_launch(hns_grid* g, int lb, int k, bool src_is_zero, const float* div, const float* src, float* dst, float dx2, float omega, void* stream);
int hns_grid_host_tables(const hns_grid* g);  // make topo.nbr27 / topo.hash valid on the host
