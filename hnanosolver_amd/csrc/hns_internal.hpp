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
enum { kRbgsAuto = 0, kRbgsColor = 1, kRbgsWave = 2, kRbgsPair = 3, kRbgsTile = 4, kRbgsBlock = 5 };
enum { kScheduleAuto = 0, kScheduleLinear = 1, kScheduleChunk = 2 };
struct Options {
	std::atomic<int> rbgs{kRbgsAuto};          // "rbgs": auto | color | wave | pair | tile | block
	std::atomic<int> advect_generic{0};        // "advect": auto | generic (64-bit addressed kernels)
	std::atomic<int> stencil_block{0};         // "stencil": auto | block (512-thread divergence / gradient)
	std::atomic<int> schedule{kScheduleAuto};  // "schedule": auto | linear | chunk (read when launch tables are built)
	std::atomic<int> schedule_segment{0};      // "schedule_segment": leaves per XCD segment of the launch order, 0 = by size
	std::atomic<int> alternate{1};             // "alternate": odd SOR sweeps walk the records backwards
	std::atomic<int> rev{1};                   // "rev": divergence / advect_scalars walk the leaves backwards
	std::atomic<int> cook_cache{1};            // "cook_cache": operator calls keep their device buffers with the grid
	std::atomic<int> cook_pipeline{1};         // "cook_pipeline": hns_compute_sim overlaps transfers with the substep
	std::atomic<int> dist_wire_us{0};          // "dist_wire_us": loopback transport only, emulated time on the wire per exchange
	std::atomic<int> dist_mirror{1};           // "dist_mirror": multi-GPU pressure loop with sweeps_per_exchange = 1 over the ipc / local transport delivers its halo inside the sweep kernel
	std::atomic<int> dist_chain{1};            // "dist_chain": with dist_mirror, EVERY kernel of the substep of such a rank is one launch that delivers its own halo (read when the ranks connect)
	std::atomic<int> divergence_form{0};       // "divergence": 0 auto (by size) | 1 row | 2 coalesced (own leaf fetched in memory order, handed to the row owners through LDS)
	std::atomic<int> dist_pipeline{1};         // "dist_pipeline": between blocks of the exchanged pressure loop the compute stream waits for the boundary KERNEL of the posted exchange only, not for its messages (hns_dist.hip: complete_boundary_only)
	std::atomic<int> dist_unsplit{1};          // "dist_unsplit": the exchanged pressure loop sweeps ALL owned leaves in one launch that packs its own messages, then exchanges and unpacks on the same stream (no boundary / interior split, no events)
	std::atomic<int> dist_pack{1};             // "dist_pack": the blocked boundary sweep of an exchanged pressure loop writes the peers' messages itself (no pack launch per exchange)
	std::atomic<int> dist_block{1};            // "dist_block": a rank with sweeps_per_exchange >= 2 sweeps its launch ranges two iterations per launch (hns_sorblock.hip over a range)
	std::atomic<int> dist_spread{1};           // "dist_spread": the owned launch range of a sweeps_per_exchange = 1 rank deals its boundary leaves out to all XCDs (read at hns_dist_create)
	std::atomic<int> sor_block_lb{0};          // "sor_block_lb": temporally blocked SOR (hns_sorblock.hip), block edge in leaves: 0 = by size, 1, 2
	std::atomic<int> sor_block_k{0};           // "sor_block_k": ... iterations per launch: 0 = by shape, 2, 4 (4: one-leaf blocks only)
	std::atomic<int> sor_block_stagger{8};     // "sor_block_stagger": ... launch-start stagger of the two workgroups of a CU, x 1,024 cycles (0 = off; hns_sorblock.hip)
	std::atomic<int> sor_block_lean{0};        // "sor_block_lean" = auto | 0 | 1 | dma | xy (stored 0 .. 4): ... its lean forms (row state in LDS, three workgroups per CU); xy = auto: the sweep threads fetch their own rows; 1: waves sorted by parity; dma: 1 with div through LDS-DMA
	std::atomic<int> fuse_pointwise{1};        // "fuse": hns_sim_substep / hns_compute_sim without a collision field run divergence + combustion + buoyancy as one launch and advect the four combustion fields out of one 16-byte-per-voxel array
	std::atomic<int> sor_block_seg{0};         // "sor_block_seg": ... blocks per XCD segment of its launch order (0: one chunk per XCD; read when the block table is built)
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
	int rev;  // 1: walk the launch order backwards (rows of eight workgroups reversed, see k_rbgs_pair)
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

namespace hns {
#ifndef HNS_TILE_Y  // (overridable for A/B builds: profiles/micro/exp/build.sh)
#define HNS_TILE_Y 2
#define HNS_TILE_Z 2
#endif
constexpr int kTileY = HNS_TILE_Y, kTileZ = HNS_TILE_Z;  // wave records per workgroup of the blocked SOR kernel: y x z (powers of two)
}

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
	void* d_pairs = nullptr;    // launch-ordered wave records {leaf0, nbr27, leaf1 or -1, nbr27} (56 ints): z-adjacent pairs and lone leaves
	int sched_seg = -1, sched_pre = 0;  // parameters of the current launch order (hns_grid_upload_schedule), for GridDev
	uint64_t chain_boundary = 0;  // leaves at the head of the active range that are a multi-GPU rank's BOUNDARY leaves (hns_dist: the range its chained sweeps run over); correctness, not speed
	uint64_t sched_prefix = 0;    // leaves at the head of the active range that the launch order deals out to all XCDs first (hns_dist: boundary leaves)
	void* d_sched_mem = nullptr;  // storage of d_sched (d_sched itself is null under the linear schedule)
	void* d_scratch = nullptr;    // schedule-build scratch
	void* d_arena = nullptr;      // the one device allocation all of the above are slices of (arena pool, hns_api.hip)
	size_t arena_bytes = 0;
	uint64_t n_pairs = 0, n_singles = 0;  // waves to launch / how many of them carry a lone leaf
	// blocked SOR kernel: complete kTileY x kTileZ groups of wave records (record indices, kTileY*kTileZ per group) and the
	// records outside them; built by hns_grid_build_tiles
	void* d_tile_groups = nullptr;
	void* d_tile_rest = nullptr;
	void* d_tile_mem = nullptr;
	uint64_t n_tile_groups = 0, n_tile_rest = 0;
	bool tiles_built = false;  // d_tile_* are filled on first use (hns_grid_build_tiles)
	// temporally blocked SOR kernel (hns_sorblock.hip): records of the 16^3-voxel blocks in launch order (64 leaves under each tile),
	// built on first use into an arena allocation of their own
	void* d_sb_tab = nullptr;
	size_t sb_bytes = 0;
	uint64_t n_sb = 0;
	bool sb_built = false;
	int sb_seg = 0;
	uint64_t sb_first = 0, sb_count = 0;                  // the launch range the records were built for
	std::vector<std::pair<void*, size_t>> sb_retired;     // superseded tables: back to the pool when the grid goes, or -- beyond four of them -- behind a device synchronise
	std::mutex build_mutex;              // guards the tables built on first use (tile groups, block records): cooks from several host threads may share a grid
	std::mutex host_mutex;               // guards the lazy host copy of the device-built tables and sim_cache
	std::vector<hns_sim*> sim_cache;     // device-resident state kept between operator calls (hns_api.hip: make_sim)
	hns::GridDev dev() const;
};

// implemented in hns_pressure.hip: hns_dev_rbgs_iterate with the option of starting from p = 0 without reading (or clearing) p_a
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_iterate(hns_grid* g, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                                void* stream, bool from_zero);

// implemented in hns_pressure.hip: one sweep of a multi-GPU rank that mirrors its boundary rows into the peers' ghost voxels
// itself (hns_flags.hpp: PhaseMirror), and the number of wave records of `g` that touch a local leaf below n_boundary
namespace hns { struct PhaseMirror; struct PackMirror; }
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_pack_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                 const hns::PackMirror* m, void* stream, bool* done);  // hns_sorblock.hip: the boundary sweep that packs its own messages
extern "C" __attribute__((visibility("hidden"))) bool hns_rbgs_block_packable(hns_grid* g);  // hns_sorblock.hip: would hns_rbgs_block_pack_launch launch on this grid's range?
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_mirror_sweep(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                           const hns::PhaseMirror* m, void* stream, bool backwards);
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_block_mirror_launch(hns_grid* g, const float* div, const float* src, float* dst, float dx, float omega, bool src_is_zero,
                                                                                   const hns::PhaseMirror* m, void* stream);  // hns_sorblock.hip: two iterations per chained launch
extern "C" __attribute__((visibility("hidden"))) int hns_chain_divergence(hns_grid* g, const float* vel3, float* div, float inv_dx, const hns::PhaseMirror* m, void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_chain_subtract_pressure_gradient(hns_grid* g, const float* vel3, const float* p, float* out3, float inv_dx,
                                                                                          const hns::PhaseMirror* m, void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_rbgs_count_boundary_records(hns_grid* g, int n_boundary, unsigned* d_scratch, unsigned* out2, void* stream);

// the fused middle of hns_sim_substep (round 6): divergence + combustion_oxygen + temperature_buoyancy in one launch (hns_pressure.hip), leaving {fuel, waste, temperature,
// flame} as one 16-byte element per voxel, and the advect_scalars launch that gathers its taps from that array (hns_advect.hip)
extern "C" __attribute__((visibility("hidden"))) int hns_divergence_combust_buoyancy(hns_grid* g, const float* vel3, float* div, float inv_dx, const float* fuel, const float* waste,
                                                                                      const float* temperature, const float* flame, float* q4, float* vel3_out, float temp_gain,
                                                                                      float expansion, float dt, float ambient, float strength, void* stream);
extern "C" __attribute__((visibility("hidden"))) int hns_advect_scalars_q4(hns_grid* g, const float* vel3, const float* q4, float* const* q4_out, const float* const* in,
                                                                            float* const* out, int n, float dt, float inv_dx, void* stream);
extern "C" __attribute__((visibility("hidden"))) bool hns_advect_q4_ok(const hns_grid* g);

// implemented in hns_pointwise.hip: combustion_oxygen split into its divergence update (needs fuel, waste) and the rest (hns_dist.hip: the boundary leaves' divergence first)
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
int hns_grid_build_tiles(hns_grid* g);      // tile groups of the blocked SOR kernel for the current wave records
// implemented in hns_sorblock.hip: the temporally blocked SOR form (k iterations per launch)
int hns_grid_build_blocks(hns_grid* g);     // records of the 16^3-voxel blocks of the grid's launch range
void hns_grid_retire_blocks(hns_grid* g);   // superseded records back to the pool (grid destruction / rebuild only)
int hns_rbgs_block_shape(hns_grid* g, int* k_max);  // block edge in leaves this grid is swept with (0: not by this form) and the iterations per launch
bool hns_rbgs_block_lean(hns_grid* g, int lb, int k);  // is that launch the lean form of the kernel (row state in LDS, three workgroups per CU)?
int hns_rbgs_block_launch(hns_grid* g, int lb, int k, bool src_is_zero, const float* div, const float* src, float* dst, float dx2, float omega, void* stream);
int hns_grid_host_tables(const hns_grid* g);  // make topo.nbr27 / topo.hash valid on the host
