/*
 * hns.h -- C ABI of libhns.so: the MI355X-native (HIP, gfx950) implementation of HNanoSolver's per-substep hot path.
 *
 * This is the drop-in boundary. Each entry point replaces one of the reference's `extern "C"` C++-typed functions
 * (paths relative to the reference checkout), keeping argument meaning, synchronous in-place semantics and refusal
 * conditions, but with plain pointers/sizes and int return codes instead of C++ types and exceptions:
 *
 *   hns_grid_create                 <- CreateIndexGrid          src/Cuda/HNanoSolver.cu:375-390   (decl. src/SOP/HNanoSolver/SOP_HNanoSolver.hpp:82)
 *   hns_compute_sim                 <- Compute_Sim              src/Cuda/HNanoSolver.cu:9-372,393-396 (decl. src/SOP/HNanoSolver/SOP_HNanoSolver.hpp:84-85)
 *   hns_advect_index_grid           <- AdvectIndexGrid          src/Cuda/Advection.cu:13-112,169-171  (decl. src/SOP/Advection/SOP_VDBAdvect.hpp:66)
 *   hns_advect_index_grid_velocity  <- AdvectIndexGridVelocity  src/Cuda/Advection.cu:114-166,173-175 (decl. src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.hpp:60)
 *   hns_project_non_divergent       <- ProjectNonDivergent      src/Cuda/PressureProjection.cu:9-78,132-135 (decl. src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp:69)
 *   hns_divergence                  <- Divergence               src/Cuda/PressureProjection.cu:81-129       (decl. src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp:70)
 *
 * The type crossing the boundary in the reference is HNS::GridIndexedData (src/Utils/GridData.hpp:16-166): a
 * coordinate array plus named float / Vec3f blocks in insertion order. Here it is `hns_field[]` (same order) and the
 * `hns_grid` handle built from the coordinate array. All host arrays are caller-owned and overwritten in place.
 *
 * Data layout (identical to the reference's): flat leaf-dense arrays, element = leaf*512 + (x<<6 | y<<3 | z) where
 * leaf l is the l-th block of 512 coordinates handed to hns_grid_create (src/Utils/GridBuilder.hpp:156-166);
 * Vec3f fields are AoS float[3] on the host. All arithmetic is float32.
 *
 * `stream` is a hipStream_t passed as void* (NULL = the default stream). Every host-pointer entry point is
 * synchronous: it returns after the stream has drained, like the reference's cudaStreamSynchronize at the end of
 * each driver (HNanoSolver.cu:371, PressureProjection.cu:72, Advection.cu:99-103,158).
 *
 * Threading: hns_last_error() is thread-local, and one grid may be used by several host threads at once (Houdini cooks verbs
 * concurrently): each operator call works on its own device buffers -- the set kept with the grid if it is free, a private one
 * otherwise. Process-wide state, unlike the reference's entry points (which have none): the options of hns_set_option (atomics,
 * read by every entry point when it is called -- switch them while no call is in flight) and the pool of idle device
 * allocations (hns_trim_memory; internally locked).
 *
 * There is no CPU fallback: without a usable HIP device every compute entry point fails with HNS_ERR_NO_DEVICE.
 */
#ifndef HNS_H
#define HNS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNS_VERSION 100

/* Return codes. Negative = failure; hns_last_error() holds the message (thread-local). */
#define HNS_OK 0
#define HNS_ERR_INVALID_ARGUMENT (-1) /* the reference throws std::invalid_argument (HNanoSolver.cu:12-23) */
#define HNS_ERR_RUNTIME (-2)          /* the reference throws std::runtime_error  (HNanoSolver.cu:44,62,196; Advection.cu:20,28) */
#define HNS_ERR_HIP (-3)              /* a HIP runtime call failed (the reference's CUDA_CHECK, Utils.cuh:10-18) */
#define HNS_ERR_NO_DEVICE (-4)        /* no HIP device / host-only grid */
#define HNS_ERR_TOPOLOGY (-5)         /* coordinates are not leaf-dense 8^3 blocks, or a leaf appears twice */

const char* hns_last_error(void);
int hns_version(void);
int hns_device_count(void); /* 0 when no HIP device is visible; never initialises a device context */
/* Device memory of destroyed simulation state and grid tables is kept in a small process-wide pool (at most six idle
 * allocations; the device is idle when memory enters it) so that the next cook, typically on a slightly different
 * topology, does not pay hipMalloc/hipFree again; this returns it to the driver. */
int hns_trim_memory(void);
/* Alternative kernel forms and data-movement strategies, kept as cross-checks of the default ones and for A/B measurement (all of them produce the same bits:
 * tests/test_kernel_variants_gpu.py). Process-wide, read by every entry point when it is called. value = NULL restores the default. Twelve names (round 6; the
 * twenty-four of round 5 -- five more SOR forms among them -- are history: DESIGN_HISTORY.md, profiles/micro/exp/):
 *   "rbgs"          auto | color. auto: temporally blocked red-black SOR (hns_sorblock.hip: two iterations per launch on 16^3-voxel blocks, two or four on one-leaf
 *                   blocks for grids of up to 600 leaves). color: the reference's own decomposition, two launches per iteration in place -- the independent cross-check
 *   "sor_block_lb"  0 = by size | 1 | 2: block edge of the temporally blocked form in leaves, whatever the size of the grid (how the tests reach both kernels on every leaf set)
 *   "advect"        auto | generic (64-bit addressed advection kernels)
 *   "stencil"       auto | block (512-thread divergence and gradient kernels)
 *   "divergence"    auto | row | coalesced | zpair: the divergence kernel fetches its own leaf row by row, or in memory order with a hand-over through LDS, or that with
 *                   two z-adjacent leaves per workgroup handing each other their common z face (auto: the last from 16,384 leaves, the first below)
 *   "schedule"      auto | linear (workgroup -> leaf order: one chunk of the leaf list per XCD, 128-leaf segments beyond 40,000 leaves | plain leaf order; takes
 *                   effect when a grid's launch tables are next built)
 *   "fuse"          1 | 0 (hns_sim_substep / hns_compute_sim without a collision field: divergence + combustion_oxygen + temperature_buoyancy as ONE launch that leaves
 *                   {fuel, waste, temperature, flame} as one 16-byte element per voxel, which advect_scalars then gathers its taps from; 0 = the reference's three
 *                   launches over five float arrays)
 *   "cook_cache"    1 | 0 (operator calls keep their device buffers with the grid)
 *   "cook_pipeline" 1 | 0 (hns_compute_sim overlaps its transfers with the substep)
 *   "dist_mirror"   1 | 0 | guarded: over the ipc or local transport a rank whose owned range is swept in 16^3 blocks, created with sweeps_per_exchange = 2, runs the
 *                   CHAINED substep -- every kernel stores the voxels its peers read into their ghost voxels itself, no exchanges (read when the ranks connect; all
 *                   ranks must agree). 0 = the exchanged substep, which is what RCCL ranks run. guarded = one wave waits for the peers in front of every chained launch
 *                   instead of the boundary workgroups inside it (ranks sharing one GPU)
 *   "dist_unsplit"  1 | 0: a rank of the exchanged substep with up to 16,384 owned leaves runs the sweeps of its pressure loop, its divergence and its gradient subtraction
 *                   as ONE launch over all owned leaves with pack / transfer / unpack behind it on the compute stream; 0 = boundary leaves on a communication stream beside
 *                   the interior launch at every size (what larger ranks always do)
 *   "dist_wire_us"  N: the loopback transport of hns_dist holds every exchange N microseconds (emulated wire time) */
int hns_set_option(const char* name, const char* value);
const char* hns_get_option(const char* name); /* current value as a word; NULL for an unknown name */

/* ------------------------------------------------------------------------------------------------------------ */
/* Index grid (topology)                                                                                         */
/* ------------------------------------------------------------------------------------------------------------ */

typedef struct hns_grid hns_grid;

#define HNS_GRID_DEFAULT 0u
#define HNS_GRID_HOST_ONLY 1u     /* build the host tables only (no device upload); compute calls then fail */
#define HNS_GRID_SKIP_VALIDATE 2u /* trust that coords are leaf-dense; only every 512th coordinate is read */

/* Replaces CreateIndexGrid. coords_xyz = n_voxels x 3 int32 (openvdb::Coord[n], GridData.hpp:95), leaf-dense: each
 * consecutive block of 512 is one 8^3 leaf in x<<6|y<<3|z order. The flat index of coords[i] is i by construction
 * (the reference relies on offset(coords[i]) == i+1, Kernel.cu:505 vs :511). */
hns_grid* hns_grid_create(const int32_t* coords_xyz, uint64_t n_voxels, float voxel_size, unsigned flags, int* err);
/* Same topology from the 8-aligned leaf origins alone (n_leaves x 3). */
hns_grid* hns_grid_create_from_leaves(const int32_t* leaf_origins_xyz, uint64_t n_leaves, float voxel_size, unsigned flags, int* err);
void hns_grid_destroy(hns_grid*);

uint64_t hns_grid_leaf_count(const hns_grid*);
uint64_t hns_grid_voxel_count(const hns_grid*);
float hns_grid_voxel_size(const hns_grid*);
/* Kernels update leaves [0, n_active) only; leaves [n_active, leaf_count) are ghosts that are read but never written
 * (multi-GPU halo leaves). Default n_active = leaf_count. */
int hns_grid_set_active_leaves(hns_grid*, uint64_t n_active);
/* The same for any contiguous range: kernels update leaves [first, first + count). The multi-GPU driver orders a rank's
 * leaves [boundary | interior | ghosts] and sweeps the boundary leaves first, so that their halo is on the wire while the
 * interior is being computed. */
int hns_grid_set_active_range(hns_grid*, uint64_t first, uint64_t count);
uint64_t hns_grid_active_leaves(const hns_grid*);
/* advect_scalars reads ELEMENT 0 of each array for taps outside the domain (reference Kernel.cu:133,192,225). On a
 * leaf-partitioned rank "element 0" of the global arrays lives at another local index (the ghost copy of global leaf 0):
 * this sets the flat element index those taps read. Default 0 = the reference's behaviour on an unpartitioned grid. */
int hns_grid_set_outside_element(hns_grid*, uint64_t element_index);
/* Host-side queries (work on HOST_ONLY grids): IndexOffsetSampler<0>::offset (Stencils.hpp:59-61): 1-based, 0 = outside. */
int hns_grid_offsets(const hns_grid*, const int32_t* ijk, uint64_t n, uint64_t* out);
/* 27-neighbour leaf table, n_leaves x 27 int32, entry (dx+1)*9+(dy+1)*3+(dz+1), -1 = absent. */
int hns_grid_neighbor_table(const hns_grid*, int32_t* out);
/* Writes the n_voxels x 3 coordinate array the grid represents (what the reference keeps as d_coords). */
int hns_grid_coords(const hns_grid*, int32_t* out_xyz);
/* 1 if `coords_xyz` lists exactly this grid's leaves in the same order (the grid, and the device state kept with it, can
 * serve the next cook), 0 if not, < 0 if the coordinates are not leaf-dense (same checks and flags as hns_grid_create). */
int hns_grid_matches(const hns_grid*, const int32_t* coords_xyz, uint64_t n_voxels, unsigned flags);
/* Operator calls (hns_compute_sim ... hns_divergence) keep their device buffers with the grid between calls, so a cook on
 * an unchanged topology allocates nothing; this frees them early (hns_grid_destroy does it too). */
int hns_grid_release_cache(hns_grid*);
/* Serialises the grid as a NanoVDB NanoGrid<ValueOnIndex> buffer (32.7.0 layout), the format the reference keeps its index
 * grid in (create_index_grid, HNanoSolver.cu:375-384 -> nanovdb voxelsToGrid): same header, tree, node and leaf contents,
 * so NanoVDB accessors return offset(ijk) = hns_grid_offsets(ijk). *size_out receives the byte size; pass buffer = NULL
 * to query it. `buffer` must be 32-byte aligned host memory. Works on HOST_ONLY grids. */
int hns_grid_export_nanovdb(const hns_grid*, void* buffer, uint64_t capacity, uint64_t* size_out);
/* Copy of the device-built launch order (inspection / tests): sched = n_active leaf ids in workgroup order. */
int hns_grid_launch_tables(const hns_grid*, int32_t* sched);

/* ------------------------------------------------------------------------------------------------------------ */
/* Either side of the path, without OpenVDB (host code; PARITY UNPINNED: OpenVDB is absent from the build image).   */
/* What HNS::IndexGridBuilder (src/Utils/GridBuilder.hpp:87-216) and the domain dilation of the HNanoSolver SOP      */
/* (src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:186-199) do to OpenVDB trees, over raw 8^3 leaf buffers: a leaf is its     */
/* 8-aligned origin, an optional 512-bit active mask (byte x*8+y, bit z) and 512 values in x<<6|y<<3|z order.          */
/* ------------------------------------------------------------------------------------------------------------ */

#define HNS_FILL_ZERO 0 /* float / Vec3f sources: a domain leaf the source lacks reads 0 (GridBuilder.hpp:125-129,147-151) */
#define HNS_FILL_SDF 1  /* SDF sources: filled with BYTES 0x01, i.e. 2.4e-38f, as memset(..., 1, ...) does (:108) */
/* IndexGridBuilder::build: out[i] = the source leaf at domain_origins[i] (512*ncomp floats), or the fill. Tiles of the
 * source are ignored, as in the reference (only leaves are probed). */
int hns_gather_leaves(const int32_t* domain_origins, uint64_t n_domain, const int32_t* src_origins, uint64_t n_src, const float* src_values, int ncomp, int fill,
                      float* out);
/* IndexGridBuilder::writeIndexGrid: every domain leaf receives all of its 512 values (:198-211). */
int hns_scatter_leaves(const float* flat, uint64_t n_domain, int ncomp, float* const* leaf_buffers);
/* dilateVoxels(padding, NN_FACE_EDGE_VERTEX, IGNORE_TILES) of a leaf set, as a leaf set in OpenVDB leaf order: every leaf
 * with an active voxel (active_masks: n x 64 bytes, NULL = all active) within `padding_voxels` of its box. out_origins may
 * be NULL to query *n_out. */
int hns_dilate_leaves(const int32_t* origins, uint64_t n, const unsigned char* active_masks, int padding_voxels, int32_t* out_origins, uint64_t capacity,
                      uint64_t* n_out);
/* topologyUnion of two leaf sets, in OpenVDB leaf order, duplicates removed. */
int hns_union_leaves(const int32_t* a, uint64_t na, const int32_t* b, uint64_t nb, int32_t* out_origins, uint64_t capacity, uint64_t* n_out);

/* ------------------------------------------------------------------------------------------------------------ */
/* Drop-in operators (host pointers in, results in place, synchronous)                                           */
/* ------------------------------------------------------------------------------------------------------------ */

typedef struct {
	const char* name; /* block name, e.g. "density", "vel", "collision_sdf" */
	int ncomp;        /* 1 = float block, 3 = Vec3f block (AoS) */
	float* host;      /* n_voxels * ncomp floats, caller-owned */
} hns_field;

typedef struct { /* CombustionParams, src/Cuda/Kernels.cuh:6-13 */
	float expansionRate, temperatureRelease, buoyancyStrength, ambientTemp, vorticityScale, factorScale;
} hns_combustion_params;

/* Compute_Sim: [collision] -> advect_vector -> vorticity -> divergence -> combustion -> buoyancy -> `iterations` x
 * (red, black) SOR sweeps -> gradient subtraction -> [collision] -> advect_scalars over every float block except
 * "collision_sdf". Requires exactly one ncomp==3 field, >=1 float field, and float fields named fuel, waste,
 * temperature, flame (HNanoSolver.cu:42-63,193-201). omega = 2/(1+sinf(3.14159f*voxel_size)) (:257). */
int hns_compute_sim(hns_grid*, hns_field* fields, int n_fields, int iterations, float dt, float voxel_size,
                    const hns_combustion_params* params, int has_collision, void* stream);
/* The same cook when the caller feeds the previous cook's output straight back in -- what the reference's SOP does with its first
 * input (src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:106: the "feedback" VDBs of frame n are frame n+1's state). resident[i] != 0 says that
 * fields[i].host still holds, untouched, what the previous hns_compute_sim(_resident) on this grid handed back for the block of that name;
 * such a field is not uploaded again: the device buffer it was downloaded from still holds it (the cook cache keeps operator state with
 * the grid, "cook_cache"). Two strengths:
 *   resident[i] = 1  VOUCHED. The caller knows what it changed (the SOP adds its sources itself, SOP_HNanoSolver.cpp:159-179: a block it
 *                    sourced into must NOT be flagged). The library only trips on gross mistakes: a signature of the array (its size and
 *                    4,096 evenly spread elements) must match the one taken when it was handed back. A sparse edit that misses the samples
 *                    is NOT detected and that field would silently stay stale on the device.
 *   resident[i] = 2  CHECKED. A 64-bit digest of EVERY element must match as well: an edit is noticed and the field uploaded, except with probability ~2^-64 (a
 *                    digest, not a comparison: two different arrays can collide). The digest is
 *                    taken on the DEVICE when the field is handed back (of the buffer the array is downloaded from: the same bits, one pass at
 *                    memory speed beside the downloads) by a call that asked for it, and on the HOST, on up to 8 threads, when the array comes
 *                    in again (hns_digest.hpp: an order-independent sum over 16-byte pieces). Measured at 256^3: 14.8 ms per cook against 19.8
 *                    for the plain warm cook and 11.9 vouched (profiles/r05_final_cook256.json); the first cook that asks finds no digest to
 *                    compare with and uploads.
 * Either way the field is uploaded as usual when the check fails, when the topology changed, or when another operator used the state in
 * between. resident == NULL: hns_compute_sim. *uploads_skipped (may be NULL) receives the number of fields that stayed on the device.
 * Results are bit-identical to hns_compute_sim whenever the promise holds. */
int hns_compute_sim_resident(hns_grid*, hns_field* fields, int n_fields, const unsigned char* resident, int* uploads_skipped, int iterations, float dt,
                             float voxel_size, const hns_combustion_params* params, int has_collision, void* stream);
/* AdvectIndexGrid: every float field through the single-field BFECC kernel (advect_scalar), no collision. */
int hns_advect_index_grid(hns_grid*, hns_field* fields, int n_fields, float dt, float voxel_size, void* stream);
/* AdvectIndexGridVelocity: BFECC self-advection of the one Vec3f field. */
int hns_advect_index_grid_velocity(hns_grid*, hns_field* fields, int n_fields, float dt, float voxel_size, void* stream);
/* ProjectNonDivergent: divergence -> iterations x (red, black) -> u -= grad p. omega uses double sin (PressureProjection.cu:53). */
int hns_project_non_divergent(hns_grid*, hns_field* fields, int n_fields, uint64_t iterations, float voxel_size, void* stream);
/* Divergence: writes the float field named "divergence" (PressureProjection.cu:91). */
int hns_divergence(hns_grid*, hns_field* fields, int n_fields, float voxel_size, void* stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Device-resident simulation state (upload once, many substeps; what bench.py and the multi-GPU driver use)      */
/* ------------------------------------------------------------------------------------------------------------ */

typedef struct hns_sim hns_sim;

/* Allocates velocity + the named float fields + scratch on the current device. Field order = insertion order. */
hns_sim* hns_sim_create(hns_grid*, const char* const* float_names, int n_float, int* err);
void hns_sim_destroy(hns_sim*);
int hns_sim_upload(hns_sim*, const hns_field* fields, int n_fields, void* stream);   /* matches fields by name; ncomp 3 = velocity */
int hns_sim_download(hns_sim*, hns_field* fields, int n_fields, void* stream);       /* synchronous */
/* One Compute_Sim substep entirely on the device (no H2D/D2H, no allocation, asynchronous on `stream`). */
int hns_sim_substep(hns_sim*, int iterations, float dt, float voxel_size, const hns_combustion_params*, int has_collision, void* stream);
/* The metric's core substep: advect_vector -> divergence -> iterations x RB-SOR -> gradient subtraction ->
 * advect_scalars over every float field (SURVEY.md 8d "core substep"); asynchronous on `stream`. */
int hns_sim_core_substep(hns_sim*, int iterations, float dt, float voxel_size, void* stream);
/* Only the pressure hot loop on the sim's divergence/pressure buffers (pressure zeroed first); asynchronous. */
int hns_sim_pressure_solve(hns_sim*, int iterations, float voxel_size, void* stream);
/* hipEvent timing of the pressure hot loop on its launch stream: after hns_sim_timing(sim, max_solves) every pressure
 * loop (up to max_solves) is bracketed by an event pair; hns_sim_pressure_time() returns the summed milliseconds and the
 * number of fused-iteration launches they contained. hns_sim_timing(sim, 0) switches it off. */
int hns_sim_timing(hns_sim*, int max_solves);
int hns_sim_pressure_time(hns_sim*, float* total_ms, long long* launches);
/* hns_sim_stage_timing(sim, n) brackets the five stages of the next n hns_sim_core_substep / hns_sim_substep calls (six events per substep;
 * a switch of its own because the events cost microseconds each on the launch stream); hns_sim_stage_times: ms5 receives the
 * summed milliseconds of {advect_vector, divergence, pressure loop, gradient subtraction, advect_scalars} over *substeps
 * (hns_sim_substep: {collision + advect_vector + vorticity, divergence + combustion + buoyancy, pressure loop, gradient + collision, advect_scalars}). */
int hns_sim_stage_timing(hns_sim*, int max_substeps);
int hns_sim_stage_times(hns_sim*, float* ms5, long long* substeps);
/* Raw device pointers of the sim's buffers (Vec3f AoS velocity, float fields, divergence, pressure). */
float* hns_sim_velocity_ptr(hns_sim*);
float* hns_sim_field_ptr(hns_sim*, const char* name);
float* hns_sim_divergence_ptr(hns_sim*);
float* hns_sim_pressure_ptr(hns_sim*);

/* ------------------------------------------------------------------------------------------------------------ */
/* Kernel-level entry points on caller-owned DEVICE memory (asynchronous on `stream`).                           */
/* Velocity fields are Vec3f AoS on the device too (3 floats per voxel, `vel3`), exactly the host/reference layout.  */
/* ------------------------------------------------------------------------------------------------------------ */

/* advect_vector (Kernel.cu:354-453); out3 must not alias vel3 */
int hns_dev_advect_vector(hns_grid*, const float* vel3, float* out3, const float* sdf, int has_collision, float dt, float inv_dx, void* stream);
/* advect_scalar (Kernel.cu:269-352) */
int hns_dev_advect_scalar(hns_grid*, const float* vel3, const float* in, float* out, const float* sdf, int has_collision, float dt, float inv_dx,
                          void* stream);
/* advect_scalars (Kernel.cu:118-266); in/out are HOST arrays of n device pointers */
int hns_dev_advect_scalars(hns_grid*, const float* vel3, const float* const* in, float* const* out, int n, const float* sdf, int has_collision,
                           float dt, float inv_dx, void* stream);
/* divergence / divergence_opt (Kernel.cu:455-519) */
int hns_dev_divergence(hns_grid*, const float* vel3, float* div, float inv_dx, void* stream);
/* One colour of redBlackGaussSeidelUpdate(_opt) in place (Kernel.cu:521-623): the two-launch form. */
int hns_dev_rbgs_color(hns_grid*, const float* div, float* p, float dx, float omega, int color, void* stream);
/* `iterations` full (red, black) iterations, starting from p_a and ping-ponging p_a -> p_b -> p_a ... once per LAUNCH.
 * Bit-identical to 2*iterations calls of hns_dev_rbgs_color. How many iterations a launch holds depends on the form the
 * library picks for this grid (hns_grid_rbgs_plan: the temporally blocked form does 2 or 4 per launch, 1 for an odd one left over; "rbgs" = color: two launches per iteration), so
 * WHICH BUFFER HOLDS THE RESULT IS NOT A FUNCTION OF `iterations`: *result_in_b is 1 when it is p_b, 0 when it is p_a
 * (e.g. blocked form: iterations = 2 -> p_b, 4 -> p_b or p_a, 6 -> p_b or p_a). A caller that wants the result must pass
 * result_in_b and read it (NULL is accepted from callers that only time the solve). The other buffer holds an intermediate
 * iterate. p_a and p_b must not alias.
 * On a grid with a LAUNCH RANGE (hns_grid_set_active_range / _leaves: a multi-GPU rank's boundary / interior / owned leaves) only the leaves
 * of the range are written, and the statement above holds for ONE launch: a temporally blocked launch advances the leaves outside the range
 * inside its tiles instead of reading them as fixed, so its result on the range equals what a WHOLE-grid sweep leaves there -- provided p within
 * 2K voxels of the range and div within 2K - 1 are current in the leaves outside it (K = iterations per launch, hns_grid_rbgs_plan). A second
 * launch reads the other buffer, whose out-of-range leaves the first never wrote: the caller must refresh those voxels in BOTH buffers between
 * launches (hns_dist does: an exchange, or its peers' mirror stores, after every launch). With nothing refreshing them, use one launch per call. */
int hns_dev_rbgs_iterate(hns_grid*, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int* result_in_b,
                         void* stream);
/* subtractPressureGradient(_opt) (Kernel.cu:694-829); out3 may alias vel3 (each voxel reads only its own velocity) */
int hns_dev_subtract_pressure_gradient(hns_grid*, const float* vel3, const float* p, float* out3, const float* sdf, int has_collision, float inv_dx,
                                       void* stream);
/* combustion_oxygen (Kernel.cu:923-966) */
int hns_dev_combustion_oxygen(const float* fuel, const float* waste, const float* temperature, float* divergence, const float* flame,
                              float* out_fuel, float* out_waste, float* out_temperature, float* out_flame, float temp_gain, float expansion,
                              uint64_t n, void* stream);
/* temperature_buoyancy (Kernel.cu:831-847); out3 may alias vel3 */
int hns_dev_temperature_buoyancy(const float* vel3, const float* temperature, float* out3, float dt, float ambient, float strength, uint64_t n,
                                 void* stream);
/* vorticityConfinement (Kernel.cu:970-1024), out-of-place (the reference's in-place launch races when factor_scale >= 1) */
int hns_dev_vorticity_confinement(hns_grid*, const float* vel3, float* out3, float dt, float inv_dx, float confinement_scale, float factor_scale,
                                  void* stream);
/* enforceCollisionBoundaries (Kernel.cu:77-116), in place */
int hns_dev_enforce_collision_boundaries(hns_grid*, float* vel3, const float* sdf, float voxel_size, void* stream);

/* Halo support for leaf-partitioned multi-GPU runs: copy whole leaves (512*ncomp floats each; ncomp 1 = float field,
 * 3 = Vec3f field) between a field and a packed buffer. leaf_ids is a DEVICE array of n leaf indices. */
int hns_dev_pack_leaves(const float* field, const int32_t* leaf_ids, uint64_t n, float* packed, int ncomp, void* stream);
int hns_dev_unpack_leaves(const float* packed, const int32_t* leaf_ids, uint64_t n, float* field, int ncomp, void* stream);

/* ------------------------------------------------------------------------------------------------------------ */
/* Leaf-partitioned multi-GPU core substep (new: the reference is single-GPU). One hns_dist per rank = per GPU.      */
/* Rank r owns the r-th of `world` equal ranges of the global leaf list in slab order (hns_dist_partition_axis below), */
/* keeps one layer of ghost leaves and refreshes exactly the ghost voxels the next kernel can read (hns_dist_*.hip). */
/* Owned results are bit-identical to the single-domain hns_sim_core_substep -- as long as every tap of an advection   */
/* back-trace that leaves the 27-leaf neighbourhood of its voxel's leaf (|u| dt / dx above ~8 voxels) still lands in   */
/* a leaf this rank holds (owned or ghost: a rank holds ONE layer of ghost leaves). A far tap whose leaf is not here   */
/* may exist on another rank: it is detected on the device and the next hns_dist_*_substep / hns_dist_synchronize /    */
/* hns_dist_download returns HNS_ERR_RUNTIME saying so; hns_dist_upload clears it. (Far taps deep inside a rank are    */
/* answered through the origin hash like on one GPU; a far tap beyond the true edge of the domain is reported too --   */
/* the rank cannot tell it from one beyond its ghost layer. The single-GPU path has no such limit.)                    */
/* ------------------------------------------------------------------------------------------------------------ */

typedef struct hns_dist hns_dist;

typedef struct {
	int world, rank, peers, sweeps_per_exchange;
	uint64_t boundary_leaves, interior_leaves, ghost_leaves; /* local leaf order: [boundary | interior | ghosts] */
	/* per halo region type {advection inputs (whole leaves, global leaf 0 = the element-0 mirror among them), L1 reach 1, div (2k-1), p (2k)}: */
	uint64_t region_voxels_sent[4]; /* voxels this rank sends per exchange of that type (a property of the plan)   */
	uint64_t bytes_sent[4];         /* payload bytes this rank sent during the last substep                          */
	uint64_t messages_sent, exchanges; /* point-to-point messages / exchange rounds of the last substep              */
	uint64_t halo_peers; /* peers this rank exchanges div / p / reach-1 halo voxels with (slab partition: the rank before and the rank behind it);
	                        `peers` also counts ranks it only shares the element-0 mirror of the caller's leaf 0 with (its owner: every rank) */
	uint64_t packed_exchanges; /* exchanges of the last substep whose messages the sweep wrote itself as it stored (no pack launch) */
	uint64_t chained;          /* 1 once connected if this rank runs the CHAINED substep (ipc / local transports, 16^3-block ranks, sweeps_per_exchange = 2: every kernel stores
	                              its boundary values into the peers' ghost voxels itself, no exchanges); 0 = the exchanged substep */
} hns_dist_stats;

/* sweeps_per_exchange (1..4, 0 = default 4): the pressure loop refreshes the ghosts of p after every k-th fused sweep and
 * sweeps the ghost leaves locally in between. n_scalars float fields are advected (the metric's core substep uses 1). */
#define HNS_DIST_DEFAULT 0u
#define HNS_DIST_PLAN_ONLY 1u /* build the partition plan on the host only (inspection / CPU tests); compute calls then fail */
#define HNS_DIST_LEAF_ORDER 2u /* rank r owns leaves [n*r/world, n*(r+1)/world) of the caller's list whatever the domain (the rule of rounds 1-4) */
hns_dist* hns_dist_create(const int32_t* global_leaf_origins_xyz, uint64_t n_leaves, int world, int rank, float voxel_size, int n_scalars,
                          int sweeps_per_exchange, unsigned flags, int* err);
void hns_dist_destroy(hns_dist*);
/* Transport, RCCL over xGMI (one process per GPU): rank 0 calls hns_dist_unique_id and hands the 128 bytes to every rank
 * by any means (torch.distributed broadcast, MPI, a file); every rank then calls hns_dist_connect_rccl (collective). */
int hns_dist_unique_id(void* out128);
int hns_dist_connect_rccl(hns_dist*, const void* unique_id128);
/* Transport, one-sided over mapped peer memory (one process per GPU, xGMI peer access; also between processes sharing one
 * GPU): every rank calls hns_dist_ipc_export, the HNS_DIST_IPC_BLOB_BYTES-byte blobs travel by any host means, every rank
 * calls hns_dist_connect_ipc with all `world` blobs in rank order. A rank then PUTS its messages into the peer's receive
 * buffer / ghost voxels with a copy kernel; the two sides meet through sequence-numbered flags in device memory (bounded
 * waits: a peer that does not answer within 20 s makes the next call fail instead of hanging the device). */
#define HNS_DIST_IPC_BLOB_BYTES 2048
int hns_dist_ipc_export(hns_dist*, void* out_blob);
int hns_dist_connect_ipc(hns_dist*, const void* blobs_of_all_ranks);
/* Transport, local: all `world` ranks live in this process on ONE device; a message is a device copy out of the peer's
 * send buffer. Same plan, kernels, streams and events as the RCCL path (tests; per-rank overhead without a wire). */
int hns_dist_connect_local(hns_dist* const* ranks, int world);
/* Transport, loopback (TIMING ONLY, results are meaningless): this rank alone, every message answered with the rank's own
 * payload of the same size. Measures what one rank costs next to the single-GPU substep before any wire time. */
int hns_dist_connect_loopback(hns_dist*);
/* The same with every message carried by RCCL (a one-rank communicator, sends and receives to itself in the groups the
 * multi-rank path issues): what a single-GPU box can check of the RCCL path. Leaves what the copy-based loopback leaves. */
int hns_dist_connect_loopback_rccl(hns_dist*);
uint64_t hns_dist_owned_leaves(const hns_dist*);
/* Which leaves a rank owns (round 5): the leaves in SLAB order -- by leaf coordinate along the axis whose cuts cross the fewest leaves, the
 * caller's order inside a plane -- cut into `world` equal ranges, so that a rank exchanges halos with the rank before and the rank behind
 * it only (BASELINE config 5 in 8 ranks: 2 halo peers instead of 7). Where the x cut selects the same leaf sets as contiguous ranges of the
 * caller's list (box domains made of whole 128-voxel NanoVDB nodes per rank) the caller's order is kept: hns_dist_partition_axis = -1 and
 * hns_dist_first_owned_leaf = the first leaf of the contiguous run. Otherwise hns_dist_partition_axis = 0 / 1 / 2, hns_dist_first_owned_leaf
 * = ~0, and hns_dist_owned_leaf_ids lists the owned leaves (positions in the caller's list) in the order hns_dist_upload / _download expect. */
uint64_t hns_dist_first_owned_leaf(const hns_dist*);
int hns_dist_owned_leaf_ids(const hns_dist*, int64_t* out_global_ids); /* hns_dist_owned_leaves entries */
int hns_dist_partition_axis(const hns_dist*);
/* sweeps_per_exchange to create the ranks with for the chained one-sided substep (ipc / local transports) of `world` ranks over n_leaves leaves: 2 = the
 * temporally blocked chained sweep, two iterations per launch, where every rank's range is large enough for 16^3 blocks under the current options; else 1 */
int hns_dist_one_sided_sweeps(uint64_t n_leaves, int world);
int hns_dist_info(const hns_dist*, hns_dist_stats* out);
/* The plan (also on PLAN_ONLY handles): global id of every local leaf in local order [boundary | interior | ghosts]; the
 * rank of peer i (-1 beyond the last); and per peer, halo region type (0..3 as in hns_dist_stats) and direction the
 * local leaves whose voxels travel plus a 64-byte mask each (byte x*8+y, bit z). Any output pointer may be NULL. */
int hns_dist_local_leaves(const hns_dist*, int64_t* out_global_ids);
int hns_dist_peer_rank(const hns_dist*, int peer);
int hns_dist_peer_region(const hns_dist*, int peer, int type, int is_send, int32_t* leaves, unsigned char* masks, uint64_t* n_leaves, uint64_t* n_voxels);
/* Host arrays over the OWNED leaves in the order of hns_dist_owned_leaf_ids (vel3: 512*3 floats per leaf, each scalar 512 per leaf).
 * Collective in effect: every rank uploads before the next substep (the first exchange then carries the new fields).
 * Synchronous. download: any pointer may be NULL; `pressure` receives the last solve's p. */
int hns_dist_upload(hns_dist*, const float* vel3, const float* const* scalars, void* stream);
int hns_dist_download(hns_dist*, float* vel3, float* const* scalars, float* pressure, void* stream);
/* Diagnostics: one field over ALL local leaves, ghosts included, in local order (hns_dist_local_leaves), as the device holds it
 * now; which = -2: the last solve's p, -1: velocity (3 floats per voxel), s >= 0: scalar s. Synchronous. With
 * hns_dist_peer_region it lets a driver check that a rank's ghost voxels equal their owners' values (DistRank.ghost_check). */
int hns_dist_download_local(hns_dist*, int which, float* out, void* stream);
/* One core substep of this rank, asynchronous on `stream` and on the rank's communication stream (RCCL transport, or world 1). */
int hns_dist_core_substep(hns_dist*, int iterations, float dt, void* stream);
/* The same for locally connected ranks: all of them advance together, phase by phase, on `stream`. */
int hns_dist_local_core_substep(hns_dist* const* ranks, int world, int iterations, float dt, void* stream);
/* hipEvent bracketing of the pressure loops (halo exchanges included), as hns_sim_timing / hns_sim_pressure_time. */
int hns_dist_timing(hns_dist*, int max_solves);
/* The whole Compute_Sim substep on the partitioned domain (reference HNanoSolver.cu:150-356; single GPU: hns_sim_substep): collision,
 * advect_vector, vorticity confinement, divergence, combustion, buoyancy, the pressure solve, gradient subtraction, collision, and the
 * advection of every float field except collision_sdf. field_index[5]: the positions of fuel, waste, temperature, flame and
 * collision_sdf (-1: none) among the rank's scalars in hns_dist_upload order. Every kernel boundary a stencil crosses is a halo
 * exchange (vorticity confinement reads whole ghost leaves of u*); the pointwise kernels run on the owned leaves. Owned results are
 * bit-identical to hns_sim_substep on the whole domain (tests/test_dist_gpu.py). factor_scale must stay within 0..6. */
int hns_dist_sim_substep(hns_dist*, int iterations, float dt, const hns_combustion_params*, const int* field_index, int has_collision, void* stream);
int hns_dist_local_sim_substep(hns_dist* const* ranks, int world, int iterations, float dt, const hns_combustion_params*, const int* field_index, int has_collision,
                               void* stream);
int hns_dist_pressure_time(hns_dist*, float* total_ms, long long* sweeps);
int hns_dist_synchronize(hns_dist*, void* stream); /* waits for `stream` and the communication stream */

/* Timing helper: runs `iterations` fused RB-SOR iterations `reps` times on `stream`, bracketing each launch group with
 * hipEvents on that stream, and returns the mean milliseconds per fused-iteration launch. */
int hns_dev_time_rbgs(hns_grid*, const float* div, float* p_a, float* p_b, float dx, float omega, int iterations, int reps, float* ms_per_launch,
                      void* stream);

/* Which kernel form hns_dev_rbgs_iterate / the operators' pressure loops use for `iterations` iterations on this grid under the
 * current options, as text (e.g. "k_rbgs_block<2,2>: 2 iterations per launch on 16^3-voxel blocks"), and how many kernel launches
 * that is. For bench / profile labels; `description` may be null. */
int hns_grid_rbgs_plan(hns_grid*, int iterations, char* description, uint64_t description_bytes, int* launches, int* iterations_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* HNS_H */
