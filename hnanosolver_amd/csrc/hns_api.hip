// hns_api.hip -- the device-resident simulation state and the drop-in operators of
// include/hns.h. The launch orders follow the reference's host drivers (reference src/Cuda/HNanoSolver.cu:150-356,
// src/Cuda/PressureProjection.cu:43-66, src/Cuda/Advection.cu:76-91,148-155); what differs is that fields can stay
// resident across substeps and that nothing is allocated inside a substep.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hns_internal.hpp"
#include "hns_digest.hpp"

using namespace hns;

#define HNS_HIP(call)                                                                                  \
	do {                                                                                               \
		hipError_t e__ = (call);                                                                       \
		if (e__ != hipSuccess) {                                                                       \
			set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);     \
			return HNS_ERR_HIP;                                                                        \
		}                                                                                              \
	} while (0)

#define HNS_TRY(call)              \
	do {                           \
		int rc__ = (call);         \
		if (rc__ != HNS_OK) return rc__; \
	} while (0)

// ---------------------------------------------------------------------------------------------------------------
// grid: device tables
// ---------------------------------------------------------------------------------------------------------------

GridDev hns_grid::dev() const {
	GridDev d;
	d.origins = (const int4*)d_origins;
	d.nbr27 = (const int*)d_nbr27;
	d.hash = (const int*)d_hash;
	d.sched = (const int*)d_sched;
	d.sched_seg = sched_seg, d.sched_pre = sched_pre;
	d.blk = (const int*)d_blk;
	d.hash_mask = topo.hash_mask;
	d.n_leaves = (int)topo.n_leaves;
	d.n_active = (int)n_active;
	d.first = (int)first_active;
	d.oob = (int)outside_element;
	d.rev = 0;
	d.far_flag = far_flag;
	return d;
}

extern "C" int hns_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident simulation state
// ---------------------------------------------------------------------------------------------------------------

struct hns_sim {
	hns_grid* grid = nullptr;
	uint64_t n = 0;  // voxels
	std::vector<std::string> names;
	std::vector<float*> cur;  // current value of each float field (the reference's d_inputs)
	std::vector<float*> nxt;  // scratch / next value        (the reference's d_outputs)
	float* vel = nullptr;  // d_velocity      (Vec3f AoS, 3n floats: the host/reference layout, so H2D/D2H are plain copies)
	float* adv = nullptr;  // d_advectedVel
	float* tmp = nullptr;  // out-of-place vorticity target; the buoyed u* of the fused divergence / combustion / buoyancy launch
	float* q4 = nullptr;   // {fuel, waste, temperature, flame} as one 16-byte element per voxel between that launch and advect_scalars (sims that hold those four fields)
	float* div = nullptr;
	float* p_a = nullptr;
	float* p_b = nullptr;
	float* p_result = nullptr;  // whichever of p_a/p_b holds the last solve
	// optional hipEvent bracketing of the pressure hot loop (hns_sim_timing), on the stream the kernels run on
	bool timing = false;
	std::vector<hipEvent_t> ev;  // start/stop pairs
	size_t ev_used = 0;
	long long timed_launches = 0;
	bool stage_timing = false;    // hns_sim_stage_timing: also bracket the five stages of hns_sim_core_substep
	std::vector<hipEvent_t> sev;  // stage boundaries: six per substep
	size_t sev_used = 0;
	hipStream_t xfer = nullptr;  // transfer stream + hand-off events of the pipelined operator path (compute_sim_pipelined)
	hipEvent_t xev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	bool cached = false, in_use = false;  // owned by the grid's cook cache / currently lent to an operator call
	// Device-resident feedback across cooks (hns_compute_sim_resident): a signature of what the last hns_compute_sim on this state handed
	// back for the velocity and for float field i -- those bytes are still in `vel` / cur[i]. 0 = nothing to vouch for (any upload clears it).
	uint64_t sig_vel = 0, dig_vel = 0;  // (sig: sample signature; dig: full digest, 0 = not taken)
	unsigned long long* d_dig = nullptr;  // 16 accumulators of the digest kernels (hns_digest.hpp): a slice of the arena
	unsigned long long* h_dig = nullptr;  // pinned host copy of them (read asynchronously on the cook's own stream)
	std::vector<uint64_t> sig_cur, dig_cur;
	void* arena = nullptr;  // every field above is a slice of this one allocation (see the arena pool below)
	size_t arena_bytes = 0;
	int device = -1;
	int find(const char* name) const {
		for (size_t i = 0; i < names.size(); ++i)
			if (names[i] == name) return (int)i;
		return -1;
	}
};

// ---- arena pool ------------------------------------------------------------------------------------------------
// A sim's fields are slices of ONE device allocation, and allocations that are no longer needed go to a small
// process-wide pool instead of back to the driver. The reference pays cudaMallocAsync x 15+ and the matching frees every
// cook (HNanoSolver.cu:87-133,361-369); with separate hipMallocs here a cold cook at 256^3 spent 4.2 ms in hipFree
// alone. A sparse simulation changes its topology nearly every frame, so "cold" is its normal cook: with the pool the
// new grid's fields land in the previous grid's memory whenever that is large enough. hns_trim_memory() empties it.
namespace {
struct Arena {
	void* p;
	size_t bytes;
	int device;
};
// makes `device` current for the scope (allocations, frees and synchronisation of pooled memory belong to ITS device,
// whatever the calling thread has current)
struct DeviceScope {
	int prev = -1;
	bool switched = false;
	explicit DeviceScope(int device) {
		if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
	}
	~DeviceScope() {
		if (switched) (void)hipSetDevice(prev);
	}
};
std::mutex g_pool_mutex;
std::vector<Arena> g_pool;  // at most kPoolMax idle arenas
constexpr size_t kPoolMax = 6;  // simulation states (GBs) and grid tables (MBs) share it; the smallest goes first

int arena_get(size_t need, int device, Arena& out) {
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		int best = -1;
		for (size_t i = 0; i < g_pool.size(); ++i)
			if (g_pool[i].device == device && g_pool[i].bytes >= need && g_pool[i].bytes <= 2 * need + (64u << 20) &&
			    (best < 0 || g_pool[i].bytes < g_pool[(size_t)best].bytes))
				best = (int)i;
		if (best >= 0) {
			out = g_pool[(size_t)best];
			g_pool.erase(g_pool.begin() + best);
			return HNS_OK;
		}
	}
	out.bytes = need + need / 8;  // headroom: the next, slightly larger topology still fits
	out.device = device;
	DeviceScope scope(device);
	if (hipMalloc(&out.p, out.bytes) != hipSuccess) {
		(void)hipGetLastError();
		std::vector<Arena> drop;  // out of memory with idle arenas around: release them and retry at the exact size
		{
			std::lock_guard<std::mutex> lock(g_pool_mutex);
			drop.swap(g_pool);
		}
		for (Arena& a : drop) (void)hipFree(a.p);
		out.bytes = need;
		HNS_HIP(hipMalloc(&out.p, out.bytes));
	}
	return HNS_OK;
}

// The hipFree this pool replaces waits for the device; so does this: whoever draws the memory next may use it on any
// stream without ordering itself after the previous owner's queued kernels and copies. Cooks are synchronous, so the
// device is normally idle here and the wait costs microseconds.
void arena_put(const Arena& a) {
	if (!a.p) return;
	DeviceScope scope(a.device);
	(void)hipDeviceSynchronize();
	Arena evict{nullptr, 0, -1};
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		g_pool.push_back(a);
		if (g_pool.size() > kPoolMax) {  // drop the smallest
			size_t k = 0;
			for (size_t i = 1; i < g_pool.size(); ++i)
				if (g_pool[i].bytes < g_pool[k].bytes) k = i;
			evict = g_pool[k];
			g_pool.erase(g_pool.begin() + (long)k);
		}
	}
	if (evict.p) (void)hipFree(evict.p);
}
}  // namespace

extern "C" int hns_arena_get(size_t need, int device, void** p, size_t* bytes) {
	Arena a{nullptr, 0, -1};
	const int rc = arena_get(need, device, a);
	*p = a.p;
	*bytes = a.bytes;
	return rc;
}
extern "C" void hns_arena_put(void* p, size_t bytes, int device) { arena_put(Arena{p, bytes, device}); }

// Returns the idle pooled device memory to the driver.
extern "C" int hns_trim_memory(void) {
	std::vector<Arena> drop;
	{
		std::lock_guard<std::mutex> lock(g_pool_mutex);
		drop.swap(g_pool);
	}
	for (Arena& a : drop) HNS_HIP(hipFree(a.p));
	return HNS_OK;
}

extern "C" void hns_sim_destroy(hns_sim* s) {
	if (!s) return;
	for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
	for (hipEvent_t e : s->sev) (void)hipEventDestroy(e);
	for (hipEvent_t e : s->xev)
		if (e) (void)hipEventDestroy(e);
	if (s->xfer) (void)hipStreamDestroy(s->xfer);
	if (s->h_dig) (void)hipHostFree(s->h_dig);
	arena_put(Arena{s->arena, s->arena_bytes, s->device});
	delete s;
}

// Frees the device-resident state operator calls left with the grid (see make_sim below).
extern "C" int hns_grid_release_cache(hns_grid* g) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_release_cache: null grid");
	std::vector<hns_sim*> drop;
	{
		std::lock_guard<std::mutex> lock(g->host_mutex);
		for (size_t i = 0; i < g->sim_cache.size();) {
			if (g->sim_cache[i]->in_use) {
				++i;
				continue;
			}
			drop.push_back(g->sim_cache[i]);
			g->sim_cache.erase(g->sim_cache.begin() + (long)i);
		}
	}
	for (hns_sim* s : drop) {
		s->cached = false;
		hns_sim_destroy(s);
	}
	return HNS_OK;
}

// zero: clear the arena (what hns_sim_create promises); the operator path skips it when every buffer is written before it is read
static hns_sim* sim_create(hns_grid* g, const char* const* float_names, int n_float, bool zero, void* stream, int* rc_out) {
	hns_sim* s = new hns_sim;
	s->grid = g;
	s->n = hns_grid_voxel_count(g);
	s->device = g->device;
	int rc = HNS_OK;
	for (int i = 0; i < n_float && rc == HNS_OK; ++i) {
		if (!float_names[i]) {
			rc = fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_create: null field name");
		} else if (s->find(float_names[i]) >= 0) {
			set_error("hns_sim_create: duplicate field name '%s'", float_names[i]);
			rc = HNS_ERR_INVALID_ARGUMENT;
		} else {
			s->names.push_back(float_names[i]);
		}
	}
	if (rc == HNS_OK) {
		const size_t unit = (sizeof(float) * (size_t)(s->n ? s->n : 1) + 255) & ~(size_t)255;  // one float field, 256-byte aligned
		const bool combust = s->find("fuel") >= 0 && s->find("waste") >= 0 && s->find("temperature") >= 0 && s->find("flame") >= 0;
		const size_t units = 3 * 3 + 3 + 2 * s->names.size() + (combust ? 4 : 0);               // vel, adv, tmp | div, p_a, p_b | cur, nxt per field | q4
		Arena a{nullptr, 0, -1};
		rc = arena_get(unit * units + 256, s->device, a);  // (+ the 16 digest accumulators of hns_compute_sim_resident, CHECKED fields)
		if (rc == HNS_OK) {
			s->arena = a.p, s->arena_bytes = a.bytes;
			char* q = (char*)a.p;
			auto take = [&](size_t k) {
				float* r = (float*)q;
				q += k * unit;
				return r;
			};
			s->vel = take(3), s->adv = take(3), s->tmp = take(3);
			s->div = take(1), s->p_a = take(1), s->p_b = take(1);
			for (size_t i = 0; i < s->names.size(); ++i) {
				s->cur.push_back(take(1));
				s->nxt.push_back(take(1));
			}
			if (combust) s->q4 = take(4);
			s->d_dig = (unsigned long long*)q;
			s->p_result = s->p_a;
			if (zero && hipMemsetAsync(a.p, 0, unit * units, (hipStream_t)stream) != hipSuccess) rc = fail(HNS_ERR_HIP, "hns_sim_create: clearing the field memory failed");
		}
	}
	if (rc != HNS_OK) {
		hns_sim_destroy(s);
		s = nullptr;
	}
	*rc_out = rc;
	return s;
}

extern "C" hns_sim* hns_sim_create(hns_grid* g, const char* const* float_names, int n_float, int* err) {
	int rc = HNS_OK;
	if (!g || n_float < 0 || (n_float > 0 && !float_names)) {
		rc = fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_create: bad arguments");
	} else if (!g->on_device) {
		rc = fail(HNS_ERR_NO_DEVICE, "hns_sim_create: grid has no device tables (there is no CPU fallback)");
	}
	if (rc != HNS_OK) {
		if (err) *err = rc;
		return nullptr;
	}
	hns_sim* s = sim_create(g, float_names, n_float, true, nullptr, &rc);
	// the clear ran on the null stream: finish it, so that the caller may use any stream (also a non-blocking one) afterwards
	if (s && hipStreamSynchronize(nullptr) != hipSuccess) {
		rc = fail(HNS_ERR_HIP, "hns_sim_create: clearing the field memory failed");
		hns_sim_destroy(s);
		s = nullptr;
	}
	if (err) *err = rc;
	return s;
}

extern "C" int hns_sim_upload(hns_sim* s, const hns_field* fields, int n_fields, void* stream) {
	if (!s || (n_fields > 0 && !fields)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_upload: null argument");
	hipStream_t st = (hipStream_t)stream;
	for (int i = 0; i < n_fields; ++i) {
		const hns_field& f = fields[i];
		if (!f.host) {
			set_error("hns_sim_upload: host pointer is null for block: %s", f.name ? f.name : "?");
			return HNS_ERR_RUNTIME;
		}
		if (f.ncomp == 3) {
			s->sig_vel = s->dig_vel = 0;
			HNS_HIP(hipMemcpyAsync(s->vel, f.host, sizeof(float) * 3 * (size_t)s->n, hipMemcpyHostToDevice, st));
		} else if (f.ncomp == 1) {
			const int k = f.name ? s->find(f.name) : -1;
			if (k < 0) {
				set_error("hns_sim_upload: no float field named '%s' in this sim", f.name ? f.name : "?");
				return HNS_ERR_RUNTIME;
			}
			if ((size_t)k < s->sig_cur.size()) s->sig_cur[(size_t)k] = 0;
			if ((size_t)k < s->dig_cur.size()) s->dig_cur[(size_t)k] = 0;
			HNS_HIP(hipMemcpyAsync(s->cur[k], f.host, sizeof(float) * (size_t)s->n, hipMemcpyHostToDevice, st));
		} else {
			return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_upload: ncomp must be 1 or 3");
		}
	}
	return HNS_OK;
}

extern "C" int hns_sim_download(hns_sim* s, hns_field* fields, int n_fields, void* stream) {
	if (!s || (n_fields > 0 && !fields)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_download: null argument");
	hipStream_t st = (hipStream_t)stream;
	for (int i = 0; i < n_fields; ++i) {
		hns_field& f = fields[i];
		if (!f.host) return fail(HNS_ERR_RUNTIME, "hns_sim_download: null host pointer");
		if (f.ncomp == 3) {
			HNS_HIP(hipMemcpyAsync(f.host, s->vel, sizeof(float) * 3 * (size_t)s->n, hipMemcpyDeviceToHost, st));
		} else if (f.ncomp == 1) {
			const int k = f.name ? s->find(f.name) : -1;
			if (k < 0) {
				set_error("hns_sim_download: no float field named '%s' in this sim", f.name ? f.name : "?");
				return HNS_ERR_RUNTIME;
			}
			HNS_HIP(hipMemcpyAsync(f.host, s->cur[k], sizeof(float) * (size_t)s->n, hipMemcpyDeviceToHost, st));
		} else {
			return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_download: ncomp must be 1 or 3");
		}
	}
	HNS_HIP(hipStreamSynchronize(st));
	return HNS_OK;
}

extern "C" float* hns_sim_velocity_ptr(hns_sim* s) { return s ? s->vel : nullptr; }
extern "C" float* hns_sim_field_ptr(hns_sim* s, const char* name) {
	if (!s || !name) return nullptr;
	const int k = s->find(name);
	return k < 0 ? nullptr : s->cur[k];
}
extern "C" float* hns_sim_divergence_ptr(hns_sim* s) { return s ? s->div : nullptr; }
extern "C" float* hns_sim_pressure_ptr(hns_sim* s) { return s ? s->p_result : nullptr; }

static int validate_step(float voxel_size, float dt, int64_t iterations, bool need_iter) {
	if (voxel_size <= 0.0f) return fail(HNS_ERR_INVALID_ARGUMENT, "voxelSize must be positive.");                          // HNanoSolver.cu:12-14
	if (dt < 0.0f) return fail(HNS_ERR_INVALID_ARGUMENT, "dt (time step) cannot be negative.");                           // :15-17
	if (need_iter && iterations <= 0) return fail(HNS_ERR_INVALID_ARGUMENT, "Number of pressure iterations must be positive.");  // :18-20
	return HNS_OK;
}

// the pressure hot loop: p = 0, `iterations` x (red, black); HNanoSolver.cu:256-272 / PressureProjection.cu:51-60
static int sim_pressure(hns_sim* s, int iterations, float voxel_size, float omega, void* stream) {
	int in_b = 0;  // never warm-started (HNanoSolver.cu:113): the solve starts from p = 0, which the first sweep knows without reading p_a
	const bool timed = s->timing && s->ev_used + 2 <= s->ev.size();
	if (timed) HNS_HIP(hipEventRecord(s->ev[s->ev_used], (hipStream_t)stream));
	HNS_TRY(hns_rbgs_iterate(s->grid, s->div, s->p_a, s->p_b, voxel_size, omega, iterations, &in_b, stream, true));
	if (timed) {
		HNS_HIP(hipEventRecord(s->ev[s->ev_used + 1], (hipStream_t)stream));
		s->ev_used += 2;
		s->timed_launches += iterations;
	}
	s->p_result = in_b ? s->p_b : s->p_a;
	return HNS_OK;
}

static float omega_compute(float vs) { return 2.0f / (1.0f + sinf(static_cast<float>(3.14159) * vs)); }          // HNanoSolver.cu:257
static float omega_project(float vs) { return (float)(2.0f / (1.0f + sin(3.14159 * (double)vs))); }             // PressureProjection.cu:53

extern "C" int hns_sim_pressure_solve(hns_sim* s, int iterations, float voxel_size, void* stream) {
	if (!s) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_pressure_solve: null sim");
	HNS_TRY(validate_step(voxel_size, 0.0f, iterations, true));
	return sim_pressure(s, iterations, voxel_size, omega_compute(voxel_size), stream);
}

// Bracket every following pressure loop (up to max_solves of them) with a hipEvent pair on its launch stream.
extern "C" int hns_sim_timing(hns_sim* s, int max_solves) {
	if (!s || max_solves < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_timing: bad arguments");
	while (s->ev.size() < (size_t)max_solves * 2) {
		hipEvent_t e;
		HNS_HIP(hipEventCreate(&e));
		s->ev.push_back(e);
	}
	s->timing = max_solves > 0;
	s->ev_used = 0;
	s->timed_launches = 0;
	return HNS_OK;
}

// Bracket the five stages of the next max_substeps hns_sim_core_substep calls (six events per substep: a few microseconds
// each on the launch stream, which is why this is a switch of its own and not part of hns_sim_timing).
extern "C" int hns_sim_stage_timing(hns_sim* s, int max_substeps) {
	if (!s || max_substeps < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_stage_timing: bad arguments");
	while (s->sev.size() < (size_t)max_substeps * 6) {
		hipEvent_t e;
		HNS_HIP(hipEventCreate(&e));
		s->sev.push_back(e);
	}
	s->stage_timing = max_substeps > 0;
	s->sev_used = 0;
	return HNS_OK;
}

// Per-stage device time of the core substeps run since hns_sim_stage_timing(): ms5 = {advect_vector, divergence, pressure loop,
// gradient subtraction, advect_scalars}, summed over `*substeps` substeps; events sit on the launch stream.
extern "C" int hns_sim_stage_times(hns_sim* s, float* ms5, long long* substeps) {
	if (!s || !ms5 || !substeps) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_stage_times: null argument");
	double tot[5] = {0, 0, 0, 0, 0};
	for (size_t i = 0; i + 5 < s->sev_used; i += 6) {
		HNS_HIP(hipEventSynchronize(s->sev[i + 5]));
		for (int k = 0; k < 5; ++k) {
			float ms = 0.0f;
			HNS_HIP(hipEventElapsedTime(&ms, s->sev[i + k], s->sev[i + k + 1]));
			tot[k] += ms;
		}
	}
	for (int k = 0; k < 5; ++k) ms5[k] = (float)tot[k];
	*substeps = (long long)(s->sev_used / 6);
	return HNS_OK;
}

// Sum of the bracketed pressure-loop times since hns_sim_timing() and the number of fused-iteration launches inside.
extern "C" int hns_sim_pressure_time(hns_sim* s, float* total_ms, long long* launches) {
	if (!s || !total_ms || !launches) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_pressure_time: null argument");
	double tot = 0.0;
	for (size_t i = 0; i + 1 < s->ev_used; i += 2) {
		HNS_HIP(hipEventSynchronize(s->ev[i + 1]));
		float ms = 0.0f;
		HNS_HIP(hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]));
		tot += ms;
	}
	*total_ms = (float)tot;
	*launches = s->timed_launches;
	return HNS_OK;
}

static int sim_advect_scalars(hns_sim* s, const float* sdf, bool coll, float dt, float inv_dx, void* stream) {
	std::vector<const float*> ins;
	std::vector<float*> outs;
	std::vector<int> which;
	for (size_t i = 0; i < s->names.size(); ++i) {
		if (s->names[i] == "collision_sdf") continue;  // HNanoSolver.cu:327
		ins.push_back(s->cur[i]);
		outs.push_back(s->nxt[i]);
		which.push_back((int)i);
	}
	HNS_TRY(hns_dev_advect_scalars(s->grid, s->vel, ins.data(), outs.data(), (int)ins.size(), sdf, coll, dt, inv_dx,
	                               stream));
	for (int i : which) std::swap(s->cur[i], s->nxt[i]);
	return HNS_OK;
}

// One substep in the order of reference HNanoSolver.cu:150-356, cut at the two points where it starts to need more input
// fields, so that the operator path can enqueue each part as soon as its inputs are on the device:
//   part A  needs velocity (+ collision_sdf)      collision, advect_vector, vorticity
//   part B  needs fuel/waste/temperature/flame     divergence, combustion, buoyancy, pressure solve, gradient subtraction, collision
//   part C  needs every advected float field       advect_scalars
// Round 6: without a collision field (and while a 16-byte-per-voxel array stays 32-bit addressable) part B opens with ONE launch for divergence + combustion +
// buoyancy (hns_divergence_combust_buoyancy: 60 B/voxel instead of 16 + 40 + 28) that leaves the four combustion fields as one 16-byte element per voxel in s->q4, and
// part C gathers those four from there (hns_advect_scalars_q4: a corner tap of the four is one gather, not four). Same expressions in the same order per voxel:
// bit-identical to the separate launches, which remain the path with a collision field and the hns_dev_* entry points.
namespace {
struct Substep {
	hns_sim* s;
	int iterations;
	float dt, voxel_size, inv_dx;
	const hns_combustion_params* params;
	int ci[4];
	bool coll, fused;
	const float* sdf;
	void* stream;
	hipEvent_t* stage_events = nullptr;  // six events of hns_sim_stage_timing, or null
	int mark(int k) {
		if (stage_events) HNS_HIP(hipEventRecord(stage_events[k], (hipStream_t)stream));
		return HNS_OK;
	}

	int prepare(hns_sim* sim, int iters, float dt_, float vs, const hns_combustion_params* prm, int has_collision, void* st) {
		s = sim;
		iterations = iters;
		dt = dt_;
		voxel_size = vs;
		inv_dx = 1.0f / vs;
		params = prm;
		stream = st;
		if (s->names.empty()) return fail(HNS_ERR_RUNTIME, "No float blocks found in input data.");  // :61-63
		const char* required[4] = {"fuel", "waste", "temperature", "flame"};                          // :193-201
		for (int c = 0; c < 4; ++c) {
			ci[c] = s->find(required[c]);
			if (ci[c] < 0) {
				set_error("Missing required input field for combustion: %s", required[c]);
				return HNS_ERR_RUNTIME;
			}
		}
		const int i_sdf = has_collision ? s->find("collision_sdf") : -1;  // :66-75
		coll = i_sdf >= 0;
		sdf = coll ? s->cur[i_sdf] : nullptr;
		size_t advected = 0;
		for (const std::string& n : s->names) advected += n != "collision_sdf";
		// (every leaf active: the pointwise kernels of the separate path run over ALL voxels, read-only ghost leaves included, and advection taps read them there)
		fused = !coll && s->q4 && s->grid->d_blk && s->grid->n_active == (uint64_t)s->grid->topo.n_leaves && !options().stencil_block.load() && hns_advect_q4_ok(s->grid) &&
		        advected - 4 <= 8 && options().fuse_pointwise.load();
		return HNS_OK;
	}
	int part_a() {
		hns_grid* g = s->grid;
		HNS_TRY(mark(0));
		if (coll) HNS_TRY(hns_dev_enforce_collision_boundaries(g, s->vel, sdf, voxel_size, stream));  // :153-157
		HNS_TRY(hns_dev_advect_vector(g, s->vel, s->adv, sdf, coll, dt, inv_dx, stream));  // :162-170
		if ((int)params->factorScale != 0) {  // :172-176. With (int)factorScale == 0 every vorticity-magnitude tap collapses onto the centre, the
			// gradient is 0, N = 0/(0+1e-5) = 0 and the kernel writes u + dt*(scale*0) = u: a bit-exact copy, skipped.
			HNS_TRY(hns_dev_vorticity_confinement(g, s->adv, s->tmp, dt, inv_dx,
			                                      params->vorticityScale, params->factorScale, stream));
			std::swap(s->adv, s->tmp);
		}
		return HNS_OK;
	}
	int part_b() {
		hns_grid* g = s->grid;
		HNS_TRY(mark(1));
		if (fused) {  // :181-234 in one launch; the four fields' new values live in s->q4 until part C (cur[] / nxt[] of those four are not touched here)
			HNS_TRY(hns_divergence_combust_buoyancy(g, s->adv, s->div, inv_dx, s->cur[ci[0]], s->cur[ci[1]], s->cur[ci[2]], s->cur[ci[3]], s->q4, s->tmp,
			                                        params->temperatureRelease, params->expansionRate, dt, params->ambientTemp, params->buoyancyStrength, stream));
			std::swap(s->adv, s->tmp);
		} else {
			HNS_TRY(hns_dev_divergence(g, s->adv, s->div, inv_dx, stream));  // :181-188
			HNS_TRY(hns_dev_combustion_oxygen(s->cur[ci[0]], s->cur[ci[1]], s->cur[ci[2]], s->div, s->cur[ci[3]], s->nxt[ci[0]], s->nxt[ci[1]],
			                                  s->nxt[ci[2]], s->nxt[ci[3]], params->temperatureRelease, params->expansionRate, s->n, stream));  // :211-221
			HNS_TRY(hns_dev_temperature_buoyancy(s->adv, s->nxt[ci[2]], s->adv, dt, params->ambientTemp, params->buoyancyStrength, s->n,
			                                     stream));  // :226-234 (temperature AFTER combustion)
			for (int c = 0; c < 4; ++c) std::swap(s->cur[ci[c]], s->nxt[ci[c]]);  // :239-246
		}
		HNS_TRY(mark(2));
		HNS_TRY(sim_pressure(s, iterations, voxel_size, omega_compute(voxel_size), stream));  // :256-272
		HNS_TRY(mark(3));
		HNS_TRY(hns_dev_subtract_pressure_gradient(g, s->adv, s->p_result, s->vel, sdf, coll,
		                                           inv_dx, stream));  // :278-289
		if (coll) HNS_TRY(hns_dev_enforce_collision_boundaries(g, s->vel, sdf, voxel_size, stream));  // :292-296
		return HNS_OK;
	}
	int part_c() {  // :321-356
		HNS_TRY(mark(4));
		if (!fused) return sim_advect_scalars(s, sdf, coll, dt, inv_dx, stream);
		std::vector<const float*> ins;
		std::vector<float*> outs;
		float* q4_out[4];
		for (int c = 0; c < 4; ++c) q4_out[c] = s->nxt[ci[c]];
		for (size_t i = 0; i < s->names.size(); ++i) {
			if (s->names[i] == "collision_sdf" || (int)i == ci[0] || (int)i == ci[1] || (int)i == ci[2] || (int)i == ci[3]) continue;  // :327
			ins.push_back(s->cur[i]);
			outs.push_back(s->nxt[i]);
		}
		HNS_TRY(hns_advect_scalars_q4(s->grid, s->vel, s->q4, q4_out, ins.data(), outs.data(), (int)ins.size(), dt, inv_dx, stream));
		for (size_t i = 0; i < s->names.size(); ++i)
			if (s->names[i] != "collision_sdf") std::swap(s->cur[i], s->nxt[i]);
		return HNS_OK;
	}
};
}  // namespace

extern "C" int hns_sim_substep(hns_sim* s, int iterations, float dt, float voxel_size, const hns_combustion_params* params, int has_collision,
                               void* stream) {
	if (!s || !params) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_substep: null argument");
	HNS_TRY(validate_step(voxel_size, dt, iterations, true));
	if (s->n == 0) return HNS_OK;  // HNanoSolver.cu:26-28
	Substep step;
	HNS_TRY(step.prepare(s, iterations, dt, voxel_size, params, has_collision, stream));
	// hns_sim_stage_timing: the same five brackets as the core substep -- {collision + advect_vector + vorticity, divergence + combustion + buoyancy (one launch when
	// fused), pressure loop, gradient subtraction + collision, advect_scalars}
	const bool staged = s->stage_timing && s->sev_used + 6 <= s->sev.size();
	step.stage_events = staged ? &s->sev[s->sev_used] : nullptr;
	HNS_TRY(step.part_a());
	HNS_TRY(step.part_b());
	HNS_TRY(step.part_c());
	if (staged) {
		HNS_HIP(hipEventRecord(step.stage_events[5], (hipStream_t)stream));
		s->sev_used += 6;
	}
	return HNS_OK;
}

extern "C" int hns_sim_core_substep(hns_sim* s, int iterations, float dt, float voxel_size, void* stream) {
	if (!s) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_core_substep: null sim");
	HNS_TRY(validate_step(voxel_size, dt, iterations, true));
	if (s->n == 0) return HNS_OK;
	const float inv_dx = 1.0f / voxel_size;
	hns_grid* g = s->grid;
	const bool staged = s->stage_timing && s->sev_used + 6 <= s->sev.size();
	hipEvent_t* se = staged ? &s->sev[s->sev_used] : nullptr;
	if (staged) HNS_HIP(hipEventRecord(se[0], (hipStream_t)stream));
	HNS_TRY(hns_dev_advect_vector(g, s->vel, s->adv, nullptr, 0, dt, inv_dx, stream));
	if (staged) HNS_HIP(hipEventRecord(se[1], (hipStream_t)stream));
	HNS_TRY(hns_dev_divergence(g, s->adv, s->div, inv_dx, stream));
	if (staged) HNS_HIP(hipEventRecord(se[2], (hipStream_t)stream));
	HNS_TRY(sim_pressure(s, iterations, voxel_size, omega_compute(voxel_size), stream));
	if (staged) HNS_HIP(hipEventRecord(se[3], (hipStream_t)stream));
	HNS_TRY(hns_dev_subtract_pressure_gradient(g, s->adv, s->p_result, s->vel, nullptr, 0,
	                                           inv_dx, stream));
	if (staged) HNS_HIP(hipEventRecord(se[4], (hipStream_t)stream));
	HNS_TRY(sim_advect_scalars(s, nullptr, false, dt, inv_dx, stream));
	if (staged) {
		HNS_HIP(hipEventRecord(se[5], (hipStream_t)stream));
		s->sev_used += 6;
	}
	return HNS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// drop-in operators (host pointers, synchronous, in place)
// ---------------------------------------------------------------------------------------------------------------

namespace {
struct FieldSplit {
	hns_field* velocity = nullptr;
	int n_vec3 = 0;
	std::vector<hns_field*> floats;
};

int split_fields(hns_field* fields, int n_fields, FieldSplit& out, const char* who) {
	if (n_fields < 0 || (n_fields > 0 && !fields)) {
		set_error("%s: null field array", who);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	for (int i = 0; i < n_fields; ++i) {
		if (fields[i].ncomp == 3) {
			if (!out.velocity) out.velocity = &fields[i];
			++out.n_vec3;
		} else if (fields[i].ncomp == 1) {
			out.floats.push_back(&fields[i]);
		} else {
			set_error("%s: field %d has ncomp %d (must be 1 or 3)", who, i, fields[i].ncomp);
			return HNS_ERR_INVALID_ARGUMENT;
		}
		if (!fields[i].name) {
			set_error("%s: field %d has no name", who, i);
			return HNS_ERR_INVALID_ARGUMENT;
		}
	}
	return HNS_OK;
}

// Persistent state across cooks (the reference's cook cache is an empty struct, SOP_HNanoSolver.hpp:60-64,
// and every cook pays cudaMallocAsync x 15+ and the matching frees, HNanoSolver.cu:87-133). An operator call borrows a
// device-resident hns_sim for its field-name list from the grid and returns it afterwards; the buffers live until the
// grid is destroyed or hns_grid_release_cache() is called. Two entries cover the usual "full solver + one single-field
// operator" pattern; a call that finds its entry lent out (concurrent cooks on one grid) works on a private sim.
// Option "cook_cache" = 0 disables the cache.
struct SimGuard {
	hns_sim* s = nullptr;
	~SimGuard() {
		if (!s) return;
		if (!s->cached) {
			hns_sim_destroy(s);
			return;
		}
		std::lock_guard<std::mutex> lock(s->grid->host_mutex);
		s->in_use = false;
	}
};

int make_sim(hns_grid* g, const FieldSplit& fs, SimGuard& guard, void* stream) {
	std::vector<const char*> names;
	for (hns_field* f : fs.floats) names.push_back(f->name);
	const bool use_cache = options().cook_cache.load() != 0;
	if (use_cache) {
		std::lock_guard<std::mutex> lock(g->host_mutex);
		for (hns_sim* c : g->sim_cache) {
			if (c->in_use || c->names.size() != names.size()) continue;
			bool same = true;
			for (size_t i = 0; i < names.size() && same; ++i) same = c->names[i] == names[i];
			if (!same) continue;
			c->in_use = true;
			guard.s = c;
			break;
		}
	}
	if (guard.s) {
		// A fresh sim starts zeroed and a substep only writes the active leaves; with an active prefix
		// (hns_grid_set_active_leaves) restore that guarantee for everything a kernel may read across a leaf face.
		if (g->n_active != (uint64_t)g->topo.n_leaves) {
			hns_sim* s = guard.s;
			const size_t bytes = sizeof(float) * (size_t)s->n;
			hipStream_t st = (hipStream_t)stream;
			HNS_HIP(hipMemsetAsync(s->div, 0, bytes, st));
			HNS_HIP(hipMemsetAsync(s->p_a, 0, bytes, st));
			HNS_HIP(hipMemsetAsync(s->p_b, 0, bytes, st));
			HNS_HIP(hipMemsetAsync(s->adv, 0, 3 * bytes, st));
			HNS_HIP(hipMemsetAsync(s->tmp, 0, 3 * bytes, st));
			for (float* q : s->nxt) HNS_HIP(hipMemsetAsync(q, 0, bytes, st));
		}
		return HNS_OK;
	}
	int err = HNS_OK;
	// every operator writes each buffer before it reads it when all leaves are active, so the new memory is not cleared
	guard.s = sim_create(g, names.data(), (int)names.size(), g->n_active != (uint64_t)g->topo.n_leaves, stream, &err);
	if (err != HNS_OK || !use_cache) return err;
	std::lock_guard<std::mutex> lock(g->host_mutex);
	if (g->sim_cache.size() >= 2) {  // evict the oldest entry that is not lent out
		for (size_t i = 0; i < g->sim_cache.size(); ++i)
			if (!g->sim_cache[i]->in_use) {
				g->sim_cache[i]->cached = false;
				hns_sim_destroy(g->sim_cache[i]);
				g->sim_cache.erase(g->sim_cache.begin() + (long)i);
				break;
			}
	}
	if (g->sim_cache.size() < 2) {
		guard.s->cached = guard.s->in_use = true;
		g->sim_cache.push_back(guard.s);
	}
	return HNS_OK;
}
}  // namespace

// hns_compute_sim's data movement. The reference uploads everything, runs, downloads everything (HNanoSolver.cu:87-133,
// 361-371). Here the fields go up in the order the substep consumes them, on a transfer stream of the sim's own, and each
// part of the substep is enqueued on the caller's stream as soon as its inputs are queued: advect_vector runs under the
// upload of the four combustion fields, the pressure solve under the upload of every other field (Substep::part_b),
// and advect_scalars under the download of the final velocity. Same kernels, same order per buffer; only the overlap
// differs. Option "cook_pipeline" = 0 falls back to upload-all / run / download-all.
// What a host array and the device buffer it was downloaded from have in common afterwards. Two strengths (hns_compute_sim_resident):
//  * the SAMPLE signature: element count and 4,096 evenly spread elements, hashed (FNV-1a over their bits). 16 KB read out of 67 - 201 MB: a tripwire
//    against handing in a different array, NOT a check of the promise -- an edit that misses the samples (an emitter added to a few leaves) passes.
//    resident[i] = 1 ("vouched") relies on the caller knowing what it changed, as the reference's SOP does (it adds its sources itself,
//    SOP_HNanoSolver.cpp:159-179, and must not flag a field it sourced into);
//  * the FULL digest (hns_digest.hpp): every element, an order-independent sum over 16-byte pieces -- taken on the DEVICE when a field is handed back (beside the
//    downloads) and on up to 8 HOST threads when the array comes in again (~5.5 ms for the 537 MB of a 256^3 cook, partly under the first kernels).
//    resident[i] = 2 ("checked"): an edit goes unnoticed only if the 64-bit digests collide (probability ~2^-64 for an edit that is not crafted against the digest); 14.8 ms per cook at 256^3 against 19.8 plain and 11.9 vouched.
// Neither is ever 0.
static uint64_t host_signature(const float* a, size_t count) {
	uint64_t h = 1469598103934665603ull ^ (uint64_t)count;
	const size_t samples = count < 4096 ? count : 4096;
	for (size_t i = 0; i < samples; ++i) {
		uint32_t bits;
		memcpy(&bits, a + (samples == count ? i : (size_t)(((unsigned __int128)i * count) / samples)), 4);
		h = (h ^ bits) * 1099511628211ull;
	}
	return h ? h : 1;
}
static uint64_t host_digest(const float* a, size_t count) {  // hns_digest.hpp: the number k_field_digest takes of the device copy of the same bits
	const size_t n_pieces = count / 4;
	constexpr size_t kChunk = (size_t)1 << 14;  // pieces per chunk (256 KiB)
	const size_t n_chunks = (n_pieces + kChunk - 1) / kChunk;
	std::vector<uint64_t> part(std::max<size_t>(1, n_chunks), 0);
	auto work = [&](size_t c0, size_t c1) {
		for (size_t c = c0; c < c1; ++c) {
			const size_t lo = c * kChunk, hi = std::min(n_pieces, lo + kChunk);
			uint64_t s0 = 0, s1 = 0;
			size_t i = lo;
			for (; i + 2 <= hi; i += 2) {  // two independent accumulators
				uint64_t w[4];
				memcpy(w, a + 4 * i, 32);
				s0 += hns_digest_piece(i, w[0], w[1]);
				s1 += hns_digest_piece(i + 1, w[2], w[3]);
			}
			for (; i < hi; ++i) {
				uint64_t w[2];
				memcpy(w, a + 4 * i, 16);
				s0 += hns_digest_piece(i, w[0], w[1]);
			}
			part[c] = s0 + s1;
		}
	};
	const size_t n_threads = std::min<size_t>(8, std::max<size_t>(1, n_chunks / 8));  // (2 MiB and more per thread: a thread costs ~30 us to start)
	if (n_threads <= 1) work(0, n_chunks);
	else {
		std::vector<std::thread> th;
		for (size_t t = 1; t < n_threads; ++t) th.emplace_back(work, n_chunks * t / n_threads, n_chunks * (t + 1) / n_threads);
		work(0, n_chunks / n_threads);
		for (auto& t : th) t.join();
	}
	uint64_t sum = 0;
	for (uint64_t p : part) sum += p;
	if (count & 3) {  // a zero-padded last piece
		uint32_t w[4] = {0, 0, 0, 0};
		memcpy(w, a + 4 * n_pieces, 4 * (count & 3));
		sum += hns_digest_piece(n_pieces, (uint64_t)w[0] | (uint64_t)w[1] << 32, (uint64_t)w[2] | (uint64_t)w[3] << 32);
	}
	return hns_digest_finish(sum, count);
}

// `resident` (hns_compute_sim_resident): per field of `fields` (by position), non-zero = the caller vouches that the host array still
// holds what the previous hns_compute_sim on this grid handed back for the block of that name. The field is then not uploaded -- if
// the device state lent to this call is the one that produced it and the array's signature still matches; uploaded as usual otherwise.
static int compute_sim_pipelined(hns_sim* s, FieldSplit& fs, int iterations, float dt, float voxel_size, const hns_combustion_params* params,
                                 int has_collision, void* stream, const std::vector<std::pair<const hns_field*, int>>& resident, int* skipped) {
	Substep step;
	const size_t count = (size_t)s->n;
	s->sig_cur.resize(s->names.size(), 0);
	s->dig_cur.resize(s->names.size(), 0);
	{  // (refused before anything is enqueued: the digest kernels have 16 accumulators)
		int checked = 0;
		for (auto& r : resident) checked += r.second >= 2 ? 1 : 0;
		if (checked > 16) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_compute_sim_resident: more than 16 CHECKED fields");
	}
	auto level_of = [&](const hns_field* f) {
		for (auto& r : resident)
			if (r.first == f) return r.second;
		return 0;
	};
	auto stays = [&](const hns_field* f) {
		const int level = level_of(f);
		if (!level) return false;
		uint64_t have = 0, have_full = 0;
		size_t n = count;
		if (f->ncomp == 3) have = s->sig_vel, have_full = s->dig_vel, n = 3 * count;
		else {
			const int k = s->find(f->name);
			if (k >= 0) have = s->sig_cur[(size_t)k], have_full = s->dig_cur[(size_t)k];
		}
		if (!have || have != host_signature(f->host, n)) return false;
		if (level >= 2 && (!have_full || have_full != host_digest(f->host, n))) return false;  // (no digest taken last cook: nothing to check against -> upload)
		if (skipped) ++*skipped;
		return true;
	};
	auto upload = [&](hns_field* f, void* on) { return stays(f) ? (int)HNS_OK : hns_sim_upload(s, f, 1, on); };
	// the full digest of a CHECKED field is taken ON THE DEVICE, of the buffer the host array is downloaded from (the same bits), behind the last kernel of the
	// substep and beside the downloads: one pass at memory speed instead of a second host pass over every array
	std::vector<std::pair<uint64_t*, size_t>> dig_slots;  // (where the digest goes, its element count), in the order of s->d_dig
	auto digest_on_device = [&](hipStream_t on) -> int {
		std::vector<const float*> bufs;
		if (level_of(fs.velocity) >= 2) bufs.push_back(s->vel), dig_slots.emplace_back(&s->dig_vel, 3 * count);
		for (hns_field* f : fs.floats) {
			const int k = s->find(f->name);
			if (k >= 0 && strcmp(f->name, "collision_sdf") && level_of(f) >= 2) bufs.push_back(s->cur[(size_t)k]), dig_slots.emplace_back(&s->dig_cur[(size_t)k], count);
		}
		if (bufs.empty()) return HNS_OK;
		if (!s->h_dig) HNS_HIP(hipHostMalloc((void**)&s->h_dig, sizeof(unsigned long long) * 16, hipHostMallocDefault));  // (host memory: no device synchronisation)
		HNS_HIP(hipMemsetAsync(s->d_dig, 0, sizeof(unsigned long long) * 16, on));
		for (size_t i = 0; i < bufs.size(); ++i) HNS_TRY(hns_field_digest(bufs[i], dig_slots[i].second, s->d_dig + i, on));
		HNS_HIP(hipMemcpyAsync(s->h_dig, s->d_dig, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, on));  // (on the cook's own stream, in front of its final synchronise: ADVICE r5)
		return HNS_OK;
	};
	auto sign = [&]() -> int {  // (the downloads and the digest kernels have completed: what the host arrays hold now is what vel / cur[] hold)
		s->sig_vel = host_signature(fs.velocity->host, 3 * count);
		s->dig_vel = 0;
		for (hns_field* f : fs.floats) {
			const int k = s->find(f->name);
			if (k < 0) continue;
			const bool sdf = !strcmp(f->name, "collision_sdf");  // (the SDF comes back zeroed, the device keeps it)
			s->sig_cur[(size_t)k] = sdf ? 0 : host_signature(f->host, count);
			s->dig_cur[(size_t)k] = 0;
		}
		if (!dig_slots.empty()) {
			for (size_t i = 0; i < dig_slots.size(); ++i) *dig_slots[i].first = hns_digest_finish(s->h_dig[i], dig_slots[i].second);  // (the stream the copy ran on has been synchronised)
		}
		return HNS_OK;
	};
	if (!options().cook_pipeline.load()) {
		std::vector<hns_field> all;
		HNS_TRY(upload(fs.velocity, stream));
		all.push_back(*fs.velocity);
		for (hns_field* f : fs.floats) {
			HNS_TRY(upload(f, stream));
			all.push_back(*f);
		}
		HNS_TRY(hns_sim_substep(s, iterations, dt, voxel_size, params, has_collision, stream));
		HNS_TRY(digest_on_device((hipStream_t)stream));
		HNS_TRY(hns_sim_download(s, all.data(), (int)all.size(), stream));
		return sign();
	}
	if (!s->xfer) {
		HNS_HIP(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
		for (hipEvent_t& e : s->xev) HNS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	hipStream_t st = (hipStream_t)stream, xf = s->xfer;
	HNS_TRY(step.prepare(s, iterations, dt, voxel_size, params, has_collision, stream));
	auto is_combustion_field = [](const char* n) { return !strcmp(n, "fuel") || !strcmp(n, "waste") || !strcmp(n, "temperature") || !strcmp(n, "flame"); };
	auto handoff = [&](hipEvent_t e, hipStream_t from, hipStream_t to) -> int {
		HNS_HIP(hipEventRecord(e, from));
		HNS_HIP(hipStreamWaitEvent(to, e, 0));
		return HNS_OK;
	};
	HNS_TRY(handoff(s->xev[0], st, xf));  // transfers start after whatever the caller's stream holds
	if (step.coll)
		for (hns_field* f : fs.floats)
			if (!strcmp(f->name, "collision_sdf")) HNS_TRY(hns_sim_upload(s, f, 1, xf));
	HNS_TRY(upload(fs.velocity, xf));
	HNS_TRY(handoff(s->xev[1], xf, st));
	HNS_TRY(step.part_a());
	for (hns_field* f : fs.floats)
		if (is_combustion_field(f->name)) HNS_TRY(upload(f, xf));
	HNS_TRY(handoff(s->xev[2], xf, st));
	HNS_TRY(step.part_b());  // the solve runs while every remaining field is still on its way
	HNS_HIP(hipEventRecord(s->xev[3], st));  // s->vel is final here
	for (hns_field* f : fs.floats)  // "collision_sdf" went up first if this call uses it; if not, nothing reads it and it returns zeroed
		if (!is_combustion_field(f->name) && strcmp(f->name, "collision_sdf") != 0) HNS_TRY(upload(f, xf));
	HNS_TRY(handoff(s->xev[4], xf, st));
	HNS_TRY(step.part_c());
	HNS_TRY(digest_on_device(st));  // (every field is final on the device here; the kernels run beside the downloads)
	HNS_HIP(hipStreamWaitEvent(xf, s->xev[3], 0));
	HNS_HIP(hipMemcpyAsync(fs.velocity->host, s->vel, sizeof(float) * 3 * (size_t)s->n, hipMemcpyDeviceToHost, xf));
	HNS_TRY(handoff(s->xev[0], st, xf));
	std::vector<hns_field> outs;
	for (hns_field* f : fs.floats)
		if (strcmp(f->name, "collision_sdf") != 0) outs.push_back(*f);  // the caller's SDF array comes back zeroed (hns_compute_sim): no bytes to fetch
	HNS_TRY(hns_sim_download(s, outs.data(), (int)outs.size(), xf));  // synchronises xf
	HNS_HIP(hipStreamSynchronize(st));
	return sign();
}

extern "C" int hns_compute_sim(hns_grid* g, hns_field* fields, int n_fields, int iterations, float dt, float voxel_size,
                               const hns_combustion_params* params, int has_collision, void* stream) {
	return hns_compute_sim_resident(g, fields, n_fields, nullptr, nullptr, iterations, dt, voxel_size, params, has_collision, stream);
}

extern "C" int hns_compute_sim_resident(hns_grid* g, hns_field* fields, int n_fields, const unsigned char* resident, int* uploads_skipped, int iterations, float dt,
                                        float voxel_size, const hns_combustion_params* params, int has_collision, void* stream) {
	if (uploads_skipped) *uploads_skipped = 0;
	HNS_TRY(validate_step(voxel_size, dt, iterations, true));
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "Invalid grid handle provided (null grid).");  // HNanoSolver.cu:21-23
	if (!params) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_compute_sim: null combustion params");
	if (hns_grid_voxel_count(g) == 0) return HNS_OK;  // :26-28
	FieldSplit fs;
	HNS_TRY(split_fields(fields, n_fields, fs, "hns_compute_sim"));
	if (fs.n_vec3 != 1) {  // :42-45
		set_error("Expected exactly one Vec3f block (velocity), found %d", fs.n_vec3);
		return HNS_ERR_RUNTIME;
	}
	if (!fs.velocity->host) return fail(HNS_ERR_RUNTIME, "Host velocity data pointer is null");  // :48-51
	if (fs.floats.empty()) return fail(HNS_ERR_RUNTIME, "No float blocks found in input data.");  // :61-63
	for (hns_field* f : fs.floats)
		if (!f->host) {
			set_error("Host float data pointer is null for block: %s", f->name);  // :80-82
			return HNS_ERR_RUNTIME;
		}
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_compute_sim: grid has no device tables (there is no CPU fallback)");
	std::vector<std::pair<const hns_field*, int>> res;
	if (resident)
		for (int i = 0; i < n_fields; ++i)
			if (resident[i]) res.emplace_back(&fields[i], (int)resident[i]);
	SimGuard guard;
	HNS_TRY(make_sim(g, fs, guard, stream));
	if (int rc = compute_sim_pipelined(guard.s, fs, iterations, dt, voxel_size, params, has_collision, stream, res, uploads_skipped)) {
		guard.s->sig_vel = guard.s->dig_vel = 0;  // whatever the buffers hold now, nobody was handed it
		std::fill(guard.s->sig_cur.begin(), guard.s->sig_cur.end(), 0);
		std::fill(guard.s->dig_cur.begin(), guard.s->dig_cur.end(), 0);
		// copies on the transfer stream and kernels on the caller's may still be queued: let them finish before the
		// guard hands the buffers on (and before the caller reuses its host arrays)
		if (guard.s->xfer) (void)hipStreamSynchronize(guard.s->xfer);
		(void)hipStreamSynchronize((hipStream_t)stream);
		return rc;
	}
	// The reference copies every float block back from its OUTPUT buffer; "collision_sdf" is never advected, so its
	// output buffer is still the memset zeros and the caller's SDF array comes back zeroed (HNanoSolver.cu:115-117,327,364-369).
	for (hns_field* f : fs.floats)
		if (strcmp(f->name, "collision_sdf") == 0) memset(f->host, 0, sizeof(float) * (size_t)hns_grid_voxel_count(g));
	return HNS_OK;
}

extern "C" int hns_advect_index_grid(hns_grid* g, hns_field* fields, int n_fields, float dt, float voxel_size, void* stream) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_advect_index_grid: null grid");
	FieldSplit fs;
	HNS_TRY(split_fields(fields, n_fields, fs, "hns_advect_index_grid"));
	if (fs.n_vec3 != 1) return fail(HNS_ERR_RUNTIME, "Expected exactly one Vec3f block (velocity)");  // Advection.cu:19-21
	if (fs.floats.empty()) return fail(HNS_ERR_RUNTIME, "No float blocks found");                     // :27-29
	if (!fs.velocity->host) return fail(HNS_ERR_RUNTIME, "Velocity data not found");                  // :32-34
	for (hns_field* f : fs.floats)
		if (!f->host) {
			set_error("Block '%s' not found or type mismatch", f->name);  // :46-48
			return HNS_ERR_RUNTIME;
		}
	if (hns_grid_voxel_count(g) == 0) return HNS_OK;
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_advect_index_grid: grid has no device tables (there is no CPU fallback)");
	SimGuard guard;
	HNS_TRY(make_sim(g, fs, guard, stream));
	hns_sim* s = guard.s;
	HNS_TRY(hns_sim_upload(s, fields, n_fields, stream));
	const float inv_dx = 1.0f / voxel_size;
	for (size_t i = 0; i < s->names.size(); ++i) {  // one advect_scalar per float block (Advection.cu:88-91)
		HNS_TRY(hns_dev_advect_scalar(g, s->vel, s->cur[i], s->nxt[i], nullptr, 0, dt, inv_dx, stream));
		std::swap(s->cur[i], s->nxt[i]);
	}
	std::vector<hns_field> outs;
	for (hns_field* f : fs.floats) outs.push_back(*f);  // velocity is not copied back (Advection.cu:94-96)
	return hns_sim_download(s, outs.data(), (int)outs.size(), stream);
}

extern "C" int hns_advect_index_grid_velocity(hns_grid* g, hns_field* fields, int n_fields, float dt, float voxel_size, void* stream) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_advect_index_grid_velocity: null grid");
	FieldSplit fs;
	HNS_TRY(split_fields(fields, n_fields, fs, "hns_advect_index_grid_velocity"));
	if (fs.n_vec3 != 1) return fail(HNS_ERR_RUNTIME, "Expected exactly one Vec3f block (velocity)");  // Advection.cu:119-121
	if (!fs.velocity->host) return fail(HNS_ERR_RUNTIME, "Velocity data not found");
	if (hns_grid_voxel_count(g) == 0) return HNS_OK;
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_advect_index_grid_velocity: grid has no device tables (there is no CPU fallback)");
	SimGuard guard;
	FieldSplit only_vel;
	only_vel.velocity = fs.velocity;
	only_vel.n_vec3 = 1;
	HNS_TRY(make_sim(g, only_vel, guard, stream));
	hns_sim* s = guard.s;
	HNS_TRY(hns_sim_upload(s, fs.velocity, 1, stream));
	HNS_TRY(hns_dev_advect_vector(g, s->vel, s->adv, nullptr, 0, dt, 1.0f / voxel_size, stream));
	std::swap(s->vel, s->adv);
	return hns_sim_download(s, fs.velocity, 1, stream);
}

extern "C" int hns_project_non_divergent(hns_grid* g, hns_field* fields, int n_fields, uint64_t iterations, float voxel_size, void* stream) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_project_non_divergent: null grid");
	FieldSplit fs;
	HNS_TRY(split_fields(fields, n_fields, fs, "hns_project_non_divergent"));
	if (fs.n_vec3 != 1) return fail(HNS_ERR_RUNTIME, "Expected exactly one Vec3f block (velocity)");  // PressureProjection.cu:14-17
	if (!fs.velocity->host) return fail(HNS_ERR_RUNTIME, "Velocity data not found");
	if (voxel_size <= 0.0f) return fail(HNS_ERR_INVALID_ARGUMENT, "voxelSize must be positive.");
	if (iterations > 0x7fffffffull) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_project_non_divergent: iteration count too large");
	if (hns_grid_voxel_count(g) == 0) return HNS_OK;
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_project_non_divergent: grid has no device tables (there is no CPU fallback)");
	SimGuard guard;
	FieldSplit only_vel;
	only_vel.velocity = fs.velocity;
	only_vel.n_vec3 = 1;
	HNS_TRY(make_sim(g, only_vel, guard, stream));
	hns_sim* s = guard.s;
	HNS_TRY(hns_sim_upload(s, fs.velocity, 1, stream));
	const float inv_dx = 1.0f / voxel_size;
	HNS_TRY(hns_dev_divergence(g, s->vel, s->div, inv_dx, stream));              // :48
	HNS_TRY(sim_pressure(s, (int)iterations, voxel_size, omega_project(voxel_size), stream));            // :51-60 (0 iterations leaves p = 0)
	HNS_TRY(hns_dev_subtract_pressure_gradient(g, s->vel, s->p_result, s->vel, nullptr, 0,
	                                           inv_dx, stream));  // :64, in place
	return hns_sim_download(s, fs.velocity, 1, stream);
}

extern "C" int hns_divergence(hns_grid* g, hns_field* fields, int n_fields, float voxel_size, void* stream) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_divergence: null grid");
	FieldSplit fs;
	HNS_TRY(split_fields(fields, n_fields, fs, "hns_divergence"));
	if (fs.n_vec3 != 1) return fail(HNS_ERR_RUNTIME, "Expected exactly one Vec3f block (velocity)");  // PressureProjection.cu:85-88
	if (!fs.velocity->host) return fail(HNS_ERR_RUNTIME, "Velocity data not found");
	hns_field* out = nullptr;
	for (hns_field* f : fs.floats)
		if (strcmp(f->name, "divergence") == 0) out = f;  // :91
	if (!out || !out->host) return fail(HNS_ERR_RUNTIME, "hns_divergence: no float block named 'divergence' to receive the result");
	if (voxel_size <= 0.0f) return fail(HNS_ERR_INVALID_ARGUMENT, "voxelSize must be positive.");
	if (hns_grid_voxel_count(g) == 0) return HNS_OK;
	if (!g->on_device) return fail(HNS_ERR_NO_DEVICE, "hns_divergence: grid has no device tables (there is no CPU fallback)");
	SimGuard guard;
	FieldSplit only_vel;
	only_vel.velocity = fs.velocity;
	only_vel.n_vec3 = 1;
	HNS_TRY(make_sim(g, only_vel, guard, stream));
	hns_sim* s = guard.s;
	HNS_TRY(hns_sim_upload(s, fs.velocity, 1, stream));
	HNS_TRY(hns_dev_divergence(g, s->vel, s->div, 1.0f / voxel_size, stream));
	HNS_HIP(hipMemcpyAsync(out->host, s->div, sizeof(float) * (size_t)s->n, hipMemcpyDeviceToHost, (hipStream_t)stream));
	HNS_HIP(hipStreamSynchronize((hipStream_t)stream));
	return HNS_OK;
}
