// hns_api.hip -- the device-resident simulation state and the drop-in operators of
// include/hns.h. The launch orders follow the reference's host drivers (reference src/Cuda/HNanoSolver.cu:150-356,
// src/Cuda/PressureProjection.cu:43-66, src/Cuda/Advection.cu:76-91,148-155); what differs is that fields can stay
// resident across substeps and that nothing is allocated inside a substep.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hns_internal.hpp"

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
	d.blk = (const int*)d_blk;
	d.hash_mask = topo.hash_mask;
	d.n_leaves = (int)topo.n_leaves;
	d.n_active = (int)n_active;
	d.oob = (int)outside_element;
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
	float* tmp = nullptr;  // out-of-place vorticity target
	float* div = nullptr;
	float* p_a = nullptr;
	float* p_b = nullptr;
	float* p_result = nullptr;  // whichever of p_a/p_b holds the last solve
	// optional hipEvent bracketing of the pressure hot loop (hns_sim_timing), on the stream the kernels run on
	bool timing = false;
	std::vector<hipEvent_t> ev;  // start/stop pairs
	size_t ev_used = 0;
	long long timed_launches = 0;
	int find(const char* name) const {
		for (size_t i = 0; i < names.size(); ++i)
			if (names[i] == name) return (int)i;
		return -1;
	}
};

static int sim_alloc(float** p, uint64_t n) {
	HNS_HIP(hipMalloc((void**)p, sizeof(float) * (size_t)(n ? n : 1)));
	HNS_HIP(hipMemset(*p, 0, sizeof(float) * (size_t)(n ? n : 1)));
	return HNS_OK;
}

extern "C" void hns_sim_destroy(hns_sim* s) {
	if (!s) return;
	for (float* p : s->cur) hipFree(p);
	for (float* p : s->nxt) hipFree(p);
	hipFree(s->vel);
	hipFree(s->adv);
	hipFree(s->tmp);
	hipFree(s->div);
	hipFree(s->p_a);
	hipFree(s->p_b);
	for (hipEvent_t e : s->ev) hipEventDestroy(e);
	delete s;
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
	hns_sim* s = new hns_sim;
	s->grid = g;
	s->n = hns_grid_voxel_count(g);
	auto alloc_all = [&]() -> int {
		for (int i = 0; i < n_float; ++i) {
			if (!float_names[i]) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_create: null field name");
			if (s->find(float_names[i]) >= 0) {
				set_error("hns_sim_create: duplicate field name '%s'", float_names[i]);
				return HNS_ERR_INVALID_ARGUMENT;
			}
			s->names.push_back(float_names[i]);
			s->cur.push_back(nullptr);
			s->nxt.push_back(nullptr);
			HNS_TRY(sim_alloc(&s->cur.back(), s->n));
			HNS_TRY(sim_alloc(&s->nxt.back(), s->n));
		}
		HNS_TRY(sim_alloc(&s->vel, 3 * s->n));
		HNS_TRY(sim_alloc(&s->adv, 3 * s->n));
		HNS_TRY(sim_alloc(&s->tmp, 3 * s->n));
		HNS_TRY(sim_alloc(&s->div, s->n));
		HNS_TRY(sim_alloc(&s->p_a, s->n));
		HNS_TRY(sim_alloc(&s->p_b, s->n));
		s->p_result = s->p_a;
		return HNS_OK;
	};
	rc = alloc_all();
	if (rc != HNS_OK) {
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
			HNS_HIP(hipMemcpyAsync(s->vel, f.host, sizeof(float) * 3 * (size_t)s->n, hipMemcpyHostToDevice, st));
		} else if (f.ncomp == 1) {
			const int k = f.name ? s->find(f.name) : -1;
			if (k < 0) {
				set_error("hns_sim_upload: no float field named '%s' in this sim", f.name ? f.name : "?");
				return HNS_ERR_RUNTIME;
			}
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
	HNS_HIP(hipMemsetAsync(s->p_a, 0, sizeof(float) * (size_t)s->n, (hipStream_t)stream));  // never warm-started (HNanoSolver.cu:113)
	int in_b = 0;
	const bool timed = s->timing && s->ev_used + 2 <= s->ev.size();
	if (timed) HNS_HIP(hipEventRecord(s->ev[s->ev_used], (hipStream_t)stream));
	HNS_TRY(hns_dev_rbgs_iterate(s->grid, s->div, s->p_a, s->p_b, voxel_size, omega, iterations, &in_b, stream));
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

extern "C" int hns_sim_substep(hns_sim* s, int iterations, float dt, float voxel_size, const hns_combustion_params* params, int has_collision,
                               void* stream) {
	if (!s || !params) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_substep: null argument");
	HNS_TRY(validate_step(voxel_size, dt, iterations, true));
	if (s->n == 0) return HNS_OK;  // HNanoSolver.cu:26-28
	if (s->names.empty()) return fail(HNS_ERR_RUNTIME, "No float blocks found in input data.");  // :61-63
	const char* required[4] = {"fuel", "waste", "temperature", "flame"};                          // :193-201
	int ci[4];
	for (int c = 0; c < 4; ++c) {
		ci[c] = s->find(required[c]);
		if (ci[c] < 0) {
			set_error("Missing required input field for combustion: %s", required[c]);
			return HNS_ERR_RUNTIME;
		}
	}
	const int i_sdf = has_collision ? s->find("collision_sdf") : -1;  // :66-75
	const bool coll = i_sdf >= 0;
	const float* sdf = coll ? s->cur[i_sdf] : nullptr;
	const float inv_dx = 1.0f / voxel_size;
	hns_grid* g = s->grid;

	if (coll) HNS_TRY(hns_dev_enforce_collision_boundaries(g, s->vel, sdf, voxel_size, stream));  // :153-157
	HNS_TRY(hns_dev_advect_vector(g, s->vel, s->adv, sdf, coll, dt, inv_dx, stream));  // :162-170
	if ((int)params->factorScale != 0) {  // :172-176. With (int)factorScale == 0 every vorticity-magnitude tap collapses onto the centre, the
		// gradient is 0, N = 0/(0+1e-5) = 0 and the kernel writes u + dt*(scale*0) = u: a bit-exact copy, skipped.
		HNS_TRY(hns_dev_vorticity_confinement(g, s->adv, s->tmp, dt, inv_dx,
		                                      params->vorticityScale, params->factorScale, stream));
		std::swap(s->adv, s->tmp);
	}
	HNS_TRY(hns_dev_divergence(g, s->adv, s->div, inv_dx, stream));  // :181-188
	HNS_TRY(hns_dev_combustion_oxygen(s->cur[ci[0]], s->cur[ci[1]], s->cur[ci[2]], s->div, s->cur[ci[3]], s->nxt[ci[0]], s->nxt[ci[1]],
	                                  s->nxt[ci[2]], s->nxt[ci[3]], params->temperatureRelease, params->expansionRate, s->n, stream));  // :211-221
	HNS_TRY(hns_dev_temperature_buoyancy(s->adv, s->nxt[ci[2]], s->adv, dt, params->ambientTemp, params->buoyancyStrength, s->n,
	                                     stream));  // :226-234 (temperature AFTER combustion)
	for (int c = 0; c < 4; ++c) std::swap(s->cur[ci[c]], s->nxt[ci[c]]);  // :239-246
	HNS_TRY(sim_pressure(s, iterations, voxel_size, omega_compute(voxel_size), stream));  // :256-272
	HNS_TRY(hns_dev_subtract_pressure_gradient(g, s->adv, s->p_result, s->vel, sdf, coll,
	                                           inv_dx, stream));  // :278-289
	if (coll) HNS_TRY(hns_dev_enforce_collision_boundaries(g, s->vel, sdf, voxel_size, stream));  // :292-296
	return sim_advect_scalars(s, sdf, coll, dt, inv_dx, stream);  // :321-356
}

extern "C" int hns_sim_core_substep(hns_sim* s, int iterations, float dt, float voxel_size, void* stream) {
	if (!s) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_sim_core_substep: null sim");
	HNS_TRY(validate_step(voxel_size, dt, iterations, true));
	if (s->n == 0) return HNS_OK;
	const float inv_dx = 1.0f / voxel_size;
	hns_grid* g = s->grid;
	HNS_TRY(hns_dev_advect_vector(g, s->vel, s->adv, nullptr, 0, dt, inv_dx, stream));
	HNS_TRY(hns_dev_divergence(g, s->adv, s->div, inv_dx, stream));
	HNS_TRY(sim_pressure(s, iterations, voxel_size, omega_compute(voxel_size), stream));
	HNS_TRY(hns_dev_subtract_pressure_gradient(g, s->adv, s->p_result, s->vel, nullptr, 0,
	                                           inv_dx, stream));
	return sim_advect_scalars(s, nullptr, false, dt, inv_dx, stream);
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

struct SimGuard {
	hns_sim* s = nullptr;
	~SimGuard() { hns_sim_destroy(s); }
};

int make_sim(hns_grid* g, const FieldSplit& fs, SimGuard& guard) {
	std::vector<const char*> names;
	for (hns_field* f : fs.floats) names.push_back(f->name);
	int err = HNS_OK;
	guard.s = hns_sim_create(g, names.data(), (int)names.size(), &err);
	return err;
}
}  // namespace

extern "C" int hns_compute_sim(hns_grid* g, hns_field* fields, int n_fields, int iterations, float dt, float voxel_size,
                               const hns_combustion_params* params, int has_collision, void* stream) {
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
	SimGuard guard;
	HNS_TRY(make_sim(g, fs, guard));
	HNS_TRY(hns_sim_upload(guard.s, fields, n_fields, stream));
	HNS_TRY(hns_sim_substep(guard.s, iterations, dt, voxel_size, params, has_collision, stream));
	HNS_TRY(hns_sim_download(guard.s, fields, n_fields, stream));
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
	HNS_TRY(make_sim(g, fs, guard));
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
	HNS_TRY(make_sim(g, only_vel, guard));
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
	HNS_TRY(make_sim(g, only_vel, guard));
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
	HNS_TRY(make_sim(g, only_vel, guard));
	hns_sim* s = guard.s;
	HNS_TRY(hns_sim_upload(s, fs.velocity, 1, stream));
	HNS_TRY(hns_dev_divergence(g, s->vel, s->div, 1.0f / voxel_size, stream));
	HNS_HIP(hipMemcpyAsync(out->host, s->div, sizeof(float) * (size_t)s->n, hipMemcpyDeviceToHost, (hipStream_t)stream));
	HNS_HIP(hipStreamSynchronize((hipStream_t)stream));
	return HNS_OK;
}
