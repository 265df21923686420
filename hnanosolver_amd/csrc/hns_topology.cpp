// hns_topology.cpp -- host-side index-grid construction (no device code).
//
// The reference builds a NanoVDB NanoGrid<ValueOnIndex> on the GPU every cook (create_index_grid, reference
// src/Cuda/HNanoSolver.cu:375-384 -> externals/nanovdb/tools/cuda/PointsToGrid.cuh:511-1064: ~25 launches + 8 CUB
// sorts) and then walks root->upper->lower->leaf on every stencil tap. Its domain is always leaf-dense
// (src/Utils/GridBuilder.hpp:156-166,229), so the only information in that tree is "which 8^3 leaves exist and in
// which order". This file keeps exactly that: a leaf-origin table in the caller's order, a 27-neighbour table per
// leaf and an origin hash, so a tap costs one table read instead of three dependent node loads.
#include <cstdlib>
#include <cstring>
#include <thread>

#include "hns_internal.hpp"

namespace hns {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
}

Options& options() {
	static Options o;
	return o;
}

uint32_t hash_origin(int32_t x, int32_t y, int32_t z) {
	// leaf coordinates (origin >> 3) mixed with three odd 32-bit constants; the same function runs on the device
	uint32_t h = (uint32_t)(x >> 3) * 0x9E3779B1u;
	h ^= (uint32_t)(y >> 3) * 0x85EBCA77u;
	h ^= (uint32_t)(z >> 3) * 0xC2B2AE3Du;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	h ^= h >> 12;
	return h;
}

int64_t Topology::find_leaf(int32_t ox, int32_t oy, int32_t oz) const {
	if (n_leaves == 0) return -1;
	uint32_t s = hash_origin(ox, oy, oz) & hash_mask;
	for (;;) {
		const int32_t l = hash[s];
		if (l < 0) return -1;
		const int32_t* o = &origins[4 * (size_t)l];
		if (o[0] == ox && o[1] == oy && o[2] == oz) return l;
		s = (s + 1) & hash_mask;
	}
}

uint64_t Topology::offset(int32_t i, int32_t j, int32_t k) const {
	const int64_t l = find_leaf(i & ~7, j & ~7, k & ~7);
	if (l < 0) return 0;
	return (uint64_t)l * 512u + (uint64_t)(((i & 7) << 6) | ((j & 7) << 3) | (k & 7)) + 1u;
}

int Topology::prepare(const int32_t* leaf_origins_xyz, int64_t n) {
	if (n < 0 || n > (int64_t(1) << 22)) {
		set_error("hns_grid: %lld leaves exceeds the 2^22-leaf (2^31-voxel) limit of 32-bit voxel indices", (long long)n);
		return HNS_ERR_TOPOLOGY;
	}
	n_leaves = n;
	have_tables = false;
	origins.assign((size_t)n * 4, 0);
	uint32_t size = 16;
	while ((int64_t)size < 2 * n + 2) size <<= 1;
	hash_mask = size - 1;
	for (int64_t l = 0; l < n; ++l) {
		const int32_t ox = leaf_origins_xyz[3 * l], oy = leaf_origins_xyz[3 * l + 1], oz = leaf_origins_xyz[3 * l + 2];
		if ((ox & 7) || (oy & 7) || (oz & 7)) {
			set_error("hns_grid: leaf %lld origin (%d,%d,%d) is not 8-aligned: coordinates are not leaf-dense", (long long)l, ox, oy, oz);
			return HNS_ERR_TOPOLOGY;
		}
		origins[4 * l] = ox;
		origins[4 * l + 1] = oy;
		origins[4 * l + 2] = oz;
	}
	return HNS_OK;
}

int Topology::build_tables() {
	const int64_t n = n_leaves;
	hash.assign((size_t)hash_mask + 1, -1);
	for (int64_t l = 0; l < n; ++l) {
		const int32_t ox = origins[4 * l], oy = origins[4 * l + 1], oz = origins[4 * l + 2];
		uint32_t s = hash_origin(ox, oy, oz) & hash_mask;
		while (hash[s] >= 0) {
			const int32_t* o = &origins[4 * (size_t)hash[s]];
			if (o[0] == ox && o[1] == oy && o[2] == oz) {
				set_error("hns_grid: leaf origin (%d,%d,%d) appears twice (leaves %d and %lld)", ox, oy, oz, hash[s], (long long)l);
				return HNS_ERR_TOPOLOGY;
			}
			s = (s + 1) & hash_mask;
		}
		hash[s] = (int32_t)l;
	}
	nbr27.assign((size_t)n * 27, -1);
	for (int64_t l = 0; l < n; ++l) {
		const int32_t* o = &origins[4 * (size_t)l];
		for (int dx = -1; dx <= 1; ++dx)
			for (int dy = -1; dy <= 1; ++dy)
				for (int dz = -1; dz <= 1; ++dz) {
					// int64 so that origins at the int32 edge cannot wrap into a valid neighbour
					const int64_t nx = (int64_t)o[0] + 8 * dx, ny = (int64_t)o[1] + 8 * dy, nz = (int64_t)o[2] + 8 * dz;
					int64_t nb = -1;
					if (nx >= INT32_MIN && nx <= INT32_MAX && ny >= INT32_MIN && ny <= INT32_MAX && nz >= INT32_MIN && nz <= INT32_MAX)
						nb = find_leaf((int32_t)nx, (int32_t)ny, (int32_t)nz);
					nbr27[(size_t)l * 27 + (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)] = (int32_t)nb;
				}
	}
	have_tables = true;
	return HNS_OK;
}

}  // namespace hns

using namespace hns;

extern "C" {

const char* hns_last_error(void) { return g_last_error.c_str(); }
int hns_version(void) { return HNS_VERSION; }

namespace {
struct OptionDesc {
	const char* name;
	std::atomic<int> Options::*field;
	const char* const* words;  // value words, index = stored value; nullptr: a non-negative integer
};
const char* const kWordsRbgs[] = {"auto", "color", nullptr};
const char* const kWordsAdvect[] = {"auto", "generic", nullptr};
const char* const kWordsStencil[] = {"auto", "block", nullptr};
const char* const kWordsSchedule[] = {"auto", "linear", nullptr};
const char* const kWordsBool[] = {"0", "1", nullptr};
const char* const kWordsMirror[] = {"0", "1", "guarded", nullptr};
const char* const kWordsUnsplit[] = {"0", "1", "always", nullptr};
const char* const kWordsDivergence[] = {"auto", "row", "coalesced", "zpair", nullptr};
const OptionDesc kOptions[] = {
    {"rbgs", &Options::rbgs, kWordsRbgs},
    {"advect", &Options::advect_generic, kWordsAdvect},
    {"stencil", &Options::stencil_block, kWordsStencil},
    {"schedule", &Options::schedule, kWordsSchedule},
    {"cook_cache", &Options::cook_cache, kWordsBool},
    {"cook_pipeline", &Options::cook_pipeline, kWordsBool},
    {"divergence", &Options::divergence_form, kWordsDivergence},
    {"fuse", &Options::fuse_pointwise, kWordsBool},
    {"sor_block_lb", &Options::sor_block_lb, nullptr},
    {"dist_wire_us", &Options::dist_wire_us, nullptr},
    {"dist_mirror", &Options::dist_mirror, kWordsMirror},
    {"dist_unsplit", &Options::dist_unsplit, kWordsUnsplit},
};
const Options kDefaults;
}  // namespace

int hns_set_option(const char* name, const char* value) {
	if (!name) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_set_option: null name");
	for (const OptionDesc& d : kOptions) {
		if (strcmp(d.name, name) != 0) continue;
		if (!value) {  // back to the default
			(options().*d.field).store((kDefaults.*d.field).load());
			return HNS_OK;
		}
		if (!d.words) {
			char* end = nullptr;
			const long v = strtol(value, &end, 10);
			if (end == value || *end || v < 0 || v > 1 << 20) break;
			(options().*d.field).store((int)v);
			return HNS_OK;
		}
		for (int i = 0; d.words[i]; ++i)
			if (strcmp(d.words[i], value) == 0) {
				(options().*d.field).store(i);
				return HNS_OK;
			}
		break;
	}
	set_error("hns_set_option: unknown option or value: %s = %s", name, value ? value : "(default)");
	return HNS_ERR_INVALID_ARGUMENT;
}

const char* hns_get_option(const char* name) {
	static thread_local char buf[16];
	if (!name) return nullptr;
	for (const OptionDesc& d : kOptions) {
		if (strcmp(d.name, name) != 0) continue;
		const int v = (options().*d.field).load();
		if (d.words) return d.words[v];
		snprintf(buf, sizeof(buf), "%d", v);
		return buf;
	}
	return nullptr;
}

static hns_grid* grid_from_origins(const int32_t* origins, uint64_t n_leaves, float voxel_size, unsigned flags, int* err) {
	int rc = HNS_OK;
	hns_grid* g = new hns_grid;
	g->voxel_size = voxel_size;
	rc = g->topo.prepare(origins, (int64_t)n_leaves);
	if (rc == HNS_OK) {
		g->n_active = n_leaves;
		rc = (flags & HNS_GRID_HOST_ONLY) ? g->topo.build_tables() : hns_grid_upload(g);
	}
	if (rc != HNS_OK) {
		hns_grid_free_device(g);
		delete g;
		g = nullptr;
	}
	if (err) *err = rc;
	return g;
}

hns_grid* hns_grid_create_from_leaves(const int32_t* leaf_origins_xyz, uint64_t n_leaves, float voxel_size, unsigned flags, int* err) {
	if (!leaf_origins_xyz && n_leaves) {
		if (err) *err = fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_create_from_leaves: null origin array");
		return nullptr;
	}
	return grid_from_origins(leaf_origins_xyz, n_leaves, voxel_size, flags, err);
}

// Leaf-density check of all N coordinates on the host, where they already are: shipping them to the device would cost
// more than reading them once. Threads take contiguous leaf ranges; the lowest offending coordinate is reported, as a
// serial scan would. Writes the leaf origins (every 512th coordinate) to `origins` (n_leaves x 3).
static int scan_coords(const char* who, const int32_t* coords, uint64_t n_voxels, unsigned flags, std::vector<int32_t>& origins) {
	if (!coords && n_voxels) {
		set_error("%s: null coordinate array", who);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	if (n_voxels % 512u) {
		set_error("%s: %llu coordinates is not a multiple of 512: the domain must be leaf-dense", who, (unsigned long long)n_voxels);
		return HNS_ERR_TOPOLOGY;
	}
	const uint64_t n_leaves = n_voxels / 512u;
	origins.resize((size_t)n_leaves * 3);
	const bool validate = !(flags & HNS_GRID_SKIP_VALIDATE);
	unsigned n_threads = 1;
	if (validate && n_leaves >= 2048) {
		n_threads = std::thread::hardware_concurrency();
		if (n_threads == 0) n_threads = 1;
		if (n_threads > 16) n_threads = 16;
	}
	std::vector<uint64_t> first_bad(n_threads, UINT64_MAX);
	auto scan = [&](unsigned t) {
		const uint64_t l0 = n_leaves * t / n_threads, l1 = n_leaves * (t + 1) / n_threads;
		for (uint64_t l = l0; l < l1; ++l) {
			const int32_t* c = coords + 3 * 512 * l;
			origins[3 * l] = c[0];
			origins[3 * l + 1] = c[1];
			origins[3 * l + 2] = c[2];
			if (!validate || first_bad[t] != UINT64_MAX) continue;
			for (int n = 1; n < 512; ++n) {
				const int32_t* v = c + 3 * n;
				if (v[0] != c[0] + (n >> 6) || v[1] != c[1] + ((n >> 3) & 7) || v[2] != c[2] + (n & 7)) {
					first_bad[t] = 512 * l + n;
					break;
				}
			}
		}
	};
	if (n_threads == 1) {
		scan(0);
	} else {
		std::vector<std::thread> pool;
		for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(scan, t);
		scan(0);
		for (auto& th : pool) th.join();
	}
	for (unsigned t = 0; t < n_threads; ++t) {
		if (first_bad[t] == UINT64_MAX) continue;
		const uint64_t i = first_bad[t];
		const int32_t* v = coords + 3 * i;
		set_error("%s: coordinate %llu = (%d,%d,%d) breaks the leaf-dense x<<6|y<<3|z order of leaf %llu", who, (unsigned long long)i, v[0], v[1], v[2],
		          (unsigned long long)(i / 512));
		return HNS_ERR_TOPOLOGY;
	}
	return HNS_OK;
}

hns_grid* hns_grid_create(const int32_t* coords, uint64_t n_voxels, float voxel_size, unsigned flags, int* err) {
	std::vector<int32_t> origins;
	const int rc = scan_coords("hns_grid_create", coords, n_voxels, flags, origins);
	if (rc != HNS_OK) {
		if (err) *err = rc;
		return nullptr;
	}
	return grid_from_origins(origins.data(), n_voxels / 512u, voxel_size, flags, err);
}

// 1 when `coords` describes exactly the leaves of `g` in the same order (so every table, and any device state kept with
// the grid, is still valid), 0 when it does not, < 0 when the coordinates are not leaf-dense. The caller decides
// whether to keep the grid or build a new one: this is the "topology unchanged" test of a cook.
int hns_grid_matches(const hns_grid* g, const int32_t* coords, uint64_t n_voxels, unsigned flags) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_matches: null grid");
	std::vector<int32_t> origins;
	const int rc = scan_coords("hns_grid_matches", coords, n_voxels, flags, origins);
	if (rc != HNS_OK) return rc;
	const uint64_t n_leaves = n_voxels / 512u;
	if (n_leaves != (uint64_t)g->topo.n_leaves) return 0;
	for (uint64_t l = 0; l < n_leaves; ++l) {
		const int32_t* o = &g->topo.origins[4 * (size_t)l];
		if (o[0] != origins[3 * l] || o[1] != origins[3 * l + 1] || o[2] != origins[3 * l + 2]) return 0;
	}
	return 1;
}

void hns_grid_destroy(hns_grid* g) {
	if (!g) return;
	hns_grid_free_device(g);
	delete g;
}

uint64_t hns_grid_leaf_count(const hns_grid* g) { return g ? (uint64_t)g->topo.n_leaves : 0; }
uint64_t hns_grid_voxel_count(const hns_grid* g) { return g ? (uint64_t)g->topo.n_leaves * 512u : 0; }
float hns_grid_voxel_size(const hns_grid* g) { return g ? g->voxel_size : 0.0f; }
uint64_t hns_grid_active_leaves(const hns_grid* g) { return g ? g->n_active : 0; }

int hns_grid_set_active_range(hns_grid* g, uint64_t first, uint64_t count) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_active_range: null grid");
	if (first > (uint64_t)g->topo.n_leaves || count > (uint64_t)g->topo.n_leaves - first)
		return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_active_range: range beyond the grid's leaves");
	g->first_active = count ? first : 0;
	g->n_active = count;
	return g->on_device ? hns_grid_upload_schedule(g) : HNS_OK;
}

int hns_grid_set_active_leaves(hns_grid* g, uint64_t n_active) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_active_leaves: null grid");
	if (n_active > (uint64_t)g->topo.n_leaves) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_active_leaves: more active leaves than leaves");
	return hns_grid_set_active_range(g, 0, n_active);
}

int hns_grid_set_outside_element(hns_grid* g, uint64_t element_index) {
	if (!g) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_outside_element: null grid");
	if (element_index >= (uint64_t)g->topo.n_leaves * 512u && element_index != 0)
		return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_set_outside_element: index beyond the grid");
	g->outside_element = element_index;
	return HNS_OK;
}

int hns_grid_offsets(const hns_grid* g, const int32_t* ijk, uint64_t n, uint64_t* out) {
	if (!g || (!ijk && n) || (!out && n)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_offsets: null argument");
	if (int rc = hns_grid_host_tables(g)) return rc;
	for (uint64_t t = 0; t < n; ++t) out[t] = g->topo.offset(ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]);
	return HNS_OK;
}

int hns_grid_neighbor_table(const hns_grid* g, int32_t* out) {
	if (!g || !out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_neighbor_table: null argument");
	if (int rc = hns_grid_host_tables(g)) return rc;
	memcpy(out, g->topo.nbr27.data(), g->topo.nbr27.size() * sizeof(int32_t));
	return HNS_OK;
}

int hns_grid_coords(const hns_grid* g, int32_t* out) {
	if (!g || !out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_coords: null argument");
	for (int64_t l = 0; l < g->topo.n_leaves; ++l) {
		const int32_t* o = &g->topo.origins[4 * (size_t)l];
		for (int n = 0; n < 512; ++n) {
			int32_t* c = out + 3 * (l * 512 + n);
			c[0] = o[0] + (n >> 6);
			c[1] = o[1] + ((n >> 3) & 7);
			c[2] = o[2] + (n & 7);
		}
	}
	return HNS_OK;
}

}  // extern "C"
