// hns_leafio.cpp -- the steps either side of the path, without OpenVDB: what the reference's
// HNS::IndexGridBuilder (src/Utils/GridBuilder.hpp:87-216) and the domain dilation of SOP_HNanoSolverVerb::cook
// (src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:186-199) do to OpenVDB trees, restated over raw 8^3 leaf buffers: a leaf is its
// 8-aligned origin, an optional 512-bit active mask (byte x*8+y, bit z) and 512 values in x<<6|y<<3|z order -- exactly an
// OpenVDB LeafNode's origin, value mask and buffer. Host code (the reference does this on the host with TBB).
//
// PARITY UNPINNED: OpenVDB is absent from this image, so nothing here has been checked against the reference's output; the
// tests check it against brute force. Quirks kept on purpose: leaves missing from an SDF source are
// filled with BYTES 0x01 (memset(..., 1, ...), GridBuilder.hpp:108: 0x01010101 = 2.4e-38f, not 1.0f), from any other source
// with zeros (:125-129,147-151); tile values of the sources are ignored (only leaves are probed, :105,122,144); output grids
// receive all 512 values of every domain leaf (:198-211).
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

#include "hns_internal.hpp"

using namespace hns;

namespace {

struct Key {
	int32_t x, y, z;
	bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct KeyHash {
	size_t operator()(const Key& k) const { return (size_t)hash_origin(k.x, k.y, k.z) * 0x9E3779B97F4A7C15ull ^ (size_t)(uint32_t)k.z; }
};

// OpenVDB's leaf order (LeafManager over root table -> 32^3 -> 16^3 children, x major) = NanoVDB's (tests/fields.nanovdb_order):
// signed root-tile coordinate (coord >> 12) x, y, z; then child offset in the 4096^3 node; then in the 128^3 node.
bool leaf_less(const Key& a, const Key& b) {
	auto parts = [](const Key& k, int64_t (&p)[5]) {
		p[0] = k.x >> 12, p[1] = k.y >> 12, p[2] = k.z >> 12;
		p[3] = ((int64_t)((k.x & 4095) >> 7) << 10) | ((int64_t)((k.y & 4095) >> 7) << 5) | ((k.z & 4095) >> 7);
		p[4] = ((int64_t)((k.x & 127) >> 3) << 8) | ((int64_t)((k.y & 127) >> 3) << 4) | ((k.z & 127) >> 3);
	};
	int64_t pa[5], pb[5];
	parts(a, pa);
	parts(b, pb);
	return std::lexicographical_compare(pa, pa + 5, pb, pb + 5);
}

int emit_sorted(std::vector<Key>& keys, int32_t* out, uint64_t capacity, uint64_t* n_out, const char* who) {
	std::sort(keys.begin(), keys.end(), leaf_less);
	if (n_out) *n_out = keys.size();
	if (!out) return HNS_OK;
	if (keys.size() > capacity) {
		set_error("%s: %zu leaves do not fit the output capacity %llu", who, keys.size(), (unsigned long long)capacity);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	for (size_t i = 0; i < keys.size(); ++i) out[3 * i] = keys[i].x, out[3 * i + 1] = keys[i].y, out[3 * i + 2] = keys[i].z;
	return HNS_OK;
}

int check_aligned(const int32_t* o, uint64_t n, const char* who) {
	for (uint64_t i = 0; i < 3 * n; ++i)
		if (o[i] & 7) {
			set_error("%s: leaf origin %llu is not 8-aligned", who, (unsigned long long)(i / 3));
			return HNS_ERR_TOPOLOGY;
		}
	return HNS_OK;
}

}  // namespace

extern "C" {

// IndexGridBuilder::build (GridBuilder.hpp:99-154): for every leaf of the domain, the source leaf with the same origin is
// copied whole; a domain leaf the source lacks is filled (zeros, or bytes 0x01 for an SDF source).
int hns_gather_leaves(const int32_t* domain_origins, uint64_t n_domain, const int32_t* src_origins, uint64_t n_src, const float* src_values, int ncomp, int fill,
                      float* out) {
	if ((n_domain && (!domain_origins || !out)) || (n_src && (!src_origins || !src_values)) || (ncomp != 1 && ncomp != 3) ||
	    (fill != HNS_FILL_ZERO && fill != HNS_FILL_SDF))
		return fail(HNS_ERR_INVALID_ARGUMENT, "hns_gather_leaves: bad arguments");
	std::unordered_map<Key, uint64_t, KeyHash> where;
	where.reserve((size_t)n_src * 2);
	for (uint64_t i = 0; i < n_src; ++i) where[Key{src_origins[3 * i], src_origins[3 * i + 1], src_origins[3 * i + 2]}] = i;
	const size_t leaf_floats = 512u * (size_t)ncomp;
	for (uint64_t i = 0; i < n_domain; ++i) {
		float* dst = out + i * leaf_floats;
		const auto it = where.find(Key{domain_origins[3 * i], domain_origins[3 * i + 1], domain_origins[3 * i + 2]});
		if (it != where.end())
			memcpy(dst, src_values + it->second * leaf_floats, leaf_floats * sizeof(float));
		else
			memset(dst, fill == HNS_FILL_SDF ? 1 : 0, leaf_floats * sizeof(float));
	}
	return HNS_OK;
}

// IndexGridBuilder::writeIndexGrid (GridBuilder.hpp:171-213): every domain leaf receives its 512 values.
int hns_scatter_leaves(const float* flat, uint64_t n_domain, int ncomp, float* const* leaf_buffers) {
	if ((n_domain && (!flat || !leaf_buffers)) || (ncomp != 1 && ncomp != 3)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_scatter_leaves: bad arguments");
	const size_t leaf_floats = 512u * (size_t)ncomp;
	for (uint64_t i = 0; i < n_domain; ++i) {
		if (!leaf_buffers[i]) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_scatter_leaves: null leaf buffer");
		memcpy(leaf_buffers[i], flat + i * leaf_floats, leaf_floats * sizeof(float));
	}
	return HNS_OK;
}

// Leaves of the domain after dilateVoxels(padding, NN_FACE_EDGE_VERTEX, IGNORE_TILES) (SOP_HNanoSolver.cpp:190-193): a leaf is
// in the result iff some ACTIVE voxel lies within `padding` voxels (Chebyshev distance: faces, edges and vertices) of its
// box. The index grid then takes every voxel of those leaves (the domain is leaf-dense, GridBuilder.hpp:156-166,229).
// active_masks: n x 64 bytes, NULL = every voxel active. Output in OpenVDB leaf order; out may be NULL to query n_out.
int hns_dilate_leaves(const int32_t* origins, uint64_t n, const unsigned char* active_masks, int padding_voxels, int32_t* out_origins, uint64_t capacity,
                      uint64_t* n_out) {
	if ((n && !origins) || padding_voxels < 0 || padding_voxels > 1024) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dilate_leaves: bad arguments");
	if (int rc = check_aligned(origins, n, "hns_dilate_leaves")) return rc;
	const int p = padding_voxels, D = (p + 7) / 8;
	std::unordered_set<Key, KeyHash> have;
	have.reserve((size_t)n * 4);
	for (uint64_t i = 0; i < n; ++i) {
		const Key o{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]};
		const unsigned char* m = active_masks ? active_masks + 64 * i : nullptr;
		bool any = !m;
		for (int b = 0; m && b < 64 && !any; ++b) any = m[b] != 0;
		if (!any) continue;  // a leaf without active voxels contributes nothing (it would not be a leaf of the mask tree)
		for (int dx = -D; dx <= D; ++dx)
			for (int dy = -D; dy <= D; ++dy)
				for (int dz = -D; dz <= D; ++dz) {
					const int d[3] = {dx, dy, dz};
					int lo[3], hi[3];
					bool possible = true;
					for (int a = 0; a < 3; ++a) {
						lo[a] = d[a] > 0 ? std::max(0, 8 * d[a] - p) : 0;
						hi[a] = d[a] < 0 ? std::min(7, 8 * d[a] + 7 + p) : 7;
						possible &= lo[a] <= hi[a];
					}
					if (!possible) continue;
					bool hit = !m;
					for (int x = lo[0]; m && x <= hi[0] && !hit; ++x)
						for (int y = lo[1]; y <= hi[1] && !hit; ++y) {
							const unsigned zmask = (0xFFu >> (7 - hi[2])) & (0xFFu << lo[2]);
							hit = (m[x * 8 + y] & zmask) != 0;
						}
					if (!hit) continue;
					const int64_t nx = (int64_t)o.x + 8 * dx, ny = (int64_t)o.y + 8 * dy, nz = (int64_t)o.z + 8 * dz;
					if (nx < INT32_MIN || nx > INT32_MAX - 7 || ny < INT32_MIN || ny > INT32_MAX - 7 || nz < INT32_MIN || nz > INT32_MAX - 7) continue;
					have.insert(Key{(int32_t)nx, (int32_t)ny, (int32_t)nz});
				}
	}
	std::vector<Key> keys(have.begin(), have.end());
	return emit_sorted(keys, out_origins, capacity, n_out, "hns_dilate_leaves");
}

// topologyUnion of two leaf sets (SOP_HNanoSolver.cpp:189,195-197), in OpenVDB leaf order, duplicates removed.
int hns_union_leaves(const int32_t* a, uint64_t na, const int32_t* b, uint64_t nb, int32_t* out_origins, uint64_t capacity, uint64_t* n_out) {
	if ((na && !a) || (nb && !b)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_union_leaves: bad arguments");
	if (int rc = check_aligned(a, na, "hns_union_leaves")) return rc;
	if (int rc = check_aligned(b, nb, "hns_union_leaves")) return rc;
	std::unordered_set<Key, KeyHash> have;
	have.reserve((size_t)(na + nb) * 2);
	for (uint64_t i = 0; i < na; ++i) have.insert(Key{a[3 * i], a[3 * i + 1], a[3 * i + 2]});
	for (uint64_t i = 0; i < nb; ++i) have.insert(Key{b[3 * i], b[3 * i + 1], b[3 * i + 2]});
	std::vector<Key> keys(have.begin(), have.end());
	return emit_sorted(keys, out_origins, capacity, n_out, "hns_union_leaves");
}

}  // extern "C"
