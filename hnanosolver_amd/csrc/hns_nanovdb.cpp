// hns_nanovdb.cpp -- serialises an hns_grid as a NanoVDB NanoGrid<ValueOnIndex> buffer (NanoVDB 32.7.0 layout).
//
// The reference keeps its index grid in exactly this format (create_index_grid, reference src/Cuda/HNanoSolver.cu:375-384
// -> externals/nanovdb/tools/cuda/PointsToGrid.cuh). Nothing in libhns reads it -- the kernels use the flat tables of
// hns_topology.cpp -- but other NanoVDB consumers and .nvdb writers do, so this is the interop format of SURVEY.md 8f-4.
// Field by field the buffer follows what voxelsToGrid<ValueOnIndex> writes for a leaf-dense voxel set
// (PointsToGrid.cuh:774-975 grid/tree/root/internal nodes, :981-1064 + :447-473 leaves, :1139-1191 bounding boxes);
// offsets are those of NanoVDB.h:1810-1832 (GridData), :2254-2260 (TreeData), :2536-2565 (RootData + Tile),
// :3164 ff. (InternalData) and :4143-4156 (LeafIndexBase), verified against the real headers by
// tests/test_nanovdb_export.py through oracle/_ref.
//
// One deliberate generalisation: leaves are stored in NanoVDB's breadth-first order, but each leaf's mOffset is
// 1 + 512 * (position of the leaf in the CALLER's coordinate array). For callers that pass leaves in NanoVDB order
// (the reference's IndexGridBuilder does) that is the reference's value; for any other order the accessor still returns
// the caller's own flat index, which is what the solver fields are laid out by.
#include <algorithm>
#include <cstring>

#include "hns_internal.hpp"

namespace {

constexpr uint64_t kGridBytes = 672, kTreeBytes = 64, kRootBytes = 96, kTileBytes = 32, kUpperBytes = 270400, kLowerBytes = 33856, kLeafBytes = 96;
constexpr uint64_t kMagicNumb = 0x304244566f6e614eull;  // "NanoVDB0"
constexpr uint64_t kMagicGrid = 0x314244566f6e614eull;  // "NanoVDB1"
constexpr uint32_t kVersion = (32u << 21) | (7u << 10) | 0u;
constexpr uint32_t kFlagHasBBox = 2u, kFlagBreadthFirst = 32u;
constexpr uint32_t kGridTypeOnIndex = 20u;
constexpr uint32_t kGridClassUnknown = 0u;  // voxelsToGrid only sets IndexGrid for ValueIndex (PointsToGrid.cuh:892-895); ValueOnIndex stays Unknown

struct Writer {
	uint8_t* base;
	template <typename T>
	void put(uint64_t off, T v) const {
		memcpy(base + off, &v, sizeof(T));
	}
};

struct LeafKey {
	int64_t tile[3];  // origin >> 12, signed: the root-tile sort key
	uint32_t upper;   // child slot inside the 4096^3 upper node
	uint32_t lower;   // child slot inside the 128^3 lower node
	int32_t leaf;     // index in the caller's order
};

struct Box {
	int32_t lo[3], hi[3];
	void reset() {
		for (int a = 0; a < 3; ++a) {
			lo[a] = INT32_MAX;
			hi[a] = INT32_MIN;
		}
	}
	void add(const int32_t* mn, const int32_t* mx) {
		for (int a = 0; a < 3; ++a) {
			lo[a] = std::min(lo[a], mn[a]);
			hi[a] = std::max(hi[a], mx[a]);
		}
	}
	void write(const Writer& w, uint64_t off) const {
		for (int a = 0; a < 3; ++a) {
			w.put<int32_t>(off + 4 * a, lo[a]);
			w.put<int32_t>(off + 12 + 4 * a, hi[a]);
		}
	}
};

}  // namespace

extern "C" int hns_grid_export_nanovdb(const hns_grid* g, void* buffer, uint64_t capacity, uint64_t* size_out) {
	using namespace hns;
	if (!g || !size_out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_export_nanovdb: null argument");
	const int64_t n = g->topo.n_leaves;
	std::vector<LeafKey> keys((size_t)n);
	for (int64_t l = 0; l < n; ++l) {
		const int32_t* o = &g->topo.origins[4 * (size_t)l];
		LeafKey& k = keys[(size_t)l];
		for (int a = 0; a < 3; ++a) k.tile[a] = (int64_t)o[a] >> 12;
		k.upper = (uint32_t)((((o[0] & 4095) >> 7) << 10) | (((o[1] & 4095) >> 7) << 5) | ((o[2] & 4095) >> 7));
		k.lower = (uint32_t)((((o[0] & 127) >> 3) << 8) | (((o[1] & 127) >> 3) << 4) | ((o[2] & 127) >> 3));
		k.leaf = (int32_t)l;
	}
	// breadth-first order: root tiles in signed (x, y, z) order, then child slot at each level (PointsToGrid.cuh:596-602,640-645)
	std::sort(keys.begin(), keys.end(), [](const LeafKey& a, const LeafKey& b) {
		for (int c = 0; c < 3; ++c)
			if (a.tile[c] != b.tile[c]) return a.tile[c] < b.tile[c];
		if (a.upper != b.upper) return a.upper < b.upper;
		return a.lower < b.lower;
	});
	uint64_t n_upper = 0, n_lower = 0;
	for (int64_t i = 0; i < n; ++i) {
		const bool new_tile = i == 0 || memcmp(keys[i].tile, keys[i - 1].tile, sizeof(keys[i].tile)) != 0;
		if (new_tile) ++n_upper;
		if (new_tile || keys[i].upper != keys[i - 1].upper) ++n_lower;
	}
	const uint64_t off_tree = kGridBytes, off_root = off_tree + kTreeBytes, off_upper = off_root + kRootBytes + kTileBytes * n_upper,
	               off_lower = off_upper + kUpperBytes * n_upper, off_leaf = off_lower + kLowerBytes * n_lower, total = off_leaf + kLeafBytes * (uint64_t)n;
	*size_out = total;
	if (!buffer) return HNS_OK;  // size query
	if (capacity < total) {
		set_error("hns_grid_export_nanovdb: buffer of %llu bytes is too small (%llu needed)", (unsigned long long)capacity, (unsigned long long)total);
		return HNS_ERR_INVALID_ARGUMENT;
	}
	if ((uintptr_t)buffer & 31u) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_grid_export_nanovdb: buffer must be 32-byte aligned (NANOVDB_DATA_ALIGNMENT)");
	memset(buffer, 0, (size_t)total);
	const Writer w{(uint8_t*)buffer};

	// ---- nodes: one pass over the sorted leaves ----
	Box root_box, upper_box, lower_box;
	root_box.reset();
	int64_t i_upper = -1, i_lower = -1;
	uint64_t upper_at = 0, lower_at = 0;
	auto close_lower = [&]() {
		if (i_lower >= 0) lower_box.write(w, lower_at);
	};
	auto close_upper = [&]() {
		if (i_upper >= 0) upper_box.write(w, upper_at);
	};
	for (int64_t i = 0; i < n; ++i) {
		const LeafKey& k = keys[(size_t)i];
		const int32_t* o = &g->topo.origins[4 * (size_t)k.leaf];
		const bool new_tile = i == 0 || memcmp(k.tile, keys[i - 1].tile, sizeof(k.tile)) != 0;
		const bool new_lower = new_tile || k.upper != keys[i - 1].upper;
		if (new_lower) close_lower();
		if (new_tile) {
			close_upper();
			++i_upper;
			upper_at = off_upper + kUpperBytes * (uint64_t)i_upper;
			upper_box.reset();
			// root tile -> this upper node (RootData::Tile::setChild): key = x<<42 | y<<21 | z of uint32(coord) >> 12
			const uint64_t t = off_root + kRootBytes + kTileBytes * (uint64_t)i_upper;
			const uint64_t key = ((uint64_t)((uint32_t)o[2] >> 12)) | ((uint64_t)((uint32_t)o[1] >> 12) << 21) | ((uint64_t)((uint32_t)o[0] >> 12) << 42);
			w.put<uint64_t>(t, key);
			w.put<int64_t>(t + 8, (int64_t)(upper_at - off_root));  // child: byte offset from the root
			// state (t+16) = 0, value (t+24) = 0
		}
		if (new_lower) {
			++i_lower;
			lower_at = off_lower + kLowerBytes * (uint64_t)i_lower;
			lower_box.reset();
			// upper: child mask bit + table entry (byte offset from the upper node to the lower node)
			const uint64_t word = upper_at + 4128 + 8 * (k.upper >> 6);
			uint64_t m;
			memcpy(&m, w.base + word, 8);
			m |= 1ull << (k.upper & 63);
			w.put<uint64_t>(word, m);
			w.put<int64_t>(upper_at + 8256 + 8 * (uint64_t)k.upper, (int64_t)(lower_at - upper_at));
		}
		// lower: child mask bit + table entry
		const uint64_t leaf_at = off_leaf + kLeafBytes * (uint64_t)i;
		{
			const uint64_t word = lower_at + 544 + 8 * (k.lower >> 6);
			uint64_t m;
			memcpy(&m, w.base + word, 8);
			m |= 1ull << (k.lower & 63);
			w.put<uint64_t>(word, m);
			w.put<int64_t>(lower_at + 1088 + 8 * (uint64_t)k.lower, (int64_t)leaf_at - (int64_t)lower_at);
		}
		// leaf (LeafIndexBase): bbox min + extent, flags, full value mask, offset, packed prefix sums
		for (int a = 0; a < 3; ++a) {
			w.put<int32_t>(leaf_at + 4 * a, o[a]);
			w.put<uint8_t>(leaf_at + 12 + a, 7);
		}
		w.put<uint8_t>(leaf_at + 15, (uint8_t)((kFlagHasBBox | kFlagBreadthFirst) | 2u));  // grid flags copied into the leaf, then "has bbox" (updateBBox)
		memset(w.base + leaf_at + 16, 0xFF, 64);
		w.put<uint64_t>(leaf_at + 80, 1ull + 512ull * (uint64_t)k.leaf);
		uint64_t prefix = 0;
		for (int j = 0; j < 7; ++j) prefix |= (uint64_t)(64 * (j + 1)) << (9 * j);  // countOn of words 0..j, 9 bits each
		w.put<uint64_t>(leaf_at + 88, prefix);
		const int32_t hi[3] = {o[0] + 7, o[1] + 7, o[2] + 7};
		lower_box.add(o, hi);
		upper_box.add(o, hi);
		root_box.add(o, hi);
	}
	close_lower();
	close_upper();

	// ---- root ----
	if (n > 0) {
		root_box.write(w, off_root);
	} else {  // CoordBBox(): empty = [max, min]
		for (int a = 0; a < 3; ++a) {
			w.put<int32_t>(off_root + 4 * a, INT32_MAX);
			w.put<int32_t>(off_root + 12 + 4 * a, INT32_MIN);
		}
	}
	w.put<uint32_t>(off_root + 24, (uint32_t)n_upper);  // mTableSize; background/min/max/average/stddev stay 0

	// ---- tree ----
	w.put<int64_t>(off_tree + 0, (int64_t)(off_leaf - off_tree));
	w.put<int64_t>(off_tree + 8, (int64_t)(off_lower - off_tree));
	w.put<int64_t>(off_tree + 16, (int64_t)(off_upper - off_tree));
	w.put<int64_t>(off_tree + 24, (int64_t)(off_root - off_tree));
	const uint32_t counts[3] = {(uint32_t)n, (uint32_t)n_lower, (uint32_t)n_upper};
	for (int a = 0; a < 3; ++a) {
		w.put<uint32_t>(off_tree + 32 + 4 * a, counts[a]);  // mNodeCount
		w.put<uint32_t>(off_tree + 44 + 4 * a, counts[a]);  // mTileCount is set to the same numbers (PointsToGrid.cuh:793-795)
	}
	w.put<uint64_t>(off_tree + 56, 512ull * (uint64_t)n);  // mVoxelCount

	// ---- grid ----
	w.put<uint64_t>(0, kMagicNumb);
	w.put<uint64_t>(8, ~0ull);  // checksum disabled
	w.put<uint32_t>(16, kVersion);
	w.put<uint32_t>(20, kFlagHasBBox | kFlagBreadthFirst);
	w.put<uint32_t>(24, 0u);  // grid index
	w.put<uint32_t>(28, 1u);  // grid count
	w.put<uint64_t>(32, total);
	// name (40..296) stays empty. Map(voxelSize): uniform scale, no translation (NanoVDB.h:1372-1382)
	const double s = (double)g->voxel_size;
	const float sf = (float)s, isf = 1.0f / (float)s;
	for (int a = 0; a < 3; ++a) {
		w.put<float>(296 + 4 * (4 * a), sf);
		w.put<float>(332 + 4 * (4 * a), isf);
		w.put<double>(384 + 8 * (4 * a), s);
		w.put<double>(456 + 8 * (4 * a), 1.0 / s);
	}
	w.put<float>(380, 1.0f);   // mTaperF
	w.put<double>(552, 1.0);   // mTaperD
	for (int a = 0; a < 3; ++a) {  // world bbox = index bbox corners through the map (math/Math.h:1271-1284; max is NOT +1)
		if (n > 0) {
			w.put<double>(560 + 8 * a, s * (double)root_box.lo[a]);
			w.put<double>(584 + 8 * a, s * (double)root_box.hi[a]);
		} else {
			w.put<double>(560 + 8 * a, s * (double)INT32_MAX);
			w.put<double>(584 + 8 * a, s * (double)INT32_MIN);
		}
		w.put<double>(608 + 8 * a, s);  // voxel size
	}
	w.put<uint32_t>(632, kGridClassUnknown);
	w.put<uint32_t>(636, kGridTypeOnIndex);
	w.put<int64_t>(640, (int64_t)total);  // no blind data: offset = end of the leaves
	w.put<uint32_t>(648, 0u);
	w.put<uint32_t>(652, 0u);
	w.put<uint64_t>(656, 1ull + 512ull * (uint64_t)n);  // mData1 = value count incl. background slot 0
	w.put<uint64_t>(664, kMagicGrid);
	return HNS_OK;
}
