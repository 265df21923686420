// ref_samplers.cpp -- driver that exposes the REFERENCE's own topology + sampler code through a C ABI.
//
// TEST INFRASTRUCTURE ONLY. This file contains no reference code: it #includes the reference headers where they
// lie (/root/reference/src/Utils/Stencils.hpp and the vendored NanoVDB under /root/reference/externals) at build
// time; oracle/Makefile compiles it into oracle/_ref/libhns_ref.so (git-ignored). The only accommodation for
// building without the CUDA toolkit is two command-line macros, -D__forceinline__=inline and -D__fmaf_rn=fmaf
// (a compiler hint and the IEEE fused multiply-add that __fmaf_rn is defined to be); no stand-in headers.
//
// What it pins for oracle/hns_oracle.c:
//   IndexOffsetSampler<0>::offset        (Stencils.hpp:59-61)   -> ref_offsets
//   IndexSampler<float,0>                (Stencils.hpp:74-93)   -> ref_sample_nearest_f
//   IndexSampler<float,1>, <Vec3f,1>     (Stencils.hpp:96-173)  -> ref_sample_trilinear_f / _v  (HOST lerp branch, :137)
//   leaf order + 1-based dense offsets of NanoGrid<ValueOnIndex> (NanoVDB.h:4219-4228) -> ref_leaf_origins
//   a foreign buffer judged by NanoVDB itself (validator, accessor, tree iterators)     -> ref_nanovdb_* (pins
//   hns_grid_export_nanovdb)
// Built on the host with NanoVDB's own builder (tools/GridBuilder.h + tools/CreateNanoGrid.h), the same call the
// reference's tests use (Tests/IndexGrid.cpp:125).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>

#include "Utils/Stencils.hpp"
#include "nanovdb/tools/CreateNanoGrid.h"
#include "nanovdb/tools/GridBuilder.h"
#include "nanovdb/tools/GridValidator.h"

namespace {
struct RefGrid {
	nanovdb::GridHandle<nanovdb::HostBuffer> handle;
	const nanovdb::NanoGrid<nanovdb::ValueOnIndex>* grid = nullptr;
};
}  // namespace

extern "C" {

// Every voxel of every listed leaf is active (the reference's domain is leaf-dense, GridBuilder.hpp:156-166,229).
void* ref_grid_create(const int32_t* leaf_origins, int64_t n_leaves) {
	nanovdb::tools::build::Grid<float> g(0.0f);
	auto acc = g.getAccessor();
	for (int64_t l = 0; l < n_leaves; ++l) {
		const nanovdb::Coord o(leaf_origins[3 * l], leaf_origins[3 * l + 1], leaf_origins[3 * l + 2]);
		for (int n = 0; n < 512; ++n) acc.setValue(o + nanovdb::Coord(n >> 6, (n >> 3) & 7, n & 7), 1.0f);
	}
	auto* r = new RefGrid;
	r->handle = nanovdb::tools::createNanoGrid<nanovdb::tools::build::Grid<float>, nanovdb::ValueOnIndex>(g, 1u, false, false);
	r->grid = r->handle.grid<nanovdb::ValueOnIndex>();
	if (!r->grid) {
		delete r;
		return nullptr;
	}
	return r;
}

// Arbitrary active voxels (for the TestNanoVDB.cu:311-355 known answer).
void* ref_grid_create_from_voxels(const int32_t* ijk, int64_t n) {
	nanovdb::tools::build::Grid<float> g(0.0f);
	auto acc = g.getAccessor();
	for (int64_t t = 0; t < n; ++t) acc.setValue(nanovdb::Coord(ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]), 1.0f);
	auto* r = new RefGrid;
	r->handle = nanovdb::tools::createNanoGrid<nanovdb::tools::build::Grid<float>, nanovdb::ValueOnIndex>(g, 1u, false, false);
	r->grid = r->handle.grid<nanovdb::ValueOnIndex>();
	if (!r->grid) {
		delete r;
		return nullptr;
	}
	return r;
}

void ref_grid_destroy(void* h) { delete static_cast<RefGrid*>(h); }

int64_t ref_leaf_count(void* h) { return static_cast<RefGrid*>(h)->grid->tree().nodeCount(0); }
uint64_t ref_value_count(void* h) { return static_cast<RefGrid*>(h)->grid->valueCount(); }
uint64_t ref_active_voxel_count(void* h) { return static_cast<RefGrid*>(h)->grid->activeVoxelCount(); }

void ref_leaf_origins(void* h, int32_t* out) {
	const auto* grid = static_cast<RefGrid*>(h)->grid;
	const auto* leaf = grid->tree().getFirstNode<0>();
	const uint32_t n = grid->tree().nodeCount(0);
	for (uint32_t l = 0; l < n; ++l) {
		const nanovdb::Coord o = leaf[l].origin();
		out[3 * l] = o[0];
		out[3 * l + 1] = o[1];
		out[3 * l + 2] = o[2];
	}
}

// the NanoGrid<ValueOnIndex>* itself, for ref_kernels.cpp (the reference's kernels take it as `domainGrid`)
const void* ref_grid_nanogrid(void* h) { return static_cast<RefGrid*>(h)->grid; }

// d_coords as the reference builds it: the coordinate of every value, in value order (offset(coords[i]) == i + 1)
void ref_coords(void* h, int32_t* out) {
	const auto* grid = static_cast<RefGrid*>(h)->grid;
	const auto* leaf = grid->tree().getFirstNode<0>();
	const uint32_t n = grid->tree().nodeCount(0);
	for (uint32_t l = 0; l < n; ++l)
		for (int v = 0; v < 512; ++v) {
			const nanovdb::Coord c = leaf[l].offsetToGlobalCoord(v);
			for (int a = 0; a < 3; ++a) out[(size_t(l) * 512 + v) * 3 + a] = c[a];
		}
}

void ref_offsets(void* h, const int32_t* ijk, int64_t n, uint64_t* out) {
	const IndexOffsetSampler<0> s(static_cast<RefGrid*>(h)->grid);
	for (int64_t t = 0; t < n; ++t) out[t] = s.offset(ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]);
}

void ref_sample_nearest_f(void* h, const float* data, const int32_t* ijk, int64_t n, float* out) {
	const IndexOffsetSampler<0> s(static_cast<RefGrid*>(h)->grid);
	const IndexSampler<float, 0> f(s, data);
	for (int64_t t = 0; t < n; ++t) out[t] = f(ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]);
}

void ref_sample_trilinear_f(void* h, const float* data, const float* xyz, int64_t n, float* out) {
	const IndexOffsetSampler<0> s(static_cast<RefGrid*>(h)->grid);
	const IndexSampler<float, 1> f(s, data);
	for (int64_t t = 0; t < n; ++t) out[t] = f(nanovdb::Vec3f(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]));
}

void ref_sample_trilinear_v(void* h, const float* data3, const float* xyz, int64_t n, float* out3) {
	const IndexOffsetSampler<0> s(static_cast<RefGrid*>(h)->grid);
	const IndexSampler<nanovdb::Vec3f, 1> f(s, reinterpret_cast<const nanovdb::Vec3f*>(data3));
	for (int64_t t = 0; t < n; ++t) {
		const nanovdb::Vec3f r = f(nanovdb::Vec3f(xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]));
		out3[3 * t] = r[0];
		out3[3 * t + 1] = r[1];
		out3[3 * t + 2] = r[2];
	}
}

// ---- a buffer produced elsewhere (hns_grid_export_nanovdb), read through NanoVDB's own classes -------------------

using OnIndexGrid = nanovdb::NanoGrid<nanovdb::ValueOnIndex>;

// NanoVDB's full validator (tools/GridValidator.h:58-153): magic, version, type/class, alignment, node bounds,
// breadth-first order. Returns 0 when it has no complaint; otherwise copies the message.
int ref_nanovdb_check(const void* buf, char* err, int errlen) {
	char msg[256];
	nanovdb::tools::checkGrid(reinterpret_cast<const OnIndexGrid*>(buf), msg, nanovdb::CheckMode::Full);
	snprintf(err, (size_t)errlen, "%s", msg);
	return msg[0] ? 1 : 0;
}

// getValue / isActive through a ReadAccessor, the call IndexOffsetSampler<0> makes (Stencils.hpp:59-61)
void ref_nanovdb_query(const void* buf, const int32_t* ijk, int64_t n, uint64_t* values, uint8_t* active) {
	auto acc = reinterpret_cast<const OnIndexGrid*>(buf)->getAccessor();
	for (int64_t t = 0; t < n; ++t) {
		const nanovdb::Coord c(ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]);
		values[t] = acc.getValue(c);
		active[t] = acc.isActive(c) ? 1 : 0;
	}
}

// Header/tree facts as NanoVDB's accessors report them.
//   u[0..11]: valueCount, activeVoxelCount, gridSize, nodeCount(0..2), isBreadthFirst, gridType, gridClass, gridCount, blindDataCount, checksum-is-empty
//   i[0..5]:  index bbox min, max     d[0..8]: world bbox min, max, voxel size
void ref_nanovdb_info(const void* buf, uint64_t* u, int32_t* i, double* d) {
	const auto* g = reinterpret_cast<const OnIndexGrid*>(buf);
	u[0] = g->valueCount();
	u[1] = g->activeVoxelCount();
	u[2] = g->gridSize();
	u[3] = g->tree().nodeCount(0);
	u[4] = g->tree().nodeCount(1);
	u[5] = g->tree().nodeCount(2);
	u[6] = g->isBreadthFirst() ? 1 : 0;
	u[7] = (uint64_t)g->gridType();
	u[8] = (uint64_t)g->gridClass();
	u[9] = g->gridCount();
	u[10] = g->blindDataCount();
	u[11] = g->checksum().isEmpty() ? 1 : 0;
	const auto bb = g->indexBBox();
	const auto wb = g->worldBBox();
	const auto vs = g->voxelSize();
	for (int a = 0; a < 3; ++a) {
		i[a] = bb.min()[a];
		i[3 + a] = bb.max()[a];
		d[a] = wb.min()[a];
		d[3 + a] = wb.max()[a];
		d[6 + a] = vs[a];
	}
}

// Walk root -> upper -> lower -> leaf with NanoVDB's child iterators; per leaf: origin, first value index, bbox extent
// and flags. Returns the number of leaves visited (out arrays sized by ref_nanovdb_info's nodeCount(0)).
int64_t ref_nanovdb_leaves(const void* buf, int32_t* origins, uint64_t* first_value, int32_t* bbox_minmax, uint8_t* flags) {
	const auto* g = reinterpret_cast<const OnIndexGrid*>(buf);
	int64_t k = 0;
	for (auto it2 = g->tree().root().cbeginChild(); it2; ++it2)
		for (auto it1 = it2->cbeginChild(); it1; ++it1)
			for (auto it0 = it1->cbeginChild(); it0; ++it0) {
				const auto& leaf = *it0;
				const auto bb = leaf.bbox();
				for (int a = 0; a < 3; ++a) {
					origins[3 * k + a] = leaf.origin()[a];
					bbox_minmax[6 * k + a] = bb.min()[a];
					bbox_minmax[6 * k + 3 + a] = bb.max()[a];
				}
				first_value[k] = leaf.getValue(0);
				flags[k] = leaf.data()->mFlags;
				++k;
			}
	return k;
}

// The same leaf set through NanoVDB's HOST builder, raw bytes out: for comparing node payloads byte for byte.
// (The CUDA builder the reference calls cannot run here; the two differ only in header fields, see the test.)
uint64_t ref_nanovdb_host_build(const int32_t* leaf_origins, int64_t n_leaves, double voxel_size, void* out, uint64_t capacity) {
	nanovdb::tools::build::Grid<float> g(0.0f);
	g.setTransform(voxel_size);
	auto acc = g.getAccessor();
	for (int64_t l = 0; l < n_leaves; ++l) {
		const nanovdb::Coord o(leaf_origins[3 * l], leaf_origins[3 * l + 1], leaf_origins[3 * l + 2]);
		for (int n = 0; n < 512; ++n) acc.setValue(o + nanovdb::Coord(n >> 6, (n >> 3) & 7, n & 7), 1.0f);
	}
	auto h = nanovdb::tools::createNanoGrid<nanovdb::tools::build::Grid<float>, nanovdb::ValueOnIndex>(g, 0u, false, false);  // no value channel
	const uint64_t size = h.size();
	if (out && capacity >= size) memcpy(out, h.data(), size);
	return size;
}

}  // extern "C"
