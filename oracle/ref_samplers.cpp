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
// Built on the host with NanoVDB's own builder (tools/GridBuilder.h + tools/CreateNanoGrid.h), the same call the
// reference's tests use (Tests/IndexGrid.cpp:125).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>

#include "Utils/Stencils.hpp"
#include "nanovdb/tools/CreateNanoGrid.h"
#include "nanovdb/tools/GridBuilder.h"

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

}  // extern "C"
