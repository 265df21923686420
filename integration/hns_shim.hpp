// hns_shim.hpp -- the reference-side binding of libhns.so, written against the INTERFACE of the reference's container.
//
// The reference's SOPs call seven `extern "C"` functions with C++ types in their signatures (declared at
// src/SOP/HNanoSolver/SOP_HNanoSolver.hpp:82-85, src/SOP/Advection/SOP_VDBAdvect.hpp:66,
// src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.hpp:60, src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp:69-70;
// defined in src/Cuda/HNanoSolver.cu:375-396, Advection.cu:169-175, PressureProjection.cu:127-135). This header implements
// their bodies on top of the C ABI (include/hns.h) as templates over the container and vector types, using nothing but the
// container interface of src/Utils/GridData.hpp:16-166 (size, pCoords, getBlocksOfType<T>, pValues<T>):
//
//   * integration/hns_shim.cpp instantiates them for the reference's own HNS::GridIndexedData / openvdb::Vec3f and exports
//     the reference's symbol names -- that file needs the reference checkout, OpenVDB and the HDK and is compile-blocked in
//     this image;
//   * tests/cpp/shim_check.cpp instantiates the same templates for this repo's container twin
//     (hnanosolver_amd/host/HNanoSolver.hpp), which has the same interface: compiled by the CPU suite, run by the GPU suite.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "hns.h"

namespace hns_shim {

// Stands where nanovdb::GridHandle<nanovdb::cuda::DeviceBuffer> stood (SOP_HNanoSolver.cpp:226-235): owns the hns_grid.
struct GridHandle {
	hns_grid* g = nullptr;
	GridHandle() = default;
	GridHandle(const GridHandle&) = delete;
	GridHandle& operator=(const GridHandle&) = delete;
	~GridHandle() { reset(); }
	void reset() {
		if (g) hns_grid_destroy(g);
		g = nullptr;
	}
	bool isEmpty() const { return !g; }
};

// HNS_ERR_INVALID_ARGUMENT <-> std::invalid_argument (HNanoSolver.cu:12-23), everything else <-> std::runtime_error
// (HNanoSolver.cu:44,62,196; Advection.cu:20,28; CUDA_CHECK, Utils.cuh:10-18), with the library's message.
inline void check(int rc) {
	if (rc == HNS_ERR_INVALID_ARGUMENT) throw std::invalid_argument(hns_last_error());
	if (rc < 0) throw std::runtime_error(hns_last_error());
}

// Float blocks first, then the Vec3f block: the order Compute() itself walks them in (HNanoSolver.cu:36-63); names are kept
// alive in `keep`. Vec3T must be three packed floats (openvdb::Vec3f is), the coordinate type three packed int32.
template <class Vec3T, class DataT>
std::vector<hns_field> fields_of(DataT& d, std::vector<std::string>& keep) {
	static_assert(sizeof(Vec3T) == 3 * sizeof(float), "Vec3f blocks must be packed float[3]");
	std::vector<hns_field> f;
	const auto floats = d.template getBlocksOfType<float>();
	const auto vecs = d.template getBlocksOfType<Vec3T>();
	keep.reserve(keep.size() + floats.size() + vecs.size());
	for (const auto& n : floats) {
		keep.push_back(n);
		f.push_back(hns_field{keep.back().c_str(), 1, d.template pValues<float>(n)});
	}
	for (const auto& n : vecs) {
		keep.push_back(n);
		f.push_back(hns_field{keep.back().c_str(), 3, reinterpret_cast<float*>(d.template pValues<Vec3T>(n))});
	}
	return f;
}

// CreateIndexGrid (HNanoSolver.cu:375-390). A handle that already holds exactly these leaves is kept, with the device
// buffers the previous cook left with it (the reference's node cache is empty, SOP_HNanoSolver.hpp:60-64).
template <class DataT>
void create_index_grid(DataT& data, GridHandle& handle, float voxelSize) {
	static_assert(sizeof(*data.pCoords()) == 3 * sizeof(int32_t), "coordinates must be packed int32[3]");
	const int32_t* coords = reinterpret_cast<const int32_t*>(data.pCoords());
	if (!coords) throw std::runtime_error("Host coordinate data pointer is null.");
	if (handle.g && hns_grid_voxel_size(handle.g) == voxelSize && hns_grid_matches(handle.g, coords, data.size(), HNS_GRID_DEFAULT) == 1) return;
	handle.reset();
	int err = 0;
	handle.g = hns_grid_create(coords, data.size(), voxelSize, HNS_GRID_DEFAULT, &err);
	if (!handle.g) check(err < 0 ? err : HNS_ERR_RUNTIME);
}

template <class Vec3T, class DataT, class ParamsT>
void compute_sim(DataT& data, const GridHandle& handle, int iteration, float dt, float voxelSize, const ParamsT& params, bool hasCollision, void* stream) {
	static_assert(sizeof(ParamsT) == sizeof(hns_combustion_params), "CombustionParams is six floats (Kernels.cuh:6-13)");
	std::vector<std::string> keep;
	auto f = fields_of<Vec3T>(data, keep);
	check(hns_compute_sim(handle.g, f.data(), (int)f.size(), iteration, dt, voxelSize, reinterpret_cast<const hns_combustion_params*>(&params), hasCollision ? 1 : 0,
	                      stream));
}

// The single-purpose operators rebuild the index grid on every call, as the reference does (Advection.cu:71,141;
// PressureProjection.cu:28,97).
template <class Vec3T, class DataT>
void advect_index_grid(DataT& data, float dt, float voxelSize, void* stream) {
	GridHandle h;
	create_index_grid(data, h, voxelSize);
	std::vector<std::string> keep;
	auto f = fields_of<Vec3T>(data, keep);
	check(hns_advect_index_grid(h.g, f.data(), (int)f.size(), dt, voxelSize, stream));
}

template <class Vec3T, class DataT>
void advect_index_grid_velocity(DataT& data, float dt, float voxelSize, void* stream) {
	GridHandle h;
	create_index_grid(data, h, voxelSize);
	std::vector<std::string> keep;
	auto f = fields_of<Vec3T>(data, keep);
	check(hns_advect_index_grid_velocity(h.g, f.data(), (int)f.size(), dt, voxelSize, stream));
}

template <class Vec3T, class DataT>
void project_non_divergent(DataT& data, size_t iterations, float voxelSize, void* stream) {
	GridHandle h;
	create_index_grid(data, h, voxelSize);
	std::vector<std::string> keep;
	auto f = fields_of<Vec3T>(data, keep);
	check(hns_project_non_divergent(h.g, f.data(), (int)f.size(), iterations, voxelSize, stream));
}

template <class Vec3T, class DataT>
void divergence(DataT& data, float voxelSize, void* stream) {
	GridHandle h;
	create_index_grid(data, h, voxelSize);
	std::vector<std::string> keep;
	auto f = fields_of<Vec3T>(data, keep);
	check(hns_divergence(h.g, f.data(), (int)f.size(), voxelSize, stream));  // writes the float block named "divergence"
}

}  // namespace hns_shim
