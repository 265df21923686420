// hns_shim.cpp -- REFERENCE-SIDE translation unit: replaces the reference's static `Kernels` library (src/Cuda/*.cu,
// src/Cuda/CMakeLists.txt:6-14; linked into every SOP at src/SOP/CMakeLists.txt:65-68) with libhns.so.
//
// COMPILE-BLOCKED IN THIS IMAGE: needs the reference checkout (src/Utils/GridData.hpp), OpenVDB (<openvdb/Types.h>) and,
// for the SOPs that call it, the Houdini HDK. It is committed as the exact source a maintainer adds to the reference tree
// (drop it next to src/SOP/, add this repo's include/ and integration/ to the include path, link libhns.so + amdhip64).
// The bodies live in hns_shim.hpp, where the CPU test-suite compiles them against this repo's container twin.
//
// Reference-side edits that go with it:
//   src/SOP/HNanoSolver/SOP_HNanoSolver.hpp:82-85   nanovdb::GridHandle<nanovdb::cuda::DeviceBuffer> -> hns_shim::GridHandle,
//                                                    cudaStream_t -> hipStream_t
//   src/SOP/HNanoSolver/SOP_HNanoSolver.cpp:226-239 `hns_shim::GridHandle handle;` (or keep it in SOP_HNanoSolverCache, :60-64,
//                                                    so that cooks on an unchanged topology reuse grid and device buffers);
//                                                    hipStreamCreate / hipStreamSynchronize / hipStreamDestroy
//   src/SOP/Advection/SOP_VDBAdvect.hpp:66, src/SOP/VelocityAdvection/SOP_VDBAdvectVelocity.hpp:60,
//   src/SOP/ProjectNonDivergent/SOP_VDBProjectNonDivergent.hpp:69-70        cudaStream_t -> hipStream_t
//   src/Utils/Memory.hpp (cudaMallocHost / cudaFreeHost of AllocationType::CudaPinned)  -> hipHostMalloc / hipHostFree
//   src/SOP/CMakeLists.txt:65-68                     link hns + amdhip64 instead of Kernels; drop src/Cuda and nanovdb/cuda
#include <hip/hip_runtime_api.h>
#include <openvdb/Types.h>

#include "../src/Cuda/Kernels.cuh"      // CombustionParams (Kernels.cuh:6-13): six floats, field for field hns_combustion_params
#include "../src/Utils/GridData.hpp"    // HNS::GridIndexedData, the reference's own container, unchanged
#include "hns_shim.hpp"

using Vec3 = openvdb::Vec3f;

extern "C" void CreateIndexGrid(HNS::GridIndexedData& data, hns_shim::GridHandle& handle, const float voxelSize) {
	hns_shim::create_index_grid(data, handle, voxelSize);
}

extern "C" void Compute_Sim(HNS::GridIndexedData& data, const hns_shim::GridHandle& handle, const int iteration, const float dt, const float voxelSize,
                            const CombustionParams& params, const bool hasCollision, const hipStream_t& stream) {
	hns_shim::compute_sim<Vec3>(data, handle, iteration, dt, voxelSize, params, hasCollision, stream);
}

extern "C" void AdvectIndexGrid(HNS::GridIndexedData& data, const float dt, const float voxelSize, const hipStream_t& stream) {
	hns_shim::advect_index_grid<Vec3>(data, dt, voxelSize, stream);
}

extern "C" void AdvectIndexGridVelocity(HNS::GridIndexedData& data, const float dt, const float voxelSize, const hipStream_t& stream) {
	hns_shim::advect_index_grid_velocity<Vec3>(data, dt, voxelSize, stream);
}

extern "C" void ProjectNonDivergent(HNS::GridIndexedData& data, const size_t iteration, const float voxelSize, const hipStream_t& stream) {
	hns_shim::project_non_divergent<Vec3>(data, iteration, voxelSize, stream);
}

extern "C" void Divergence(HNS::GridIndexedData& data, const float voxelSize, const hipStream_t& stream) {
	hns_shim::divergence<Vec3>(data, voxelSize, stream);
}
