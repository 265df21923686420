// HNanoSolver.hpp -- C++17 host-side mirror of the reference's operator interface for the substep hot path, header-only,
// on top of the C ABI of libhns.so (include/hns.h). A SOP that today includes the reference's
// src/SOP/HNanoSolver/SOP_HNanoSolver.hpp declarations and links its static `Kernels` library can include this header
// and link libhns.so instead; names, argument order, in-place/synchronous semantics and exception types are the same:
//
//   HNS::GridIndexedData      <- src/Utils/GridData.hpp:16-166   (named typed blocks + coords, insertion order kept)
//   CombustionParams          <- src/Cuda/Kernels.cuh:6-13 / src/SOP/HNanoSolver/SOP_HNanoSolver.hpp:21-28
//   CreateIndexGrid           <- src/Cuda/HNanoSolver.cu:387-390
//   Compute_Sim               <- src/Cuda/HNanoSolver.cu:393-396 (Compute, :9-372)
//   AdvectIndexGrid           <- src/Cuda/Advection.cu:169-171
//   AdvectIndexGridVelocity   <- src/Cuda/Advection.cu:173-175
//   ProjectNonDivergent       <- src/Cuda/PressureProjection.cu:132-135
//   Divergence                <- src/Cuda/PressureProjection.cu:127-129
//
// Differences a caller sees: coordinates are HNS::Coord (3 x int32, layout-compatible with openvdb::Coord) and vectors
// HNS::Vec3f (3 x float, layout-compatible with openvdb::Vec3f), so OpenVDB is not needed to use the solver; the grid
// handle is HNS::IndexGridHandle instead of nanovdb::GridHandle<nanovdb::cuda::DeviceBuffer>; the stream is a
// hipStream_t passed as void*. There is no CPU fallback: without a HIP device every operator throws std::runtime_error.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <unordered_map>
#include <vector>

#include "hns.h"

namespace HNS {

struct Coord {
	int32_t v[3];
	int32_t x() const { return v[0]; }
	int32_t y() const { return v[1]; }
	int32_t z() const { return v[2]; }
};
struct Vec3f {
	float v[3];
	float& operator[](int i) { return v[i]; }
	const float& operator[](int i) const { return v[i]; }
};
static_assert(sizeof(Coord) == 12 && sizeof(Vec3f) == 12, "layout must match openvdb::Coord / openvdb::Vec3f");

enum class AllocationType { Standard, Aligned, CudaPinned };  // kept for source compatibility; storage is std::vector here

/// Named typed value blocks over one coordinate array; getBlocksOfType() returns names in insertion order, which is what
/// fixes the order of the scalar tables inside Compute_Sim (reference GridData.hpp:136-145, HNanoSolver.cu:322-348).
class GridIndexedData {
   public:
	GridIndexedData() = default;
	GridIndexedData(const GridIndexedData&) = delete;
	GridIndexedData& operator=(const GridIndexedData&) = delete;
	GridIndexedData(GridIndexedData&&) = default;
	GridIndexedData& operator=(GridIndexedData&&) = default;

	bool allocateCoords(size_t numElements) {
		m_coords.assign(numElements, Coord{{0, 0, 0}});
		m_size = numElements;
		m_hasCoords = true;
		return true;
	}

	template <typename T>
	bool addValueBlock(const std::string& name, size_t numElements) {
		static_assert(std::is_same<T, float>::value || std::is_same<T, Vec3f>::value, "float or HNS::Vec3f blocks only");
		if (m_index.count(name)) return false;
		Block b;
		b.name = name;
		b.type = std::type_index(typeid(T));
		b.ncomp = std::is_same<T, float>::value ? 1 : 3;
		b.data.assign(numElements * b.ncomp, 0.0f);
		m_index[name] = m_blocks.size();
		m_blocks.push_back(std::move(b));
		return true;
	}

	Coord* pCoords() { return m_hasCoords ? m_coords.data() : nullptr; }
	const Coord* pCoords() const { return m_hasCoords ? m_coords.data() : nullptr; }

	template <typename T>
	T* pValues(const std::string& name) {
		auto it = m_index.find(name);
		if (it == m_index.end() || m_blocks[it->second].type != std::type_index(typeid(T))) return nullptr;
		return reinterpret_cast<T*>(m_blocks[it->second].data.data());
	}
	template <typename T>
	const T* pValues(const std::string& name) const {
		return const_cast<GridIndexedData*>(this)->pValues<T>(name);
	}

	size_t size() const { return m_size; }
	size_t numValueBlocks() const { return m_blocks.size(); }

	void clear() {
		clearValues();
		clearCoords();
		m_size = 0;
	}
	void clearValues() {
		m_blocks.clear();
		m_index.clear();
	}
	void clearCoords() {
		m_coords.clear();
		m_hasCoords = false;
	}
	void setAllocationType(AllocationType) {}

	template <typename T>
	std::vector<std::string> getBlocksOfType() const {
		std::vector<std::string> names;
		for (const Block& b : m_blocks)
			if (b.type == std::type_index(typeid(T))) names.push_back(b.name);
		return names;
	}

	/// hns_field[] over every block, in insertion order (what crosses the C ABI)
	std::vector<hns_field> fields() {
		std::vector<hns_field> f;
		for (Block& b : m_blocks) f.push_back(hns_field{b.name.c_str(), b.ncomp, b.data.data()});
		return f;
	}

   private:
	struct Block {
		std::string name;
		std::type_index type = std::type_index(typeid(void));
		int ncomp = 1;
		std::vector<float> data;
	};
	std::vector<Coord> m_coords;
	bool m_hasCoords = false;
	std::vector<Block> m_blocks;
	std::unordered_map<std::string, size_t> m_index;
	size_t m_size = 0;
};

/// Owns an hns_grid (the role of nanovdb::GridHandle<nanovdb::cuda::DeviceBuffer> in the reference's signatures).
class IndexGridHandle {
   public:
	IndexGridHandle() = default;
	~IndexGridHandle() { reset(); }
	IndexGridHandle(const IndexGridHandle&) = delete;
	IndexGridHandle& operator=(const IndexGridHandle&) = delete;
	IndexGridHandle(IndexGridHandle&& o) noexcept : m_grid(o.m_grid) { o.m_grid = nullptr; }
	IndexGridHandle& operator=(IndexGridHandle&& o) noexcept {
		if (this != &o) {
			reset();
			m_grid = o.m_grid;
			o.m_grid = nullptr;
		}
		return *this;
	}
	bool isEmpty() const { return m_grid == nullptr; }
	hns_grid* get() const { return m_grid; }
	void reset(hns_grid* g = nullptr) {
		if (m_grid) hns_grid_destroy(m_grid);
		m_grid = g;
	}

   private:
	hns_grid* m_grid = nullptr;
};

namespace detail {
inline void check(int rc) {
	if (rc >= 0) return;
	const std::string msg = hns_last_error();
	if (rc == HNS_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);  // reference HNanoSolver.cu:12-23
	throw std::runtime_error(msg);                                         // reference Utils.cuh:10-18, HNanoSolver.cu:44,62,196
}
// A handle that already holds the grid of exactly these leaves (same order, same voxel size) is kept, together with the
// device buffers earlier operator calls left with it: the "topology unchanged" cook allocates and builds nothing.
inline void gridFor(GridIndexedData& data, float voxelSize, IndexGridHandle& h, unsigned flags = HNS_GRID_DEFAULT) {
	if (h.get() && !(flags & HNS_GRID_HOST_ONLY) && hns_grid_voxel_size(h.get()) == voxelSize) {
		const int same = hns_grid_matches(h.get(), reinterpret_cast<const int32_t*>(data.pCoords()), data.size(), flags);
		if (same < 0) check(same);
		if (same == 1) return;
	}
	int err = HNS_OK;
	hns_grid* g = hns_grid_create(reinterpret_cast<const int32_t*>(data.pCoords()), data.size(), voxelSize, flags, &err);
	if (!g) check(err < 0 ? err : HNS_ERR_RUNTIME);
	h.reset(g);
}
}  // namespace detail
}  // namespace HNS

struct CombustionParams {  // reference src/Cuda/Kernels.cuh:6-13
	float expansionRate;
	float temperatureRelease;
	float buoyancyStrength;
	float ambientTemp;
	float vorticityScale;
	float factorScale;
};
static_assert(sizeof(CombustionParams) == sizeof(hns_combustion_params), "CombustionParams must match the C ABI struct");

// The grid as a NanoVDB NanoGrid<ValueOnIndex> buffer -- the reference's own handle contents (HNanoSolver.cu:375-384) --
// for code that still wants a nanovdb accessor. The returned storage is over-allocated; `data` points at the 32-byte
// aligned start of the `size`-byte grid inside it.
namespace HNS {
struct NanoVDBBuffer {
	std::vector<uint8_t> storage;
	uint8_t* data = nullptr;
	uint64_t size = 0;
};
}  // namespace HNS
inline HNS::NanoVDBBuffer ExportNanoVDB(const HNS::IndexGridHandle& handle) {
	HNS::NanoVDBBuffer b;
	HNS::detail::check(hns_grid_export_nanovdb(handle.get(), nullptr, 0, &b.size));
	b.storage.resize(b.size + 32);
	b.data = b.storage.data() + ((32 - (reinterpret_cast<uintptr_t>(b.storage.data()) & 31u)) & 31u);
	HNS::detail::check(hns_grid_export_nanovdb(handle.get(), b.data, b.size, &b.size));
	return b;
}

inline void CreateIndexGrid(HNS::GridIndexedData& data, HNS::IndexGridHandle& handle, const float voxelSize) {
	HNS::detail::gridFor(data, voxelSize, handle);
}

inline void Compute_Sim(HNS::GridIndexedData& data, const HNS::IndexGridHandle& handle, int iteration, float dt, float voxelSize,
                        const CombustionParams& params, bool hasCollision, void* stream) {
	auto f = data.fields();
	hns_combustion_params p;
	std::memcpy(&p, &params, sizeof(p));
	HNS::detail::check(hns_compute_sim(handle.get(), f.data(), (int)f.size(), iteration, dt, voxelSize, &p, hasCollision ? 1 : 0, stream));
}

// The reference rebuilds the index grid from data.pCoords() inside these four operators
// (Advection.cu:71,142; PressureProjection.cu:38,108); the optional handle lets a caller keep one across cooks.
inline void AdvectIndexGrid(HNS::GridIndexedData& data, const float dt, const float voxelSize, void* stream, const HNS::IndexGridHandle* handle = nullptr) {
	HNS::IndexGridHandle local;
	if (!handle) HNS::detail::gridFor(data, voxelSize, local);
	auto f = data.fields();
	HNS::detail::check(hns_advect_index_grid((handle ? *handle : local).get(), f.data(), (int)f.size(), dt, voxelSize, stream));
}

inline void AdvectIndexGridVelocity(HNS::GridIndexedData& data, const float dt, const float voxelSize, void* stream,
                                    const HNS::IndexGridHandle* handle = nullptr) {
	HNS::IndexGridHandle local;
	if (!handle) HNS::detail::gridFor(data, voxelSize, local);
	auto f = data.fields();
	HNS::detail::check(hns_advect_index_grid_velocity((handle ? *handle : local).get(), f.data(), (int)f.size(), dt, voxelSize, stream));
}

inline void ProjectNonDivergent(HNS::GridIndexedData& data, const size_t iterations, const float voxelSize, void* stream,
                                const HNS::IndexGridHandle* handle = nullptr) {
	HNS::IndexGridHandle local;
	if (!handle) HNS::detail::gridFor(data, voxelSize, local);
	auto f = data.fields();
	HNS::detail::check(hns_project_non_divergent((handle ? *handle : local).get(), f.data(), (int)f.size(), iterations, voxelSize, stream));
}

inline void Divergence(HNS::GridIndexedData& data, const float voxelSize, void* stream, const HNS::IndexGridHandle* handle = nullptr) {
	HNS::IndexGridHandle local;
	if (!handle) HNS::detail::gridFor(data, voxelSize, local);
	auto f = data.fields();
	HNS::detail::check(hns_divergence((handle ? *handle : local).get(), f.data(), (int)f.size(), voxelSize, stream));
}
