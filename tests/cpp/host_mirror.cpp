// host_mirror.cpp -- exercises hnanosolver_amd/host/HNanoSolver.hpp (the C++ twin of hnanosolver_amd/api.py).
//   host_mirror                      CPU-only checks: container semantics, topology validation, exception mapping
//   host_mirror gpu <in.bin> <out.bin>   reads {int64 N, coords int32[N*3], vel float[N*3], density float[N]}, runs
//                                    ProjectNonDivergent(20) then AdvectIndexGrid, writes {vel, density}
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "HNanoSolver.hpp"

#define REQUIRE(c)                                                     \
	do {                                                               \
		if (!(c)) {                                                    \
			std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
			return 1;                                                  \
		}                                                              \
	} while (0)

static int cpu_checks() {
	HNS::GridIndexedData d;
	REQUIRE(d.size() == 0 && d.pCoords() == nullptr);
	REQUIRE(d.allocateCoords(1024));
	REQUIRE(d.addValueBlock<float>("density", 1024));
	REQUIRE(d.addValueBlock<HNS::Vec3f>("vel", 1024));
	REQUIRE(d.addValueBlock<float>("temperature", 1024));
	REQUIRE(!d.addValueBlock<float>("density", 1024));
	REQUIRE(d.numValueBlocks() == 3);
	auto fl = d.getBlocksOfType<float>();
	REQUIRE(fl.size() == 2 && fl[0] == "density" && fl[1] == "temperature");
	REQUIRE(d.getBlocksOfType<HNS::Vec3f>().size() == 1);
	REQUIRE(d.pValues<float>("vel") == nullptr && d.pValues<HNS::Vec3f>("vel") != nullptr && d.pValues<float>("nope") == nullptr);
	// two leaves, leaf-dense
	for (int l = 0; l < 2; ++l)
		for (int n = 0; n < 512; ++n) d.pCoords()[l * 512 + n] = HNS::Coord{{8 * l + (n >> 6), (n >> 3) & 7, n & 7}};
	HNS::IndexGridHandle h;
	int err = 0;
	hns_grid* g = hns_grid_create(reinterpret_cast<const int32_t*>(d.pCoords()), d.size(), 0.5f, HNS_GRID_HOST_ONLY, &err);
	REQUIRE(g && err == HNS_OK);
	h.reset(g);
	REQUIRE(hns_grid_leaf_count(h.get()) == 2);
	const int32_t q[6] = {1, 2, 3, 8, 2, 3};
	uint64_t off[2];
	REQUIRE(hns_grid_offsets(h.get(), q, 2, off) == HNS_OK && off[0] == 1 + 64 + 16 + 3 && off[1] == 512 + 1 + 16 + 3);
	// the grid in the reference's own handle format (NanoGrid<ValueOnIndex>): header, tree, 1 upper, 1 lower, 2 leaves
	HNS::NanoVDBBuffer nb = ExportNanoVDB(h);
	REQUIRE(nb.size == 672 + 64 + 96 + 32 + 270400 + 33856 + 2 * 96 && (reinterpret_cast<uintptr_t>(nb.data) & 31u) == 0);
	REQUIRE(std::memcmp(nb.data, "NanoVDB0", 8) == 0);
	uint64_t value_count = 0;
	std::memcpy(&value_count, nb.data + 656, 8);
	REQUIRE(value_count == 1 + 2 * 512);
	REQUIRE(hns_grid_matches(h.get(), reinterpret_cast<const int32_t*>(d.pCoords()), d.size(), HNS_GRID_DEFAULT) == 1);
	REQUIRE(hns_grid_matches(h.get(), reinterpret_cast<const int32_t*>(d.pCoords()), 512, HNS_GRID_DEFAULT) == 0);
	// exception mapping: invalid_argument for bad scalars (reference HNanoSolver.cu:12-23)
	CombustionParams p{0.1f, 0.5f, 1.0f, 23.0f, 1.0f, 0.5f};
	bool caught = false;
	try {
		Compute_Sim(d, h, 10, 0.04f, -1.0f, p, false, nullptr);
	} catch (const std::invalid_argument&) {
		caught = true;
	}
	REQUIRE(caught);
	// runtime_error for a missing combustion field or, on a machine without a GPU, for the missing device
	caught = false;
	try {
		Compute_Sim(d, h, 10, 0.04f, 0.5f, p, false, nullptr);
	} catch (const std::runtime_error&) {
		caught = true;
	}
	REQUIRE(caught);
	// non leaf-dense coordinates
	d.pCoords()[700].v[0] += 1;
	caught = false;
	try {
		HNS::IndexGridHandle h2;
		HNS::detail::gridFor(d, 1.0f, h2, HNS_GRID_HOST_ONLY);
	} catch (const std::runtime_error& e) {
		caught = std::strstr(e.what(), "leaf-dense") != nullptr;
	}
	REQUIRE(caught);
	std::puts("host_mirror cpu checks OK");
	return 0;
}

static int gpu_run(const char* in, const char* out) {
	FILE* f = std::fopen(in, "rb");
	REQUIRE(f);
	int64_t N = 0;
	REQUIRE(std::fread(&N, sizeof(N), 1, f) == 1);
	HNS::GridIndexedData d;
	d.allocateCoords((size_t)N);
	d.addValueBlock<float>("density", (size_t)N);
	d.addValueBlock<HNS::Vec3f>("vel", (size_t)N);
	REQUIRE(std::fread(d.pCoords(), 12, (size_t)N, f) == (size_t)N);
	REQUIRE(std::fread(d.pValues<HNS::Vec3f>("vel"), 12, (size_t)N, f) == (size_t)N);
	REQUIRE(std::fread(d.pValues<float>("density"), 4, (size_t)N, f) == (size_t)N);
	std::fclose(f);
	const float vs = 1.0f / 32.0f;
	ProjectNonDivergent(d, 20, vs, nullptr);
	AdvectIndexGrid(d, 1.0f / 24.0f, vs, nullptr);
	f = std::fopen(out, "wb");
	REQUIRE(f);
	std::fwrite(d.pValues<HNS::Vec3f>("vel"), 12, (size_t)N, f);
	std::fwrite(d.pValues<float>("density"), 4, (size_t)N, f);
	std::fclose(f);
	return 0;
}

int main(int argc, char** argv) {
	if (argc == 4 && std::strcmp(argv[1], "gpu") == 0) return gpu_run(argv[2], argv[3]);
	return cpu_checks();
}
