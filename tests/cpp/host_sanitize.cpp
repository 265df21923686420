// host_sanitize.cpp -- the host-only parts of libhns (hns_topology.cpp, hns_nanovdb.cpp) built with AddressSanitizer +
// UBSan and driven through their edge cases. The device entry points they reference are stubbed here: this binary never
// touches a GPU (GPU sanitizers are not available on the target pool; the host code is what can be sanitized).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hns_internal.hpp"

int hns_grid_upload(hns_grid*) { return HNS_ERR_NO_DEVICE; }
void hns_grid_free_device(hns_grid*) {}
int hns_grid_upload_schedule(hns_grid*) { return HNS_OK; }
int hns_grid_host_tables(const hns_grid* g) { return const_cast<hns_grid*>(g)->topo.have_tables ? HNS_OK : const_cast<hns_grid*>(g)->topo.build_tables(); }

#define REQUIRE(c)                                                              \
	do {                                                                        \
		if (!(c)) {                                                             \
			std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
			return 1;                                                           \
		}                                                                       \
	} while (0)

static std::vector<int32_t> coords_of(const std::vector<int32_t>& origins) {
	std::vector<int32_t> c;
	for (size_t l = 0; l < origins.size() / 3; ++l)
		for (int n = 0; n < 512; ++n) {
			c.push_back(origins[3 * l] + (n >> 6));
			c.push_back(origins[3 * l + 1] + ((n >> 3) & 7));
			c.push_back(origins[3 * l + 2] + (n & 7));
		}
	return c;
}

int main() {
	int err = 0;
	// leaves at the int32 edges, negative coordinates, several root tiles; enough leaves for the threaded validation path
	std::vector<int32_t> origins = {2147483640, 0, 0, -2147483647 - 1, 0, 0, 2147483640, 8, 0, 0, 2147483640, -2147483647 - 1, -8, -8, -8, 0, 0, 0, 4096, 0, 0};
	for (int x = 0; x < 16; ++x)
		for (int y = 0; y < 16; ++y)
			for (int z = 0; z < 10; ++z) {
				origins.push_back(80 + 8 * x);
				origins.push_back(8 * y);
				origins.push_back(8 * z);
			}
	const uint64_t nl = origins.size() / 3;
	std::vector<int32_t> coords = coords_of(origins);
	hns_grid* g = hns_grid_create(coords.data(), coords.size() / 3, 0.25f, HNS_GRID_HOST_ONLY, &err);
	REQUIRE(g && err == HNS_OK && hns_grid_leaf_count(g) == nl);
	REQUIRE(hns_grid_matches(g, coords.data(), coords.size() / 3, HNS_GRID_DEFAULT) == 1);
	coords[3 * (512 * 2000 + 17) + 1] += 1;
	REQUIRE(hns_grid_matches(g, coords.data(), coords.size() / 3, HNS_GRID_DEFAULT) == HNS_ERR_TOPOLOGY);
	REQUIRE(hns_grid_create(coords.data(), coords.size() / 3, 0.25f, HNS_GRID_HOST_ONLY, &err) == nullptr && err == HNS_ERR_TOPOLOGY);
	coords[3 * (512 * 2000 + 17) + 1] -= 1;
	std::vector<int32_t> nbr(nl * 27);
	REQUIRE(hns_grid_neighbor_table(g, nbr.data()) == HNS_OK);
	REQUIRE(nbr[0 * 27 + 13] == 0 && nbr[0 * 27 + 22] == -1);  // the leaf at x = INT32_MAX - 7 has no +x neighbour (no wrap-around)
	const int32_t probes[] = {2147483647, 7, 7, -2147483647 - 1, 0, 0, 2147483647, 2147483647, 2147483647, -1, -1, -1, 0, 0, 0};
	uint64_t off[5];
	REQUIRE(hns_grid_offsets(g, probes, 5, off) == HNS_OK);
	REQUIRE(off[0] == 512 && off[1] == 513 && off[2] == 0 && off[3] == 4 * 512 + 512 && off[4] == 5 * 512 + 1);
	std::vector<int32_t> back(coords.size());
	REQUIRE(hns_grid_coords(g, back.data()) == HNS_OK && back == coords);
	// NanoVDB export: size query, exact-size buffer (any overrun trips ASan), too-small and misaligned buffers
	uint64_t size = 0;
	REQUIRE(hns_grid_export_nanovdb(g, nullptr, 0, &size) == HNS_OK && size > 0);
	void* buf = std::aligned_alloc(32, (size + 31) / 32 * 32);
	REQUIRE(hns_grid_export_nanovdb(g, buf, size, &size) == HNS_OK);
	REQUIRE(std::memcmp(buf, "NanoVDB0", 8) == 0);
	REQUIRE(hns_grid_export_nanovdb(g, buf, size - 1, &size) == HNS_ERR_INVALID_ARGUMENT);
	REQUIRE(hns_grid_export_nanovdb(g, (char*)buf + 8, size, &size) == HNS_ERR_INVALID_ARGUMENT);
	std::free(buf);
	REQUIRE(hns_grid_set_active_leaves(g, nl + 1) == HNS_ERR_INVALID_ARGUMENT && hns_grid_set_active_leaves(g, 3) == HNS_OK);
	REQUIRE(hns_grid_set_outside_element(g, nl * 512) == HNS_ERR_INVALID_ARGUMENT && hns_grid_set_outside_element(g, 77) == HNS_OK);
	hns_grid_destroy(g);
	// duplicates, misaligned origins, empty grid, odd counts
	std::vector<int32_t> dup = {0, 0, 0, 8, 0, 0, 0, 0, 0};
	REQUIRE(hns_grid_create_from_leaves(dup.data(), 3, 1.0f, HNS_GRID_HOST_ONLY, &err) == nullptr && err == HNS_ERR_TOPOLOGY);
	std::vector<int32_t> mis = {0, 0, 0, 8, 3, 0};
	REQUIRE(hns_grid_create_from_leaves(mis.data(), 2, 1.0f, HNS_GRID_HOST_ONLY, &err) == nullptr && err == HNS_ERR_TOPOLOGY);
	g = hns_grid_create_from_leaves(nullptr, 0, 1.0f, HNS_GRID_HOST_ONLY, &err);
	REQUIRE(g && hns_grid_voxel_count(g) == 0);
	REQUIRE(hns_grid_export_nanovdb(g, nullptr, 0, &size) == HNS_OK && size == 672 + 64 + 96);
	buf = std::aligned_alloc(32, 832);
	REQUIRE(hns_grid_export_nanovdb(g, buf, size, &size) == HNS_OK);
	std::free(buf);
	hns_grid_destroy(g);
	REQUIRE(hns_grid_create(coords.data(), 700, 1.0f, HNS_GRID_HOST_ONLY, &err) == nullptr && err == HNS_ERR_TOPOLOGY);
	REQUIRE(hns_grid_create(nullptr, 512, 1.0f, HNS_GRID_HOST_ONLY, &err) == nullptr && err == HNS_ERR_INVALID_ARGUMENT);
	// the OpenVDB-free leaf I/O (hns_leafio.cpp): int32-edge origins, exact-capacity outputs, masks, fills
	{
		const std::vector<int32_t> lo = {2147483640, 0, 0, -2147483648, -8, 8, 0, 0, 0};
		std::vector<unsigned char> masks(3 * 64, 0);
		masks[0] = 1, masks[64 + 63] = 0x80, masks[128 + 9] = 0x10;
		uint64_t n_out = 0;
		REQUIRE(hns_dilate_leaves(lo.data(), 3, masks.data(), 9, nullptr, 0, &n_out) == HNS_OK && n_out > 3);
		std::vector<int32_t> dil(3 * n_out);
		REQUIRE(hns_dilate_leaves(lo.data(), 3, masks.data(), 9, dil.data(), n_out, &n_out) == HNS_OK);
		REQUIRE(hns_dilate_leaves(lo.data(), 3, masks.data(), 9, dil.data(), n_out - 1, &n_out) == HNS_ERR_INVALID_ARGUMENT);
		uint64_t n_uni = 0;
		REQUIRE(hns_union_leaves(dil.data(), n_out, lo.data(), 3, nullptr, 0, &n_uni) == HNS_OK && n_uni == n_out);
		std::vector<float> src(2 * 512 * 3, 2.5f), flat(n_out * 512 * 3);
		REQUIRE(hns_gather_leaves(dil.data(), n_out, lo.data(), 2, src.data(), 3, HNS_FILL_SDF, flat.data()) == HNS_OK);
		std::vector<std::vector<float>> leaves(n_out, std::vector<float>(512 * 3));
		std::vector<float*> ptrs;
		for (auto& v : leaves) ptrs.push_back(v.data());
		REQUIRE(hns_scatter_leaves(flat.data(), n_out, 3, ptrs.data()) == HNS_OK);
		const int32_t bad[3] = {4, 0, 0};
		REQUIRE(hns_dilate_leaves(bad, 1, nullptr, 1, nullptr, 0, &n_out) == HNS_ERR_TOPOLOGY);
	}
	std::puts("host_sanitize OK");
	return 0;
}
