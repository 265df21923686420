// shim_check.cpp -- compiles integration/hns_shim.hpp (the reference-side binding) against this repo's container twin, whose
// interface equals the reference's HNS::GridIndexedData (src/Utils/GridData.hpp:16-166): the same template bodies that
// integration/hns_shim.cpp instantiates for the reference's own types. Built by the CPU suite (tests/test_integration.py);
// run on the GPU box it cooks one Compute_Sim and the single-purpose operators and prints checksums that the test compares
// with the Python path.
#include <cstdio>
#include <cstdlib>

#include "HNanoSolver.hpp"
#include "hns_shim.hpp"

static double checksum(const float* p, size_t n) {
	double s = 0.0;
	for (size_t i = 0; i < n; ++i) s += (double)p[i] * (double)(1 + i % 7);
	return s;
}

int main(int argc, char** argv) {
	const int R = argc > 1 ? atoi(argv[1]) : 16;
	const int nl = R / 8;
	HNS::GridIndexedData d;
	const size_t N = (size_t)nl * nl * nl * 512;
	d.allocateCoords(N);
	size_t i = 0;
	for (int lx = 0; lx < nl; ++lx)
		for (int ly = 0; ly < nl; ++ly)
			for (int lz = 0; lz < nl; ++lz)
				for (int n = 0; n < 512; ++n, ++i) d.pCoords()[i] = HNS::Coord{{lx * 8 + (n >> 6), ly * 8 + ((n >> 3) & 7), lz * 8 + (n & 7)}};
	for (const char* name : {"density", "temperature", "fuel", "waste", "flame"}) d.addValueBlock<float>(name, N);
	d.addValueBlock<HNS::Vec3f>("vel", N);
	for (size_t k = 0; k < N; ++k) {
		const HNS::Coord c = d.pCoords()[k];
		const float q = (float)((c.x() * 7 + c.y() * 3 + c.z()) % 13) / 13.0f;
		d.pValues<float>("density")[k] = q;
		d.pValues<float>("temperature")[k] = 23.0f + 10.0f * q;
		d.pValues<float>("fuel")[k] = 0.1f * q;
		d.pValues<HNS::Vec3f>("vel")[k] = HNS::Vec3f{{0.5f * q - 0.2f, 0.3f * q, 0.1f - 0.4f * q}};
	}
	try {
		hns_shim::GridHandle h;
		hns_shim::create_index_grid(d, h, 1.0f / R);
		hns_grid* first = h.g;
		hns_shim::create_index_grid(d, h, 1.0f / R);  // unchanged topology: the handle is kept
		if (h.g != first) return 3;
		const CombustionParams params{};
		hns_shim::compute_sim<HNS::Vec3f>(d, h, 5, 1.0f / 24.0f, 1.0f / R, params, false, nullptr);
		printf("compute_sim vel %.9e density %.9e\n", checksum(reinterpret_cast<float*>(d.pValues<HNS::Vec3f>("vel")), 3 * N), checksum(d.pValues<float>("density"), N));
		hns_shim::advect_index_grid<HNS::Vec3f>(d, 1.0f / 24.0f, 1.0f / R, nullptr);
		hns_shim::advect_index_grid_velocity<HNS::Vec3f>(d, 1.0f / 24.0f, 1.0f / R, nullptr);
		hns_shim::project_non_divergent<HNS::Vec3f>(d, 4, 1.0f / R, nullptr);
		d.addValueBlock<float>("divergence", N);
		hns_shim::divergence<HNS::Vec3f>(d, 1.0f / R, nullptr);
		printf("operators vel %.9e divergence %.9e\n", checksum(reinterpret_cast<float*>(d.pValues<HNS::Vec3f>("vel")), 3 * N), checksum(d.pValues<float>("divergence"), N));
		try {  // the reference's refusals arrive as the reference's exception types
			hns_shim::compute_sim<HNS::Vec3f>(d, h, 0, 1.0f / 24.0f, 1.0f / R, params, false, nullptr);
			return 4;
		} catch (const std::invalid_argument&) {
		}
	} catch (const std::exception& e) {
		fprintf(stderr, "shim_check: %s\n", e.what());
		return 2;
	}
	return 0;
}
