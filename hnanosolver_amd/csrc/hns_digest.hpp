// hns_digest.hpp -- the digest a CHECKED resident field is held to (hns_compute_sim_resident, include/hns.h): the sum, mod 2^64, over the 16-byte pieces of
// the field of hns_digest_piece(piece number, its two 8-byte words). Every bit of a piece and its position matter; the sum is order-independent, so the DEVICE
// takes it of the buffer a host array was downloaded from in one pass at memory speed (k_field_digest, hns_pointwise.hip) and the HOST takes the same number of
// the array it is handed back, on as many threads as it likes (host_digest, hns_api.hip).
#pragma once
#include <cstdint>

#ifdef __HIPCC__
__host__ __device__
#endif
inline uint64_t hns_digest_piece(uint64_t i, uint64_t w0, uint64_t w1) {
	uint64_t t = (w0 ^ (i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull)) * 0xD6E8FEB86659FD93ull;
	t ^= t >> 32;
	const uint64_t u = (w1 + t) * 0xA0761D6478BD642Full;
	return u ^ (u >> 29);
}
inline uint64_t hns_digest_finish(uint64_t sum, uint64_t count) {
	uint64_t h = (sum ^ (count * 0xC2B2AE3D27D4EB4Full)) * 0x9FB21C651E98DF25ull;
	h ^= h >> 31;
	return h ? h : 1;  // (0 = "no digest taken")
}
// adds the digest sum of `count` floats at `field` to *d_out (a zeroed 64-bit word in device memory), on `stream`
extern "C" __attribute__((visibility("hidden"))) int hns_field_digest(const float* field, uint64_t count, unsigned long long* d_out, void* stream);
