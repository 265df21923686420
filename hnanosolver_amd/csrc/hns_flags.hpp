// hns_flags.hpp -- sequence-numbered flags between ranks (kernels of different processes / devices): system-scope release
// stores and acquire loads on fine-grained device memory, bounded waits. Used by the one-sided halo transport (hns_dist.hip)
// and by the SOR sweep that writes its boundary rows into the peers' ghost voxels itself (hns_pressure.hip: k_rbgs_pair_mirror).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hns {

constexpr int kFlagSlots = 512;                      // ranks a flag page has room for
// the page of a rank: [0, 512) "rank q is ready to receive exchange seq", [512, 1024) "what q sent in exchange seq has landed",
// [1024, 1536) "q's sweep number seq is complete: its boundary rows are in my ghost voxels and it no longer reads the other buffer"
constexpr int kFlagReady = 0, kFlagLanded = kFlagSlots, kFlagSweep = 2 * kFlagSlots, kFlagWords = 3 * kFlagSlots;
constexpr long long kFlagWaitTicks = 2000000000LL;   // 20 s of the 100 MHz wall clock, then give up (status word, no hang)

__device__ __forceinline__ uint32_t flag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// sequence numbers wrap: "reached" = not behind. `status` is host-mapped: set once a wait has run out, and read by every
// later wait so that one lost peer costs one timeout, not one per exchange.
__device__ __forceinline__ bool flag_wait(const uint32_t* flag, uint32_t seq, volatile int* status) {
	if ((int32_t)(flag_load(flag) - seq) >= 0) return true;
	const long long t0 = wall_clock64();
	for (unsigned spins = 1;; ++spins) {
		if ((int32_t)(flag_load(flag) - seq) >= 0) return true;
		__builtin_amdgcn_s_sleep(2);
		if ((spins & 1023u) == 0 && (*status != 0 || wall_clock64() - t0 > kFlagWaitTicks)) {
			*status = 1;
			return false;
		}
	}
}

// The same without fences, for waves inside a large kernel: an acquire at system scope invalidates the L2 of the wave's XCD and
// a release writes it back -- per wave, in the middle of a sweep, that costs more than the sweep (measured: 350 us instead of
// 40). The caller orders its payload itself: write-through (sc0 sc1) stores, s_waitcnt vmcnt(0), then flag_store_relaxed; on
// the reading side the payload lines cannot be in a cache of this device yet (nobody reads them before the flag is seen, and
// the kernel boundary before invalidated them), so plain loads after flag_wait_relaxed fetch them from memory.
__device__ __forceinline__ void flag_store_relaxed(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool flag_wait_relaxed(const uint32_t* flag, uint32_t seq, volatile int* status) {
	if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0) return true;
	const long long t0 = wall_clock64();
	for (unsigned spins = 1;; ++spins) {
		if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0) return true;
		__builtin_amdgcn_s_sleep(2);
		if ((spins & 1023u) == 0 && (*status != 0 || wall_clock64() - t0 > kFlagWaitTicks)) {
			*status = 1;
			return false;
		}
	}
}

constexpr int kMirrorMaxPeers = 16;

// What the mirroring SOR sweep needs besides the sweep's own arguments (by value; built by hns_dist.hip).
struct RbgsMirror {
	int n_boundary;             // local leaves [0, n_boundary) have copies (ghost leaves) on other ranks
	int n_peers;
	const int* first;           // [n_boundary + 1]: entries of boundary leaf l are first[l] .. first[l + 1]
	const int2* entry;          // {peer index, that peer's local index of its ghost copy of the leaf}
	const unsigned char* mask;  // 64 bytes per entry: byte x*8+y, bit z = the voxel travels (within reach 2 of a voxel the peer owns)
	float* peer_out[kMirrorMaxPeers];     // the sweep's destination array on every peer (mapped here)
	uint32_t* peer_flag[kMirrorMaxPeers];  // the peer's "sweep complete" flag of this rank
	int peer_rank[kMirrorMaxPeers];
	const uint32_t* my_flags;   // this rank's flag page
	uint32_t seq;               // number of this sweep (all ranks count alike); boundary waves wait for the peers' seq - 1
	unsigned* count;            // device counter: boundary records done in this launch
	unsigned n_boundary_records;
	unsigned head_records;      // a multiple of 8: the records before it (the boundary leaves' among them) keep their place when a sweep walks backwards
	int* status;
};

}  // namespace hns
