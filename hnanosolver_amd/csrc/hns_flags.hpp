// hns_flags.hpp -- sequence-numbered flags between ranks (kernels of different processes / devices): system-scope release
// stores and acquire loads on fine-grained device memory, bounded waits. Used by the one-sided halo transport (hns_dist_*.hip)
// and by the kernels of a chained rank, which write their boundary values into the peers' ghost voxels themselves (PhaseMirror below).
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

// write-through stores at system scope: at their destination (a peer's memory included) when s_waitcnt vmcnt(0) returns
typedef float chain_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_through(float* p, chain_v4f v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(float* p, float4 v) { store_through(p, chain_v4f{v.x, v.y, v.z, v.w}); }
__device__ __forceinline__ void store_through(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// What a kernel of a "chained" multi-GPU rank needs besides its own arguments (by value; built by hns_dist_*.hip). Every kernel of
// the substep is ONE launch over the owned leaves that delivers its own halo: a workgroup of a boundary leaf (local leaves
// [0, n_boundary), first in the launch order) waits until every peer has completed its previous launch -- the peers' boundary
// values of that launch are then in this rank's ghost voxels, and the peers no longer read what this launch is about to write
// into theirs --, computes like any other workgroup, and stores the voxels a peer can read (the plan's region of this kernel's
// exchange type) into that peer's ghost copy of the leaf, through the peer's memory mapped here, and waits for those stores
// before it ends. A rank's "launch seq complete" flag goes up on every peer when its NEXT launch starts (first workgroup).
struct PhaseMirror {
	int n_boundary, n_peers;
	const int* first;           // [n_boundary + 1]: entries of boundary leaf l are first[l] .. first[l + 1]  (region type of this launch)
	const int2* entry;          // {peer index, that peer's local index of its ghost copy of the leaf}
	const unsigned char* mask;  // 64 bytes per entry: byte x*8+y, bit z = the voxel travels; null: every listed leaf travels whole
	char* peer_arena[kMirrorMaxPeers];               // the peers' field memory (mapped here) ...
	unsigned long long peer_unit[kMirrorMaxPeers];   // ... in which scalar field number i starts at i * unit bytes
	uint32_t* peer_flag[kMirrorMaxPeers];            // the peer's "launch complete" flag of this rank
	int peer_rank[kMirrorMaxPeers];
	const uint32_t* my_flags;   // this rank's flag page
	int* status;
	uint32_t seq;               // number of this launch (all ranks count alike); boundary workgroups wait for the peers' seq - 1
	int out_unit[8];            // which unit of the field memory each output array of this launch starts at
};
struct NoMirror {};  // the same kernels on a single GPU: every chain_* call below compiles to nothing

// The blocked boundary sweep of an EXCHANGED pressure loop (RCCL / loopback / local transports; hns_dist_substep.hip: sor_block_exchanged) packs its own message: the voxels a peer reads of
// a boundary leaf go into that peer's send buffer in the message order (leaves in region order, rows x*8+y ascending, z ascending, travelling voxels only) as the block is stored --
// plain stores, the send that follows reads them in stream order -- and the separate pack launch (7.5 us in a chain of four latency-bound launches per exchange) is gone.
struct PackMirror {
	int n_boundary;                 // local leaves [0, n_boundary) may have entries
	const int* first;               // [n_boundary + 1]: entries of leaf l are first[l] .. first[l + 1]
	const int2* entry;              // {peer index, voxel offset of the leaf's first travelling voxel in that peer's message}
	const unsigned char* mask;      // 64 bytes per entry: byte x*8+y, bit z = the voxel travels
	const unsigned short* row_pre;  // 64 per entry: travelling voxels of the leaf in front of row x*8+y
	float* msg[kMirrorMaxPeers];    // the peers' send buffers of this exchange
};

__device__ __forceinline__ float* chain_out(const PhaseMirror& m, int peer, int out) { return (float*)(m.peer_arena[peer] + (size_t)m.out_unit[out] * m.peer_unit[peer]); }

// first thing in every workgroup (before any ghost voxel is read; multi-wave workgroups need a barrier after it)
__device__ __forceinline__ void chain_begin(const NoMirror&, int) {}
__device__ __forceinline__ void chain_begin(const PackMirror&, int) {}
__device__ __forceinline__ void chain_begin(const PhaseMirror& m, int leaf) {
	leaf = __builtin_amdgcn_readfirstlane(leaf);  // (workgroup-uniform: keeps everything derived from it in scalar registers)
	// this launch has started, so the previous launch of this rank is complete (every boundary workgroup waited for its
	// write-through stores before it ended): tell the peers, before anything here waits for them
	if (blockIdx.x == 0 && (int)threadIdx.x < m.n_peers) flag_store_relaxed(m.peer_flag[threadIdx.x], m.seq - 1u);
	if (leaf < m.n_boundary && (int)threadIdx.x < m.n_peers) flag_wait_relaxed(m.my_flags + kFlagSweep + m.peer_rank[threadIdx.x], m.seq - 1u, m.status);
	asm volatile("" ::: "memory");  // the loads of the kernel stay below the poll
}
// last thing in every workgroup
__device__ __forceinline__ void chain_end(const NoMirror&, int) {}
__device__ __forceinline__ void chain_end(const PackMirror&, int) {}
__device__ __forceinline__ void chain_end(const PhaseMirror& m, int leaf) {
	if (__builtin_amdgcn_readfirstlane(leaf) < m.n_boundary) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// a z-row (8 floats) of float output `out`
__device__ __forceinline__ void chain_store_row(const NoMirror&, int, int, int, float4, float4) {}
__device__ __forceinline__ void chain_store_row(const PhaseMirror& m, int out, int leaf, int row, float4 lo, float4 hi) {
	leaf = __builtin_amdgcn_readfirstlane(leaf);  // (workgroup-uniform: the table walk below stays in scalar registers)
	if (leaf >= m.n_boundary) return;
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		const unsigned bits = m.mask ? m.mask[(size_t)e * 64 + row] : 0xFFu;
		float* r = chain_out(m, t.x, out) + (size_t)t.y * 512 + row * 8;
		if (bits == 0xFFu) {
			store_through(r, lo);
			store_through(r + 4, hi);
		} else if (bits) {
			const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
			for (int z = 0; z < 8; ++z)
				if (bits >> z & 1) store_through(r + z, v[z]);
		}
	}
}
// voxel n (x<<6|y<<3|z) of an output with NC components per voxel
template <int NC>
__device__ __forceinline__ void chain_store_voxel(const NoMirror&, int, int, int, const float*) {}
template <int NC>
__device__ __forceinline__ void chain_store_voxel(const PhaseMirror& m, int out, int leaf, int n, const float* v) {
	leaf = __builtin_amdgcn_readfirstlane(leaf);
	if (leaf >= m.n_boundary) return;
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		if (m.mask && !(m.mask[(size_t)e * 64 + (n >> 3)] >> (n & 7) & 1)) continue;
		float* r = chain_out(m, t.x, out) + ((size_t)t.y * 512 + n) * NC;
#pragma unroll
		for (int c = 0; c < NC; ++c) store_through(r + c, v[c]);
	}
}
// 16 bytes at float offset `o` of a whole-leaf output with NC components per voxel (regions of whole leaves only)
template <int NC>
__device__ __forceinline__ void chain_store_leaf16(const NoMirror&, int, int, int, float4) {}
template <int NC>
__device__ __forceinline__ void chain_store_leaf16(const PhaseMirror& m, int out, int leaf, int o, float4 v) {
	leaf = __builtin_amdgcn_readfirstlane(leaf);
	if (leaf >= m.n_boundary) return;
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		store_through(chain_out(m, t.x, out) + (size_t)t.y * 512 * NC + o, v);
	}
}

}  // namespace hns
