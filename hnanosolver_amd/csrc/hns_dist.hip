// hns_dist.hip -- the leaf-partitioned multi-GPU core substep (the reference is single-GPU, so this is new
// design): one rank per GPU owns a contiguous range of the NanoVDB-ordered leaf list, keeps one layer of ghost leaves,
// and refreshes exactly the ghost VOXELS the next kernels can read, over RCCL point-to-point (xGMI) on a communication
// stream of its own, underneath the kernels that do not need them.
//
//   local leaf order   [ B: owned leaves some other rank mirrors | I: the other owned leaves | G: ghosts, grouped by owner ]
//   launch ranges      B, I, B+I (owned) and B+I+G (all) are four active ranges over the same local leaf list
//   a kernel           runs on B first; the regions of B the peers read are packed and handed to the communication
//                      stream; the kernel then runs on I while the messages travel (a ghost is only ever needed by the
//                      NEXT kernel)
//   halo regions       per exchange the set of voxels of a ghost leaf within the stencil's reach of a voxel the receiver
//                      owns: L1 distance 1 for u* (divergence) and the final p (gradient), 2k-1 for div and 2k for p when
//                      the pressure loop exchanges every k-th sweep (a fused red+black sweep moves information two voxels
//                      and the ghost leaves are swept locally in between), the whole leaf for the advection inputs
//                      (back-traces reach up to a leaf away), and one voxel -- element 0 of global leaf 0 -- for the
//                      mirror advect_scalars' out-of-domain taps read (reference Kernel.cu:133,192,225).
//                      Both sides derive the same 512-bit masks from the global leaf list; nothing but payload is sent.
//   transports         RCCL (ncclSend/ncclRecv in one group per exchange; one process per GPU), or "local": every rank of
//                      the decomposition lives in this process on one device and a message is a device copy out of the
//                      peer's send buffer -- the same plan, kernels, streams and events without a wire; used by the tests
//                      (8 emulated ranks on one GPU) and to measure the per-rank overhead before any wire time.
//                      "ipc": one process per GPU like RCCL, but one-sided: field memory, message buffers and a page of
//                      flags of every rank are mapped into its peers (hipIpc*), a rank PUTS its messages straight into
//                      the peer's receive buffer or ghost voxels with a copy kernel and the two sides meet through
//                      sequence-numbered flags (ready-to-receive / landed) polled by tiny kernels: no RCCL launch, no
//                      rendezvous kernel; verified between processes sharing one GPU.
//
// Owned results are bit-identical to the single-domain run: every exchange sits where the single-GPU code has a kernel
// boundary that a stencil crosses, and a ghost voxel is never read beyond the depth its last refresh made valid.
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is opened on first use (see Rccl below)

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <memory>
#include <initializer_list>
#include <utility>

#include "hns_device.hpp"
#include "hns_flags.hpp"

#define HNS_TRY(call)                    \
	do {                                 \
		int rc__ = (call);               \
		if (rc__ != HNS_OK) return rc__; \
	} while (0)

// RCCL is bound at run time, the first time a multi-process transport is asked for: libhns.so then carries no load-time
// dependency on the 500 MB library (single-GPU users never touch it), and a process that already holds an RCCL -- PyTorch
// ships its own librccl.so.1 -- keeps exactly one copy instead of two interposing each other.
namespace {
struct Rccl {
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	bool ok = false;
	std::string why;  // what went wrong, captured where it went wrong (dlerror() clears itself when read)
};
Rccl& rccl() {
	static Rccl r = [] {
		Rccl t;
		void* h = nullptr;
		for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
			if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
			const char* e = dlerror();
			t.why += std::string(t.why.empty() ? "" : "; ") + (e ? e : name);
		}
		if (!h) return t;
		t.why.clear();
		t.GetUniqueId = (decltype(t.GetUniqueId))dlsym(h, "ncclGetUniqueId");
		t.CommInitRank = (decltype(t.CommInitRank))dlsym(h, "ncclCommInitRank");
		t.CommDestroy = (decltype(t.CommDestroy))dlsym(h, "ncclCommDestroy");
		t.GroupStart = (decltype(t.GroupStart))dlsym(h, "ncclGroupStart");
		t.GroupEnd = (decltype(t.GroupEnd))dlsym(h, "ncclGroupEnd");
		t.Send = (decltype(t.Send))dlsym(h, "ncclSend");
		t.Recv = (decltype(t.Recv))dlsym(h, "ncclRecv");
		t.GetErrorString = (decltype(t.GetErrorString))dlsym(h, "ncclGetErrorString");
		t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.GroupStart && t.GroupEnd && t.Send && t.Recv && t.GetErrorString;
		if (!t.ok) t.why = "the library lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / GroupStart / GroupEnd / Send / Recv / GetErrorString";
		return t;
	}();
	return r;
}
int need_rccl(const char* who) {
	if (rccl().ok) return HNS_OK;
	hns::set_error("%s: librccl.so.1 could not be loaded: %s", who, rccl().why.c_str());
	return HNS_ERR_RUNTIME;
}
}  // namespace

#define HNS_NCCL(call)                                                                                         \
	do {                                                                                                       \
		ncclResult_t r__ = (call);                                                                             \
		if (r__ != ncclSuccess) {                                                                              \
			hns::set_error("%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r__), __FILE__, __LINE__);    \
			return HNS_ERR_HIP;                                                                                \
		}                                                                                                      \
	} while (0)

namespace hns {

// ---------------------------------------------------------------------------------------------------------------
// masked pack / unpack: one wave per listed leaf, lane = z-row (x*8+y), 8-bit z-mask per row
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ int wave_exclusive_scan(int v) {
	int s = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int t = __shfl_up(s, d, 64);
		if ((int)threadIdx.x >= d) s += t;
	}
	return s - v;
}

// loopback transport only: stands in for the time a message spends on the wire (option "dist_wire_us")
__global__ void k_wire_delay(long long ticks) {
	const long long t0 = wall_clock64();  // constant 100 MHz clock
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

template <int NCOMP, bool PACK>
__global__ __launch_bounds__(64) void k_halo_copy(float* __restrict__ field, const int* __restrict__ leaf, const unsigned char* __restrict__ mask,
                                                  const int* __restrict__ off, float* __restrict__ msg) {
	const int i = blockIdx.x, l = threadIdx.x;
	const unsigned m = mask[(size_t)i * 64 + l];
	const int base = off[i] + wave_exclusive_scan(__popc(m));
	float* f = field + ((size_t)leaf[i] * 512 + l * 8) * NCOMP;
	float* q = msg + (size_t)base * NCOMP;
	int c = 0;
#pragma unroll
	for (int z = 0; z < 8; ++z) {
		if (m >> z & 1) {
#pragma unroll
			for (int k = 0; k < NCOMP; ++k) {
				if (PACK)
					q[c * NCOMP + k] = f[z * NCOMP + k];
				else
					f[z * NCOMP + k] = q[c * NCOMP + k];
			}
			++c;
		}
	}
}

// The same for ALL peers of a rank in one launch (a rank of a 3-d decomposition talks to several: 7 in the 8-range plume):
// entry i carries its peer, whose message base and region size come by value.
constexpr int kMaxBatchPeers = 16;
struct PeerMsgs {
	float* base[kMaxBatchPeers];
	int voxels[kMaxBatchPeers];
};

template <int NCOMP, bool PACK>
__global__ __launch_bounds__(64) void k_halo_copy_all(float* __restrict__ field, const int* __restrict__ leaf, const unsigned char* __restrict__ mask,
                                                      const int* __restrict__ off, const int* __restrict__ peer, const PeerMsgs msgs, const int comps_before) {
	const int i = blockIdx.x, l = threadIdx.x;
	const unsigned m = mask[(size_t)i * 64 + l];
	const int p = peer[i];
	const int base = off[i] + wave_exclusive_scan(__popc(m));
	float* f = field + ((size_t)leaf[i] * 512 + l * 8) * NCOMP;
	float* q = msgs.base[p] + (size_t)comps_before * (size_t)msgs.voxels[p] + (size_t)base * NCOMP;
	int c = 0;
#pragma unroll
	for (int z = 0; z < 8; ++z) {
		if (m >> z & 1) {
#pragma unroll
			for (int k = 0; k < NCOMP; ++k) {
				if (PACK)
					q[c * NCOMP + k] = f[z * NCOMP + k];
				else
					f[z * NCOMP + k] = q[c * NCOMP + k];
			}
			++c;
		}
	}
}

// ---- one-sided transport (hipIpc-mapped peers): sequence-numbered flags (hns_flags.hpp), bounded waits ----
constexpr int kIpcMaxSegs = 32, kIpcMaxPeers = kMirrorMaxPeers;

struct IpcPeers {
	int n;
	uint32_t* theirs_ready[kIpcMaxPeers];   // the peer's flag "rank <me> is ready to receive"   (in the PEER's memory)
	uint32_t* theirs_landed[kIpcMaxPeers];  // the peer's flag "what rank <me> sent has landed"   (in the PEER's memory)
	int rank[kIpcMaxPeers];
};

// "my receive side of exchange `seq` may be written": told to every peer; then wait for the same from every peer. One wave
// does all the waiting of a rank: waiting inside the copy kernel's workgroups filled the device with spinning waves (four
// processes of a 66k-leaf plume on one GPU: nothing else could be scheduled, every bounded wait ran out).
__global__ void k_ipc_ready(const IpcPeers peers, const uint32_t* __restrict__ my_flags, const uint32_t seq, int* status) {
	if ((int)threadIdx.x < peers.n) {
		flag_store(peers.theirs_ready[threadIdx.x], seq);
		flag_wait(my_flags + peers.rank[threadIdx.x], seq, status);
	}
}

struct IpcSegs {
	int n;
	float* dst[kIpcMaxSegs];  // in the peer's memory
	const float* src[kIpcMaxSegs];
	unsigned floats[kIpcMaxSegs];
	unsigned wg0[kIpcMaxSegs + 1];  // first workgroup of every segment
};

// Copies the segments into the peers' memory, 4,096 floats per workgroup (the receivers are ready: k_ipc_ready ran).
template <bool FENCE>  // (FENCE: the destination is another process's / device's memory; false: the loopback stand-in, this rank's own buffers)
__global__ __launch_bounds__(256) void k_ipc_put(const IpcSegs segs) {
	int s = 0;
	while (s + 1 < segs.n && blockIdx.x >= segs.wg0[s + 1]) ++s;
	const size_t first = (size_t)(blockIdx.x - segs.wg0[s]) * 4096u;
	const unsigned n = segs.floats[s];
	const float* __restrict__ src = segs.src[s];
	float* __restrict__ dst = segs.dst[s];
	if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0 && (n & 3u) == 0) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const size_t i = first + ((size_t)j * 256u + threadIdx.x) * 4u;
			if (i < n) *(float4*)(dst + i) = *(const float4*)(src + i);
		}
	} else {
#pragma unroll 4
		for (int j = 0; j < 16; ++j) {
			const size_t i = first + (size_t)j * 256u + threadIdx.x;
			if (i < n) dst[i] = src[i];
		}
	}
	if (FENCE) __threadfence_system();
}

// after the puts (kernel boundary + fence): tell every peer its message has landed, then wait for theirs
__global__ void k_ipc_landed(const IpcPeers peers, const uint32_t* __restrict__ my_flags, const uint32_t seq, int* status) {
	__threadfence_system();
	if ((int)threadIdx.x < peers.n) {
		flag_store(peers.theirs_landed[threadIdx.x], seq);
		flag_wait(my_flags + kFlagLanded + peers.rank[threadIdx.x], seq, status);
	}
}

// Chained substep, the two advection kernels: they run as they are (512 threads per leaf at a 64-register cap: the chain code
// inside them cost three spilled registers and 25 % of their speed, measured) between a one-wave gate (k_sweep_wait: "peers,
// my previous launch is complete" + wait for theirs) and this kernel, which copies what the peers read of the boundary
// leaves' new values into the peers' ghost voxels: one wave per boundary leaf, lane = z-row.
template <int NC>
__global__ __launch_bounds__(64) void k_chain_mirror(const PhaseMirror m, const int n_out, const float* f0, const float* f1, const float* f2, const float* f3,
                                                     const float* f4, const float* f5, const float* f6, const float* f7) {
	const int leaf = blockIdx.x, l = threadIdx.x;
	const float* fields[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
	const int e1 = m.first[leaf + 1];
	for (int e = m.first[leaf]; e < e1; ++e) {
		const int2 t = m.entry[e];
		const unsigned bits = m.mask ? m.mask[(size_t)e * 64 + l] : 0xFFu;
		if (!bits) continue;
#pragma unroll 1
		for (int o = 0; o < n_out; ++o) {
			const float* src = fields[o] + ((size_t)leaf * 512 + l * 8) * NC;
			float* dst = chain_out(m, t.x, o) + ((size_t)t.y * 512 + l * 8) * NC;
			if (bits == 0xFFu) {
#pragma unroll
				for (int q = 0; q < 2 * NC; ++q) store_through(dst + 4 * q, *reinterpret_cast<const float4*>(src + 4 * q));
			} else {
#pragma unroll
				for (int z = 0; z < 8; ++z)
					if (bits >> z & 1) {
#pragma unroll
						for (int c = 0; c < NC; ++c) store_through(dst + z * NC + c, src[z * NC + c]);
					}
			}
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// mirror pressure loop: a rank without boundary waves still tells its peers that its sweep is complete; and, after the last
// sweep of a solve, a rank waits for its peers' before the gradient kernel reads the ghost voxels they wrote
__global__ void k_sweep_signal(const PhaseMirror m) {
	if ((int)threadIdx.x < m.n_peers) flag_store(m.peer_flag[threadIdx.x], m.seq);
}
__global__ void k_sweep_wait(const PhaseMirror m) {  // (raises this rank's flag for m.seq first: its sweeps up to m.seq have ended)
	if ((int)threadIdx.x < m.n_peers) {
		flag_store(m.peer_flag[threadIdx.x], m.seq);
		flag_wait(m.my_flags + kFlagSweep + m.peer_rank[threadIdx.x], m.seq, m.status);
	}
}

}  // namespace hns

using namespace hns;

// ---------------------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------------------

namespace {

enum { X_ADV = 0, X_D1 = 1, X_DIV = 2, X_P = 3, X_COUNT = 4 };  // halo region types (see the file header)

struct Mask512 {
	unsigned char row[64];
	void clear() { memset(row, 0, sizeof(row)); }
	void fill() { memset(row, 0xFF, sizeof(row)); }
	void operator|=(const Mask512& o) {
		for (int i = 0; i < 64; ++i) row[i] |= o.row[i];
	}
	int count() const {
		int c = 0;
		for (int i = 0; i < 64; ++i) c += __builtin_popcount(row[i]);
		return c;
	}
};

// voxels of a leaf within L1 distance D of the neighbouring leaf in direction j = (dx+1)*9 + (dy+1)*3 + (dz+1)
Mask512 reach_mask(int j, int D) {
	const int d[3] = {j / 9 - 1, (j / 3) % 3 - 1, j % 3 - 1};
	Mask512 m;
	m.clear();
	for (int x = 0; x < 8; ++x)
		for (int y = 0; y < 8; ++y)
			for (int z = 0; z < 8; ++z) {
				const int v[3] = {x, y, z};
				int dist = 0;
				for (int a = 0; a < 3; ++a) dist += d[a] < 0 ? v[a] + 1 : (d[a] > 0 ? 8 - v[a] : 0);
				if (dist <= D) m.row[x * 8 + y] |= (unsigned char)(1u << z);
			}
	return m;
}

struct Region {  // the voxels of some local leaves that travel in one exchange with one peer
	std::vector<int> leaf;            // local leaf ids
	std::vector<unsigned char> mask;  // 64 bytes per listed leaf
	std::vector<int> off;             // voxel offset of each listed leaf in the message
	int voxels = 0;
	bool whole = false;  // every listed leaf travels whole: plain 16-byte copies instead of the masked kernel
	int direct = -1;     // whole AND the listed leaves are consecutive local leaves starting here: the message IS that slice of the field
	int* d_leaf = nullptr;
	unsigned char* d_mask = nullptr;
	int* d_off = nullptr;
};

struct Peer {
	int rank = -1;
	Region send[X_COUNT], recv[X_COUNT];
	float* sbuf[2] = {nullptr, nullptr};  // message buffers, alternating with every exchange
	float* rbuf[2] = {nullptr, nullptr};
	size_t sbuf_floats = 0, rbuf_floats = 0;
};

struct Pending {  // an exchange that has been posted and not yet consumed
	bool active = false;
	bool prepacked = false;  // the boundary kernel wrote the messages itself (PackMirror): no pack launch
	int type = 0, parity = 0;
	hipStream_t stream = nullptr;  // where its boundary kernel, packing, transfer and unpacking run
	std::vector<std::pair<float*, int>> fields;  // (device field, ncomp) in message order
};

}  // namespace

struct hns_dist {
	int world = 1, rank = 0, k = 4, n_scalars = 1;
	float voxel_size = 1.0f;
	int64_t n_global = 0;
	int nB = 0, nI = 0, nG = 0;
	std::vector<int64_t> local_global;  // global id (position in the caller's leaf list) of every local leaf, local order [B | I | G]
	std::vector<int> owned_perm;        // position of local leaf l < nB+nI in the list of owned leaves in PARTITION order (owned_global)
	std::vector<int64_t> owned_global;  // global ids of the owned leaves in partition order: the order of the host arrays of upload / download
	bool blocked = false;               // the chained pressure loop of this decomposition takes two iterations per launch (blocked_mirror_rule, decided at create)
	int part_axis = -1;                 // -1: the partition is contiguous ranges of the caller's leaf order; 0 / 1 / 2: slabs along that axis (partition_order)
	std::vector<Peer> peers;
	hns_grid *gB = nullptr, *gI = nullptr, *gO = nullptr, *gA = nullptr;
	// device state over the local leaves
	void* arena = nullptr;
	size_t arena_bytes = 0;
	int device = -1;
	float *u = nullptr, *adv = nullptr, *tmp = nullptr, *div = nullptr, *p_a = nullptr, *p_b = nullptr, *p_result = nullptr, *stage = nullptr;
	std::vector<float*> phi, phi_next;
	void* tables = nullptr;  // region tables of every peer (one allocation)
	// the same regions, all peers concatenated (one pack / unpack launch per field when a rank has several peers)
	struct AllPeers {
		int n = 0;
		int* d_leaf = nullptr;
		unsigned char* d_mask = nullptr;
		int* d_off = nullptr;
		int* d_peer = nullptr;
	} all_send[4], all_recv[4];
	int* d_perm = nullptr;
	// streams and events
	hipStream_t cs = nullptr;
	std::shared_ptr<void> cs_owner;  // keeps `cs` alive: locally connected ranks all use ONE communication stream (see connect_local)
	hipEvent_t ev_post[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_bdone[2] = {nullptr, nullptr}, ev_ready = nullptr;  // (ev_bdone: the boundary kernel of an exchange has run)
	int parity = 0;
	Pending pending;
	bool phi_in_flight = false;  // the exchange of phi (and u) that opens the next substep has already been posted
	bool u_ghosts_fresh = false;
	// transport
	ncclComm_t comm = nullptr;
	std::vector<hns_dist*> local_ranks;  // "local" transport: every rank of the decomposition, in this process
	bool single_stream = false;          // local transport: no communication stream, everything in host order on the caller's stream
	bool loopback = false;               // timing-only transport: every message is answered out of this rank's own send buffer
	// "ipc" transport: what every peer mapped of its memory into this process, and this rank's own flags
	struct IpcPeer {
		char *arena = nullptr, *tables = nullptr;  // the peer's field memory and its table / message-buffer allocation, mapped here
		uint32_t* flags = nullptr;
		void* opened[3] = {nullptr, nullptr, nullptr};
		uint64_t unit_bytes = 0, rbuf_off[2] = {0, 0};
		int recv_direct[4] = {-1, -1, -1, -1}, recv_voxels[4] = {0, 0, 0, 0};
		uint64_t recv_leaf_off[4] = {0, 0, 0, 0};  // where (in its tables allocation) the peer keeps the local indices of its ghost leaves of each region type
		uint32_t recv_leaves[4] = {0, 0, 0, 0};
	};
	std::vector<IpcPeer> ipc_peers;  // parallel to `peers`
	uint32_t* ipc_flags = nullptr;   // fine-grained device memory, written by the peers
	int* ipc_status = nullptr;       // host-mapped: non-zero once a wait on a peer ran out
	int* far_status = nullptr;       // host-mapped: raised by an advection kernel whose back-trace left the one-leaf ghost layer (GridDev::far_flag)
	uint32_t ipc_seq = 0;
	bool ipc = false;
	size_t unit_bytes = 0;  // bytes per scalar field over the local leaves (fields sit at multiples of it in the arena)
	// "mirror" pressure loop (sweeps_per_exchange = 1 over the ipc or local transport): the sweep kernel itself writes its
	// boundary rows into the peers' ghost voxels (hns_pressure.hip: k_rbgs_pair_mirror)
	bool mirror = false;
	void* mir_tables = nullptr;
	PhaseMirror mir;            // everything but the region tables, the output arrays and the launch number
	struct MirTables {          // per halo region type (X_ADV, X_D1, X_DIV, X_P)
		const int* first = nullptr;
		const int2* entry = nullptr;
		const unsigned char* mask = nullptr;  // null: whole leaves
	} mir_type[4];
	// the blocked boundary sweep of the exchanged pressure loop packs its own messages (hns_flags.hpp: PackMirror): tables per region type (X_D1, X_P), one allocation
	void* pack_tables = nullptr;
	PackMirror pack_type[4];
	bool pack_ok[4] = {false, false, false, false};
	bool chain = false;         // ... and every other kernel of the substep delivers its own halo too (no communication stream at all)
	uint32_t sweep_seq = 0;
	// statistics of the last substep
	uint64_t bytes_sent[X_COUNT] = {0, 0, 0, 0}, messages_sent = 0, exchanges = 0, packed_exchanges = 0;
	// hipEvent bracketing of the pressure loop (communication included)
	bool timing = false;
	std::vector<hipEvent_t> tev;
	size_t tev_used = 0;
	long long timed_sweeps = 0;
};

static int far_check(const hns_dist* d) {
	if (d->far_status && *(volatile int*)d->far_status)
		return fail(HNS_ERR_RUNTIME, "hns_dist: an advection back-trace reached beyond the one-leaf ghost layer of this rank (|u| dt / dx above ~8 voxels at a partition "
		                             "boundary): the owned result can differ from the single-domain run. Use a smaller time step (or fewer ranks); upload the fields again to clear this.");
	return HNS_OK;
}

namespace {

// Round 5: WHICH leaves a rank owns. Rounds 1-4 cut the caller's leaf list (NanoVDB order: hierarchical, x-major) into `world` contiguous ranges.
// That is a slab decomposition for box domains, but on BASELINE config 5 (the 66k-leaf plume on a 1024^3 extent, 8 ranks) the ranges are
// chunks of 128^3-voxel NanoVDB nodes and every rank touches SEVEN others: 14 point-to-point messages per exchange, 25+ exchanges per substep.
// Here the leaves are put in slab order -- by leaf coordinate along ONE axis, the caller's order inside a plane of leaves --, THAT list is
// cut into `world` equal ranges, and every range is put back into the caller's order: a rank touches the rank before and the rank behind it (plus the owner of the caller's leaf 0, whose element
// 0 every rank mirrors). The axis is the one whose cuts cross the fewest leaves (the plume: y, its own axis -- 1,398 boundary leaves per rank on average
// instead of 1,914, 1 - 2 halo peers instead of 4 - 7). If the cut along x selects the same leaf sets as the contiguous ranges did (every box domain
// whose slabs are whole 128-voxel NanoVDB nodes: the weak-scaling slabs of bench.py) AND costs no more than 1.1 x the cheapest cut, the caller's
// order is kept as it is: part_axis -1.
// order[i] = position in the caller's list of the i-th leaf in partition order. Every rank derives the same order from the same global list.
int partition_order(const int32_t* origins, int64_t n, int world, bool leaf_order, std::vector<int64_t>& order) {
	order.resize((size_t)n);
	for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
	if (leaf_order || world <= 1 || n == 0) return -1;
	std::vector<int64_t> best, along_x;
	int64_t best_cost = -1, cost_x = -1;
	int best_axis = -1;
	for (int a = 0; a < 3; ++a) {
		std::vector<int64_t> o = order;
		std::stable_sort(o.begin(), o.end(), [&](int64_t x, int64_t y) { return origins[(size_t)x * 3 + (size_t)a] < origins[(size_t)y * 3 + (size_t)a]; });
		// leaves in the planes a cut touches: what the two ranks at that cut exchange
		int64_t cost = 0;
		for (int r = 1; r < world; ++r) {
			const int64_t c = n * r / world;
			if (c <= 0 || c >= n) continue;
			const int32_t lo = origins[(size_t)o[(size_t)c - 1] * 3 + (size_t)a], hi = origins[(size_t)o[(size_t)c] * 3 + (size_t)a];
			auto plane = [&](int32_t v) {
				auto cmp_lo = [&](int64_t x, int32_t val) { return origins[(size_t)x * 3 + (size_t)a] < val; };
				auto cmp_hi = [&](int32_t val, int64_t x) { return val < origins[(size_t)x * 3 + (size_t)a]; };
				return (int64_t)(std::upper_bound(o.begin(), o.end(), v, cmp_hi) - std::lower_bound(o.begin(), o.end(), v, cmp_lo));
			};
			cost += plane(lo) + (hi != lo ? plane(hi) : 0);
		}
		if (a == 0) cost_x = cost, along_x = o;
		if (best_cost < 0 || cost < best_cost) best_cost = cost, best_axis = a, best.swap(o);
	}
	// Does the x cut select the same leaf sets as the contiguous ranges of the caller's order -- and is it (nearly) the cheapest cut? Then that order stays as it
	// is: a rank's memory layout is the caller's, every whole-leaf region a slice of it (rounds 1-4's partition; bench.py's weak-scaling slabs)
	bool same = cost_x >= 0 && cost_x * 10 <= best_cost * 11;
	{
		const std::vector<int64_t>& o = best_axis == 0 ? best : along_x;
		for (int r = 0; r < world && same; ++r)
			for (int64_t i = n * r / world; i < n * (r + 1) / world && same; ++i) same = o[(size_t)i] >= n * r / world && o[(size_t)i] < n * (r + 1) / world;
	}
	if (same) return -1;
	// the slabs decide WHICH leaves a rank owns; inside a rank they stay in the caller's order (a rank's memory layout then is NanoVDB order restricted to
	// its slab -- whole 128^3-voxel nodes, compact in 3-D -- where plane-major order put a leaf's neighbours along the cut axis a whole plane of leaves
	// away: measured on config 5, rank 4 of 8 alone, 13.2 against 12.3 us per iteration of the chained pressure loop)
	for (int r = 0; r < world; ++r) std::sort(best.begin() + n * r / world, best.begin() + n * (r + 1) / world);
	order.swap(best);
	return best_axis;
}

int build_plan(hns_dist* d, const int32_t* origins, int64_t n, int world, int rank, int64_t g0) {
	Topology topo;
	HNS_TRY(topo.prepare(origins, n));
	HNS_TRY(topo.build_tables());
	std::vector<int64_t> bounds((size_t)world + 1);
	for (int r = 0; r <= world; ++r) bounds[(size_t)r] = n * r / world;
	auto owner = [&](int64_t l) { return (int)(std::upper_bound(bounds.begin(), bounds.end(), l) - bounds.begin()) - 1; };
	const int64_t o0 = bounds[(size_t)rank], o1 = bounds[(size_t)rank + 1];
	const int n_owned = (int)(o1 - o0);
	const int owner0 = n ? owner(g0) : rank;  // (g0: where the caller's leaf 0 -- whose element 0 advect_scalars reads for out-of-domain taps -- sits in partition order)

	Mask512 reach[27][X_COUNT];
	const int depth[X_COUNT] = {24, 1, 2 * d->k - 1, 2 * d->k};  // X_ADV: the whole leaf
	for (int j = 0; j < 27; ++j)
		for (int t = 0; t < X_COUNT; ++t) reach[j][t] = reach_mask(j, depth[t]);

	// one pass over the neighbour rows of the owned leaves gives both directions: owned leaf l with a neighbour nb owned by
	// q is a ghost of q (reach from l towards nb), and nb is a ghost of mine (reach from nb towards l = opposite direction)
	struct Entry {
		Mask512 m[X_COUNT];
	};
	std::vector<std::vector<std::pair<int64_t, Entry>>> send_of((size_t)world), recv_of((size_t)world);  // per peer, sorted by global leaf id
	auto entry = [&](std::vector<std::pair<int64_t, Entry>>& v, std::vector<int64_t>& keys, int64_t id) -> Entry& {
		// `keys` mirrors the ids of `v`, both ascending: binary search, insert when new (the visiting order is nearly ascending)
		auto it = std::lower_bound(keys.begin(), keys.end(), id);
		const size_t pos = (size_t)(it - keys.begin());
		if (it == keys.end() || *it != id) {
			keys.insert(it, id);
			Entry z;
			for (int t = 0; t < X_COUNT; ++t) z.m[t].clear();
			v.insert(v.begin() + (long)pos, std::make_pair(id, z));
		}
		return v[pos].second;
	};
	std::vector<std::vector<int64_t>> send_keys((size_t)world), recv_keys((size_t)world);
	std::vector<char> is_boundary((size_t)n_owned, 0);
	for (int64_t l = o0; l < o1; ++l) {
		for (int j = 0; j < 27; ++j) {
			if (j == 13) continue;
			const int64_t nb = topo.nbr27[(size_t)l * 27 + (size_t)j];
			if (nb < 0 || (nb >= o0 && nb < o1)) continue;
			const int q = owner(nb);
			Entry& s = entry(send_of[(size_t)q], send_keys[(size_t)q], l);
			Entry& r = entry(recv_of[(size_t)q], recv_keys[(size_t)q], nb);
			for (int t = 0; t < X_COUNT; ++t) {
				s.m[t] |= reach[j][t];
				r.m[t] |= reach[26 - j][t];
			}
			is_boundary[(size_t)(l - o0)] = 1;
		}
	}
	// the mirror of global element 0 (voxel 0 of global leaf 0), advection inputs only; the whole leaf travels (8 KB per
	// exchange) so that the region stays one of whole leaves, which can be sent without packing
	if (world > 1 && n > 0) {
		if (owner0 == rank) {
			for (int q = 0; q < world; ++q)
				if (q != rank) entry(send_of[(size_t)q], send_keys[(size_t)q], g0).m[X_ADV].fill();
			is_boundary[(size_t)(g0 - o0)] = 1;
		} else {
			entry(recv_of[(size_t)owner0], recv_keys[(size_t)owner0], g0).m[X_ADV].fill();
		}
	}

	// Round 5: the ORDER of boundary leaves and of ghosts. A whole-leaf region that is a run of consecutive local leaves travels straight out of / into the
	// field (Region::direct: no pack / unpack launch). In ascending leaf order that holds for x-slabs of a box, but in a slab along another axis the leaves next
	// to the rank before and those next to the rank behind alternate through the list. So a leaf that other ranks hold copies of is ordered by WHO holds them
	// -- the sorted list of those ranks, compared lexicographically, then by leaf number: with peers L < U the boundary reads [only L | L and U | only U] and
	// both send regions are runs. Owner and ghost holder must enumerate a region alike: the holder sorts its ghosts by the same key, which it derives from the
	// global leaf list like everything else in the plan.
	std::unordered_map<int64_t, std::vector<int>> key_cache;
	auto holders = [&](int64_t id) -> const std::vector<int>& {  // the other ranks that hold a copy of leaf `id` (partition-order number), ascending
		auto it = key_cache.find(id);
		if (it != key_cache.end()) return it->second;
		std::vector<int> k;
		const int own = owner(id);
		if (id == g0 && world > 1) {
			for (int q = 0; q < world; ++q)
				if (q != own) k.push_back(q);
		} else {
			for (int j = 0; j < 27; ++j) {
				const int64_t nb = topo.nbr27[(size_t)id * 27 + (size_t)j];
				if (nb >= 0 && owner(nb) != own) k.push_back(owner(nb));
			}
			std::sort(k.begin(), k.end());
			k.erase(std::unique(k.begin(), k.end()), k.end());
		}
		return key_cache.emplace(id, std::move(k)).first->second;
	};
	auto before = [&](int64_t a, int64_t b) {
		const std::vector<int>&ka = holders(a), &kb = holders(b);
		if (ka != kb) return std::lexicographical_compare(ka.begin(), ka.end(), kb.begin(), kb.end());
		return a < b;
	};
	for (int q = 0; q < world; ++q) {
		std::stable_sort(send_of[(size_t)q].begin(), send_of[(size_t)q].end(), [&](const std::pair<int64_t, Entry>& a, const std::pair<int64_t, Entry>& b) { return before(a.first, b.first); });
		std::stable_sort(recv_of[(size_t)q].begin(), recv_of[(size_t)q].end(), [&](const std::pair<int64_t, Entry>& a, const std::pair<int64_t, Entry>& b) { return before(a.first, b.first); });
	}
	// local order [B | I | G]
	d->local_global.clear();
	std::vector<int> local_of_owned((size_t)n_owned, -1);
	{
		std::vector<int64_t> bl;
		for (int i = 0; i < n_owned; ++i)
			if (is_boundary[(size_t)i]) bl.push_back(o0 + i);
		std::sort(bl.begin(), bl.end(), before);
		for (int64_t id : bl) {
			local_of_owned[(size_t)(id - o0)] = (int)d->local_global.size();
			d->local_global.push_back(id);
		}
		for (int i = 0; i < n_owned; ++i)
			if (!is_boundary[(size_t)i]) {
				local_of_owned[(size_t)i] = (int)d->local_global.size();
				d->local_global.push_back(o0 + i);
			}
	}
	d->nB = 0;
	for (char b : is_boundary) d->nB += b ? 1 : 0;
	d->nI = n_owned - d->nB;
	d->owned_perm.resize((size_t)n_owned);
	for (int i = 0; i < n_owned; ++i) d->owned_perm[(size_t)local_of_owned[(size_t)i]] = i;
	d->peers.clear();
	for (int q = 0; q < world; ++q) {
		if (q == rank || (send_of[(size_t)q].empty() && recv_of[(size_t)q].empty())) continue;
		Peer p;
		p.rank = q;
		const int ghost_base = (int)d->local_global.size();
		for (auto& e : recv_of[(size_t)q]) d->local_global.push_back(e.first);
		for (int t = 0; t < X_COUNT; ++t) {
			int k = 0;
			for (auto& e : recv_of[(size_t)q]) {
				const int c = e.second.m[t].count();
				if (c) {
					p.recv[t].leaf.push_back(ghost_base + k);
					p.recv[t].mask.insert(p.recv[t].mask.end(), e.second.m[t].row, e.second.m[t].row + 64);
					p.recv[t].off.push_back(p.recv[t].voxels);
					p.recv[t].voxels += c;
				}
				++k;
			}
			for (auto& e : send_of[(size_t)q]) {
				const int c = e.second.m[t].count();
				if (c) {
					p.send[t].leaf.push_back(local_of_owned[(size_t)(e.first - o0)]);
					p.send[t].mask.insert(p.send[t].mask.end(), e.second.m[t].row, e.second.m[t].row + 64);
					p.send[t].off.push_back(p.send[t].voxels);
					p.send[t].voxels += c;
				}
			}
		}
		for (int t = 0; t < X_COUNT; ++t)
			for (Region* r : {&p.send[t], &p.recv[t]}) {
				r->whole = r->voxels == 512 * (int)r->leaf.size();
				bool run = r->whole && !r->leaf.empty();
				for (size_t i = 1; run && i < r->leaf.size(); ++i) run = r->leaf[i] == r->leaf[0] + (int)i;
				r->direct = run ? r->leaf[0] : -1;
			}
		d->peers.push_back(std::move(p));
	}
	d->nG = (int)d->local_global.size() - n_owned;
	return HNS_OK;
}

size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// ---- flags and the mirror pressure loop (ipc and local transports) ----
namespace {

int ensure_flags(hns_dist* d) {
	if (d->ipc_flags) return HNS_OK;
	// fine-grained: written by kernels of other processes / devices while kernels here poll it
	if (hipExtMallocWithFlags((void**)&d->ipc_flags, sizeof(uint32_t) * kFlagWords, hipDeviceMallocFinegrained) != hipSuccess) {
		(void)hipGetLastError();
		return fail(HNS_ERR_HIP, "hns_dist: allocating the flag page failed");
	}
	HNS_HIP(hipMemset(d->ipc_flags, 0, sizeof(uint32_t) * kFlagWords));
	HNS_HIP(hipHostMalloc((void**)&d->ipc_status, 64, hipHostMallocMapped));
	*d->ipc_status = 0;
	HNS_HIP(hipDeviceSynchronize());
	return HNS_OK;
}

// remote_leaf[t][i][j]: peer i's local index of its ghost copy of the j-th leaf of this rank's send region of type t;
// peer_arena / peer_unit / peer_flags: that peer's field memory, its bytes per scalar field and its flag page as addressable here
int setup_mirror(hns_dist* d, const std::vector<std::vector<int>> (&remote_leaf)[X_COUNT], const std::vector<char*>& peer_arena, const std::vector<uint64_t>& peer_unit,
                 const std::vector<uint32_t*>& peer_flags) {
	// (stay on the exchanged substep) -- decided from what EVERY rank knows: the owner of global leaf 0 has all other ranks as peers
	// (the element-0 mirror region), so some rank exceeds the peer table exactly when world - 1 does, and then no rank mirrors
	if (d->world - 1 > kMirrorMaxPeers || d->world > kFlagSlots) return HNS_OK;
	HNS_TRY(ensure_flags(d));
	const int nB = d->nB;
	std::vector<int> first[X_COUNT];
	std::vector<int2> entry[X_COUNT];
	std::vector<unsigned char> mask[X_COUNT];
	size_t bytes = 256;
	for (int t = 0; t < X_COUNT; ++t) {
		std::vector<std::vector<std::pair<int2, const unsigned char*>>> per_leaf((size_t)nB);
		bool whole = true;
		for (size_t i = 0; i < d->peers.size(); ++i) {
			const Region& r = d->peers[i].send[t];
			if (remote_leaf[t][i].size() != r.leaf.size()) return fail(HNS_ERR_RUNTIME, "hns_dist: send/receive plans of two ranks disagree");
			whole = whole && (r.whole || r.leaf.empty());
			for (size_t j = 0; j < r.leaf.size(); ++j) {
				if (r.leaf[j] < 0 || r.leaf[j] >= nB) return fail(HNS_ERR_RUNTIME, "hns_dist: a mirrored leaf is not a boundary leaf");
				per_leaf[(size_t)r.leaf[j]].push_back({int2{(int)i, remote_leaf[t][i][j]}, r.mask.data() + j * 64});
			}
		}
		first[t].assign((size_t)nB + 1, 0);
		for (int l = 0; l < nB; ++l) {
			first[t][(size_t)l] = (int)entry[t].size();
			for (auto& e : per_leaf[(size_t)l]) {
				entry[t].push_back(e.first);
				if (!whole) mask[t].insert(mask[t].end(), e.second, e.second + 64);
			}
		}
		first[t][(size_t)nB] = (int)entry[t].size();
		bytes += pad256(sizeof(int) * first[t].size()) + pad256(sizeof(int2) * entry[t].size()) + pad256(mask[t].size());
	}
	if (hipMalloc(&d->mir_tables, bytes) != hipSuccess) return fail(HNS_ERR_HIP, "hns_dist: allocating the mirror tables failed");
	char* q = (char*)d->mir_tables;
	auto put = [&](const void* src, size_t n) -> char* {
		char* r = q;
		if (n && hipMemcpy(q, src, n, hipMemcpyHostToDevice) != hipSuccess) r = nullptr;
		q += pad256(n);
		return r;
	};
	for (int t = 0; t < X_COUNT; ++t) {
		d->mir_type[t].first = (const int*)put(first[t].data(), sizeof(int) * first[t].size());
		d->mir_type[t].entry = (const int2*)put(entry[t].data(), sizeof(int2) * entry[t].size());
		d->mir_type[t].mask = mask[t].empty() ? nullptr : (const unsigned char*)put(mask[t].data(), mask[t].size());
		if (!d->mir_type[t].first || !d->mir_type[t].entry) return fail(HNS_ERR_HIP, "hns_dist: uploading the mirror tables failed");
	}
	PhaseMirror& m = d->mir;
	memset(&m, 0, sizeof(m));
	m.n_boundary = nB, m.n_peers = (int)d->peers.size();
	for (size_t i = 0; i < d->peers.size(); ++i) {
		m.peer_arena[i] = peer_arena[i], m.peer_unit[i] = peer_unit[i];
		m.peer_flag[i] = peer_flags[i] + kFlagSweep + d->rank;
		m.peer_rank[i] = d->peers[i].rank;
	}
	m.my_flags = d->ipc_flags, m.status = d->ipc_status;
	d->mirror = true;
	// every kernel of the substep in one launch each (32-bit addressed advection kernels: fields below 4 GiB)
	// (decided from what every rank knows alike: all ranks must take the same path)
	d->chain = (uint64_t)d->n_global * 6144u <= 0xFFFF0000ull;
	return HNS_OK;
}

// the arguments of one chained launch: region type `t`, output arrays `outs` (device fields of this rank, components per voxel)
PhaseMirror phase_args(hns_dist* d, int t, const std::vector<std::pair<const float*, int>>& outs) {
	PhaseMirror m = d->mir;
	m.first = d->mir_type[t].first, m.entry = d->mir_type[t].entry, m.mask = d->mir_type[t].mask;
	int k = 0, comps = 0;
	for (auto& f : outs) m.out_unit[k++] = (int)((size_t)((const char*)f.first - (const char*)d->arena) / d->unit_bytes), comps += f.second;
	m.seq = ++d->sweep_seq;
	for (Peer& p : d->peers) d->bytes_sent[t] += sizeof(float) * (size_t)p.send[t].voxels * (size_t)comps;
	return m;
}

// Round 4: sweeps_per_exchange = 2 mirrors as well -- its sweeps are the temporally blocked form, two iterations per chained launch
// (hns_sorblock.hip: k_rbgs_block<2, 2, ., true, PhaseMirror>), whose mirror region is the plan's reach-4 region of p. All ranks must
// take the same path (they count launches alike), so the decision uses only what every rank knows: the smallest owned range must be
// swept in 16^3 blocks (more than 600 leaves, hns_rbgs_block_shape) and the option must say so.
// Does a rank of `world` ranks over `n_global` leaves take the chained blocked sweep (two iterations per chained launch) when it runs with
// sweeps_per_exchange = `k` over the ipc / local transports? From what every rank knows alike (all ranks must count launches alike) and the options
// as they are NOW: hns_dist_create asks once and stores the answer (hns_dist::blocked), which is what create, connect and the substep use -- an option
// changed between create and connect no longer leaves a rank on two paths at once.
bool blocked_mirror_rule(int k, int world, int64_t n_global) {
	return k == 2 && options().sor_block_lb.load() != 1 && world > 0 && n_global / world > 600 && n_global <= 2000000;
}
bool blocked_mirror(const hns_dist* d) { return d->blocked; }
// (rounds 2-5 also chained ranks of one-leaf blocks, sweeps_per_exchange = 1, through a mirroring one-iteration kernel; such ranks -- 600 leaves and fewer -- take the exchanged path now)
bool mirror_wanted(const hns_dist* d) { return blocked_mirror(d) && d->world > 1 && options().dist_mirror.load() != 0; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// create / destroy
// ---------------------------------------------------------------------------------------------------------------

extern "C" {

void hns_dist_destroy(hns_dist* d) {
	if (!d) return;
	if (d->cs) (void)hipStreamSynchronize(d->cs);
	if (d->comm) (void)rccl().CommDestroy(d->comm);
	for (hns_dist::IpcPeer& q : d->ipc_peers)
		for (void* o : q.opened)
			if (o) (void)hipIpcCloseMemHandle(o);
	if (d->mir_tables) (void)hipFree(d->mir_tables);
	if (d->ipc_flags) (void)hipFree(d->ipc_flags);
	if (d->ipc_status) (void)hipHostFree(d->ipc_status);
	if (d->far_status) (void)hipHostFree(d->far_status);
	for (hipEvent_t e : d->tev) (void)hipEventDestroy(e);
	if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
	for (int i = 0; i < 2; ++i) {
		if (d->ev_post[i]) (void)hipEventDestroy(d->ev_post[i]);
		if (d->ev_done[i]) (void)hipEventDestroy(d->ev_done[i]);
		if (d->ev_bdone[i]) (void)hipEventDestroy(d->ev_bdone[i]);
	}
	d->cs_owner.reset();  // destroys the stream with its last user
	for (hns_grid* g : {d->gB, d->gI, d->gO, d->gA})
		if (g) hns_grid_destroy(g);
	if (d->arena) hns_arena_put(d->arena, d->arena_bytes, d->device);
	if (d->tables) (void)hipFree(d->tables);
	if (d->pack_tables) (void)hipFree(d->pack_tables);
	delete d;
}

// PackMirror tables of the region types the exchanged pressure loop sends (X_P between blocks of sweeps, X_D1 behind the last): per boundary leaf the peers that read it, where its
// first travelling voxel stands in that peer's message, its 64-byte mask and the count of travelling voxels in front of each row. A peer whose region travels straight out of the
// field (whole consecutive leaves) has no entries: nothing is packed for it either way.
static int build_pack_tables(hns_dist* d) {
	if (d->world < 2 || d->peers.empty() || d->peers.size() > (size_t)kMirrorMaxPeers || d->nB == 0) return HNS_OK;
	struct Host {
		std::vector<int> first;
		std::vector<int2> entry;
		std::vector<unsigned char> mask;
		std::vector<unsigned short> pre;
	} h[4];
	size_t bytes = 0;
	for (int t : {X_D1, X_P}) {
		std::vector<std::vector<std::pair<int, int>>> of((size_t)d->nB);  // per boundary leaf: (peer index, index in that peer's region)
		bool fits = true;
		for (size_t pi = 0; pi < d->peers.size(); ++pi) {
			const Region& r = d->peers[pi].send[t];
			if (r.direct >= 0) continue;
			for (size_t i = 0; i < r.leaf.size(); ++i) {
				if (r.leaf[i] < 0 || r.leaf[i] >= d->nB) fits = false;
				else of[(size_t)r.leaf[i]].emplace_back((int)pi, (int)i);
			}
		}
		if (!fits) continue;
		Host& o = h[t];
		o.first.assign((size_t)d->nB + 1, 0);
		for (int l = 0; l < d->nB; ++l) {
			o.first[(size_t)l] = (int)o.entry.size();
			for (auto& e : of[(size_t)l]) {
				const Region& r = d->peers[(size_t)e.first].send[t];
				o.entry.push_back(make_int2(e.first, r.off[(size_t)e.second]));
				const unsigned char* m = r.mask.data() + (size_t)e.second * 64;
				o.mask.insert(o.mask.end(), m, m + 64);
				unsigned short run = 0;
				for (int row = 0; row < 64; ++row) {
					o.pre.push_back(run);
					run = (unsigned short)(run + __builtin_popcount(m[row]));
				}
			}
		}
		o.first[(size_t)d->nB] = (int)o.entry.size();
		bytes += pad256(sizeof(int) * o.first.size()) + pad256(sizeof(int2) * o.entry.size()) + pad256(o.mask.size()) + pad256(sizeof(unsigned short) * o.pre.size());
		d->pack_ok[t] = true;
	}
	if (!bytes) return HNS_OK;
	if (hipMalloc(&d->pack_tables, bytes) != hipSuccess) return fail(HNS_ERR_HIP, "hns_dist_create: allocating the pack tables failed");
	char* q = (char*)d->pack_tables;
	int rc = HNS_OK;
	auto put = [&](const void* src, size_t n) -> void* {
		void* r = q;
		if (n && hipMemcpy(q, src, n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(HNS_ERR_HIP, "hns_dist_create: uploading the pack tables failed");
		q += pad256(n);
		return r;
	};
	for (int t : {X_D1, X_P}) {
		if (!d->pack_ok[t]) continue;
		PackMirror& m = d->pack_type[t];
		m.n_boundary = d->nB;
		m.first = (const int*)put(h[t].first.data(), sizeof(int) * h[t].first.size());
		m.entry = (const int2*)put(h[t].entry.data(), sizeof(int2) * h[t].entry.size());
		m.mask = (const unsigned char*)put(h[t].mask.data(), h[t].mask.size());
		m.row_pre = (const unsigned short*)put(h[t].pre.data(), sizeof(unsigned short) * h[t].pre.size());
		for (int i = 0; i < kMirrorMaxPeers; ++i) m.msg[i] = nullptr;
	}
	return rc;
}

hns_dist* hns_dist_create(const int32_t* global_leaf_origins_xyz, uint64_t n_leaves, int world, int rank, float voxel_size, int n_scalars,
                          int sweeps_per_exchange, unsigned flags, int* err) {
	int rc = HNS_OK;
	hns_dist* d = nullptr;
	auto bail = [&](int code) -> hns_dist* {
		if (d) hns_dist_destroy(d);
		if (err) *err = code;
		return nullptr;
	};
	if ((!global_leaf_origins_xyz && n_leaves) || world < 1 || rank < 0 || rank >= world || n_scalars < 0 || n_scalars > 8 || voxel_size <= 0.0f)
		return bail(fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_create: bad arguments"));
	if (sweeps_per_exchange == 0) sweeps_per_exchange = 4;
	if (sweeps_per_exchange < 1 || sweeps_per_exchange > 4)  // a ghost layer is one leaf = 8 voxels deep and a fused sweep consumes two
		return bail(fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_create: sweeps_per_exchange must be 1..4"));
	d = new hns_dist;
	d->world = world, d->rank = rank, d->k = sweeps_per_exchange, d->n_scalars = n_scalars, d->voxel_size = voxel_size, d->n_global = (int64_t)n_leaves;
	d->blocked = blocked_mirror_rule(d->k, world, (int64_t)n_leaves);
	{
		// the plan is built over the leaves in PARTITION order (build_plan's "global" ids are positions in that order); what leaves this block are
		// positions in the caller's list again
		std::vector<int64_t> order;
		d->part_axis = partition_order(global_leaf_origins_xyz, (int64_t)n_leaves, world, (flags & HNS_DIST_LEAF_ORDER) != 0, order);
		std::vector<int32_t> po;
		const int32_t* plan_origins = global_leaf_origins_xyz;
		int64_t g0 = 0;
		if (d->part_axis >= 0) {
			po.resize((size_t)n_leaves * 3);
			for (uint64_t i = 0; i < n_leaves; ++i) {
				for (int a = 0; a < 3; ++a) po[(size_t)i * 3 + (size_t)a] = global_leaf_origins_xyz[(size_t)order[(size_t)i] * 3 + (size_t)a];
				if (order[(size_t)i] == 0) g0 = (int64_t)i;
			}
			plan_origins = po.data();
		}
		if ((rc = build_plan(d, plan_origins, (int64_t)n_leaves, world, rank, g0)) != HNS_OK) return bail(rc);
		for (int64_t& id : d->local_global) id = order[(size_t)id];
		const int64_t o0 = (int64_t)n_leaves * rank / world, o1 = (int64_t)n_leaves * (rank + 1) / world;
		d->owned_global.assign(order.begin() + o0, order.begin() + o1);
	}
	if (flags & HNS_DIST_PLAN_ONLY) {  // host-side plan for inspection (tests run it through a CPU engine); no device is touched
		if (err) *err = HNS_OK;
		return d;
	}

	// the four launch ranges over the local leaves
	const int n_local = (int)d->local_global.size(), nO = d->nB + d->nI;
	std::vector<int32_t> lo((size_t)n_local * 3);
	for (int i = 0; i < n_local; ++i)
		for (int a = 0; a < 3; ++a) lo[(size_t)i * 3 + a] = global_leaf_origins_xyz[(size_t)d->local_global[(size_t)i] * 3 + a];
	hns_grid** gs[4] = {&d->gB, &d->gI, &d->gO, &d->gA};
	const uint64_t first[4] = {0, (uint64_t)d->nB, 0, 0}, count[4] = {(uint64_t)d->nB, (uint64_t)d->nI, (uint64_t)nO, (uint64_t)n_local};
	uint64_t outside = 0;
	for (int i = 0; i < n_local; ++i)
		if (d->local_global[(size_t)i] == 0) outside = (uint64_t)i * 512u;
	for (int i = 0; i < 4; ++i) {
		*gs[i] = hns_grid_create_from_leaves(lo.data(), (uint64_t)n_local, voxel_size, HNS_GRID_DEFAULT, &rc);
		if (!*gs[i]) return bail(rc);
		// the owned range deals the boundary leaves out to all eight XCDs first (the mirroring pressure loop sweeps this range: its
		// boundary waves poll, store twice and signal, and as the head of XCD 0's chunk they made that XCD the last to finish)
		if (i == 2 && (sweeps_per_exchange == 1 || blocked_mirror(d))) (*gs[i])->sched_prefix = (uint64_t)d->nB;
		// the chained blocked sweep (hns_sorblock.hip) must know which leaves of the owned range are boundary leaves whatever the launch order is
		if (i == 2 || i == 0) (*gs[i])->chain_boundary = (uint64_t)d->nB;  // (the boundary range too: its blocked sweep may pack the peers' messages, build_pack_tables)
		if ((rc = hns_grid_set_active_range(*gs[i], first[i], count[i])) != HNS_OK) return bail(rc);
		if ((rc = hns_grid_set_outside_element(*gs[i], outside)) != HNS_OK) return bail(rc);
	}
	d->device = d->gA->device;
	// A rank holds ONE layer of ghost leaves: a tap inside the 27-leaf neighbourhood of an owned leaf is always answered as the
	// single domain would; further away a leaf that is missing HERE may exist on another rank. The advection kernels raise this
	// word on such a tap and the next hns_dist call fails (the single-GPU path follows any back-trace through its origin hash).
	if (world > 1) {
		if (hipHostMalloc((void**)&d->far_status, 64, hipHostMallocMapped) != hipSuccess) return bail(fail(HNS_ERR_HIP, "hns_dist_create: hipHostMalloc failed"));
		*d->far_status = 0;
		for (int i = 0; i < 4; ++i) (*gs[i])->far_flag = d->far_status;
	}

	// device state: u, adv, tmp (Vec3f) | div, p_a, p_b | phi, phi_next per scalar | upload/download staging (Vec3f over the owned leaves)
	{
		const size_t unit = pad256(sizeof(float) * 512 * (size_t)std::max(n_local, 1));
		d->unit_bytes = unit;
		const size_t units = 3 + 3 + 3 + 3 + 2 * (size_t)n_scalars + 3;
		if ((rc = hns_arena_get(unit * units, d->device, &d->arena, &d->arena_bytes)) != HNS_OK) return bail(rc);
		if (hipMemset(d->arena, 0, unit * units) != hipSuccess) return bail(fail(HNS_ERR_HIP, "hns_dist_create: clearing the field memory failed"));
		char* q = (char*)d->arena;
		auto take = [&](size_t k) {
			float* r = (float*)q;
			q += k * unit;
			return r;
		};
		d->u = take(3), d->adv = take(3), d->tmp = take(3), d->div = take(1), d->p_a = take(1), d->p_b = take(1);
		for (int s = 0; s < n_scalars; ++s) d->phi.push_back(take(1)), d->phi_next.push_back(take(1));
		d->stage = take(3);
		d->p_result = d->p_a;
	}
	// region tables and message buffers
	{
		size_t bytes = pad256(sizeof(int) * (size_t)std::max(nO, 1));
		for (Peer& p : d->peers) {
			for (int t = 0; t < X_COUNT; ++t)
				for (Region* r : {&p.send[t], &p.recv[t]}) bytes += pad256(sizeof(int) * r->leaf.size()) + pad256(r->mask.size()) + pad256(sizeof(int) * r->off.size());
			p.sbuf_floats = (size_t)p.send[X_ADV].voxels * (size_t)(3 + n_scalars);
			p.rbuf_floats = (size_t)p.recv[X_ADV].voxels * (size_t)(3 + n_scalars);
			for (int t = 1; t < X_COUNT; ++t) {  // every other message is one Vec3f or one float per voxel of a smaller region
				p.sbuf_floats = std::max(p.sbuf_floats, (size_t)p.send[t].voxels * 3);
				p.rbuf_floats = std::max(p.rbuf_floats, (size_t)p.recv[t].voxels * 3);
			}
			bytes += 2 * pad256(sizeof(float) * p.sbuf_floats) + 2 * pad256(sizeof(float) * p.rbuf_floats);
		}
		const bool batch = d->peers.size() > 1 && d->peers.size() <= (size_t)kMaxBatchPeers;
		if (batch)
			for (int t = 0; t < X_COUNT; ++t)
				for (int dir = 0; dir < 2; ++dir) {
					size_t n = 0;
					for (Peer& p : d->peers) n += (dir ? p.recv[t] : p.send[t]).leaf.size();  // (direct regions are left out below)
					bytes += 3 * pad256(sizeof(int) * n) + pad256(64 * n);
				}
		if (hipMalloc(&d->tables, bytes) != hipSuccess) return bail(fail(HNS_ERR_HIP, "hns_dist_create: allocating the halo tables failed"));
		char* q = (char*)d->tables;
		auto put = [&](const void* src, size_t n) -> void* {
			void* r = q;
			if (n && hipMemcpy(q, src, n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(HNS_ERR_HIP, "hns_dist_create: uploading the halo tables failed");
			q += pad256(n);
			return r;
		};
		d->d_perm = (int*)put(d->owned_perm.data(), sizeof(int) * d->owned_perm.size());
		if (d->owned_perm.empty()) q += 256;
		for (Peer& p : d->peers) {
			for (int t = 0; t < X_COUNT; ++t)
				for (Region* r : {&p.send[t], &p.recv[t]}) {
					r->d_leaf = (int*)put(r->leaf.data(), sizeof(int) * r->leaf.size());
					r->d_mask = (unsigned char*)put(r->mask.data(), r->mask.size());
					r->d_off = (int*)put(r->off.data(), sizeof(int) * r->off.size());
				}
			for (int i = 0; i < 2; ++i) {
				p.sbuf[i] = (float*)q, q += pad256(sizeof(float) * p.sbuf_floats);
				p.rbuf[i] = (float*)q, q += pad256(sizeof(float) * p.rbuf_floats);
			}
		}
		if (batch)
			for (int t = 0; t < X_COUNT; ++t)
				for (int dir = 0; dir < 2; ++dir) {
					std::vector<int> leaf, off, peer;
					std::vector<unsigned char> mask;
					for (size_t pi = 0; pi < d->peers.size(); ++pi) {
						const Region& r = dir ? d->peers[pi].recv[t] : d->peers[pi].send[t];
						if (r.direct >= 0) continue;  // travels straight out of / into the field
						leaf.insert(leaf.end(), r.leaf.begin(), r.leaf.end());
						off.insert(off.end(), r.off.begin(), r.off.end());
						mask.insert(mask.end(), r.mask.begin(), r.mask.end());
						peer.insert(peer.end(), r.leaf.size(), (int)pi);
					}
					hns_dist::AllPeers& a = dir ? d->all_recv[t] : d->all_send[t];
					a.n = (int)leaf.size();
					a.d_leaf = (int*)put(leaf.data(), sizeof(int) * leaf.size());
					a.d_mask = (unsigned char*)put(mask.data(), mask.size());
					a.d_off = (int*)put(off.data(), sizeof(int) * off.size());
					a.d_peer = (int*)put(peer.data(), sizeof(int) * peer.size());
				}
		if (rc != HNS_OK) return bail(rc);
	}
	if ((rc = build_pack_tables(d)) != HNS_OK) return bail(rc);
	for (int i = 0; i < 2; ++i)
		if (hipEventCreateWithFlags(&d->ev_post[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&d->ev_done[i], hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&d->ev_bdone[i], hipEventDisableTiming) != hipSuccess)
			return bail(fail(HNS_ERR_HIP, "hns_dist_create: event creation failed"));
	if (hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming) != hipSuccess) return bail(fail(HNS_ERR_HIP, "hns_dist_create: event creation failed"));
	if (err) *err = HNS_OK;
	return d;
}

// The communication stream exists only where a second stream is used: RCCL and loopback transports. It outranks the compute
// stream: its short kernels (boundary leaves, pack, unpack) must not queue behind the thousands of waves of the interior
// kernel they run next to.
static int ensure_comm_stream(hns_dist* d) {
	if (d->cs) return HNS_OK;
	int lo_prio = 0, hi_prio = 0;
	(void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
	HNS_HIP(hipStreamCreateWithPriority(&d->cs, hipStreamNonBlocking, hi_prio));
	d->cs_owner = std::shared_ptr<void>((void*)d->cs, [](void* s) { (void)hipStreamDestroy((hipStream_t)s); });
	return HNS_OK;
}

int hns_dist_unique_id(void* out128) {
	if (!out128) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_unique_id: null argument");
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	HNS_TRY(need_rccl("hns_dist_unique_id"));
	ncclUniqueId id;
	HNS_NCCL(rccl().GetUniqueId(&id));
	memcpy(out128, &id, sizeof(id));
	return HNS_OK;
}

int hns_dist_connect_rccl(hns_dist* d, const void* unique_id128) {
	if (!d || !unique_id128) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_rccl: null argument");
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_connect_rccl: plan-only handle");
	if (d->comm || !d->local_ranks.empty()) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_rccl: already connected");
	HNS_TRY(need_rccl("hns_dist_connect_rccl"));
	HNS_TRY(ensure_comm_stream(d));
	ncclUniqueId id;
	memcpy(&id, unique_id128, sizeof(id));
	HNS_NCCL(rccl().CommInitRank(&d->comm, d->world, id, d->rank));
	return HNS_OK;
}

// Timing only: this rank alone on the device, every message answered with this rank's own payload (wrong data, right
// sizes, same streams / events / kernels). What one rank costs next to the plain single-GPU substep, before any wire time.
// The loopback transports answer every message out of the rank's own send buffer: timing only, the ghost values (and with them
// the back-traces) mean nothing -- no point in reporting that they leave the ghost layer.
static void no_far_check(hns_dist* d) {
	for (hns_grid* g : {d->gB, d->gI, d->gO, d->gA})
		if (g) g->far_flag = nullptr;
}

int hns_dist_connect_loopback(hns_dist* d) {
	if (!d || !d->gA) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_loopback: bad handle");
	if (d->comm || !d->local_ranks.empty()) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_loopback: already connected");
	HNS_TRY(ensure_comm_stream(d));
	d->loopback = true;
	no_far_check(d);
	if (mirror_wanted(d)) {  // the chained substep, looped back: boundary values go into this rank's own ghost leaves, flags to itself
		HNS_TRY(ensure_flags(d));
		std::vector<std::vector<int>> remote[X_COUNT];
		std::vector<char*> arenas;
		std::vector<uint64_t> units;
		std::vector<uint32_t*> fl;
		for (int t = 0; t < X_COUNT; ++t) remote[t].resize(d->peers.size());
		for (size_t i = 0; i < d->peers.size(); ++i) {
			const Peer& p = d->peers[i];
			for (int t = 0; t < X_COUNT; ++t)
				for (size_t j = 0; j < p.send[t].leaf.size(); ++j)
					remote[t][i].push_back(p.recv[t].leaf.empty() ? d->nB + d->nI : p.recv[t].leaf[j % p.recv[t].leaf.size()]);
			arenas.push_back((char*)d->arena), units.push_back((uint64_t)d->unit_bytes);
			fl.push_back(d->ipc_flags + (p.rank - d->rank));  // (so that the flag this rank raises "on the peer" is the one it waits for)
		}
		if (d->nG > 0) HNS_TRY(setup_mirror(d, remote, arenas, units, fl));
	}
	return HNS_OK;
}

// The same, but every message really goes through RCCL: a one-rank communicator, ncclSend / ncclRecv to itself in the groups
// the multi-rank path issues (same entry points, argument order, per-field segments, streams). What a single-GPU box can
// verify of the RCCL path: it must leave exactly what the copy-based loopback leaves.
int hns_dist_connect_loopback_rccl(hns_dist* d) {
	if (!d || !d->gA) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_loopback_rccl: bad handle");
	if (d->comm || d->loopback || !d->local_ranks.empty()) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_loopback_rccl: already connected");
	HNS_TRY(need_rccl("hns_dist_connect_loopback_rccl"));
	HNS_TRY(ensure_comm_stream(d));
	ncclUniqueId id;
	HNS_NCCL(rccl().GetUniqueId(&id));
	HNS_NCCL(rccl().CommInitRank(&d->comm, 1, id, 0));
	d->loopback = true;
	no_far_check(d);
	return HNS_OK;
}

// ---- "ipc" transport: one process per GPU, peers' memory mapped with hipIpc*, one-sided puts and flags ----
namespace {
struct IpcBlob {  // what a rank tells the others (hns_dist_ipc_export): plain data, HNS_DIST_IPC_BLOB_BYTES on the wire
	uint32_t magic, world, rank, n_peers;
	uint64_t unit_bytes, pid;
	hipIpcMemHandle_t arena, tables, flags;
	struct PeerInfo {
		int32_t rank;
		int32_t recv_direct[4], recv_voxels[4];
		uint64_t rbuf_off[2];
		uint64_t recv_leaf_off[4];  // where (in the tables allocation) the local indices of the ghost leaves of each region type are
		uint32_t recv_leaves[4];
	} peer[kIpcMaxPeers];
};
static_assert(sizeof(IpcBlob) <= HNS_DIST_IPC_BLOB_BYTES, "HNS_DIST_IPC_BLOB_BYTES is too small");
constexpr uint32_t kIpcMagic = 0x48495043u;
}  // namespace

int hns_dist_ipc_export(hns_dist* d, void* out_blob) {
	if (!d || !out_blob) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_ipc_export: null argument");
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_ipc_export: plan-only handle");
	if (d->peers.size() > (size_t)kIpcMaxPeers || d->world > kFlagSlots) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_ipc_export: more than 16 peers or 512 ranks");
	HNS_TRY(ensure_flags(d));
	IpcBlob b;
	memset(&b, 0, sizeof(b));
	b.magic = kIpcMagic, b.world = (uint32_t)d->world, b.rank = (uint32_t)d->rank, b.n_peers = (uint32_t)d->peers.size();
	b.unit_bytes = d->unit_bytes, b.pid = (uint64_t)getpid();
	HNS_HIP(hipIpcGetMemHandle(&b.arena, d->arena));
	HNS_HIP(hipIpcGetMemHandle(&b.tables, d->tables));
	HNS_HIP(hipIpcGetMemHandle(&b.flags, d->ipc_flags));
	for (size_t i = 0; i < d->peers.size(); ++i) {
		const Peer& p = d->peers[i];
		b.peer[i].rank = p.rank;
		for (int t = 0; t < X_COUNT; ++t) b.peer[i].recv_direct[t] = p.recv[t].direct, b.peer[i].recv_voxels[t] = p.recv[t].voxels;
		for (int k = 0; k < 2; ++k) b.peer[i].rbuf_off[k] = (uint64_t)((char*)p.rbuf[k] - (char*)d->tables);
		for (int t = 0; t < X_COUNT; ++t)
			b.peer[i].recv_leaf_off[t] = (uint64_t)((char*)p.recv[t].d_leaf - (char*)d->tables), b.peer[i].recv_leaves[t] = (uint32_t)p.recv[t].leaf.size();
	}
	memset(out_blob, 0, HNS_DIST_IPC_BLOB_BYTES);
	memcpy(out_blob, &b, sizeof(b));
	return HNS_OK;
}

// Collective in effect: every rank exports, the blobs travel by any host means (DistRank.connect_ipc gathers them over
// torch.distributed), every rank connects with all `world` blobs in rank order. Ranks must be separate processes.
int hns_dist_connect_ipc(hns_dist* d, const void* blobs) {
	if (!d || !blobs) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_ipc: null argument");
	if (!d->gA || !d->ipc_flags) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_ipc: call hns_dist_ipc_export first");
	if (d->comm || d->loopback || d->ipc || !d->local_ranks.empty()) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_ipc: already connected");
	HNS_TRY(ensure_comm_stream(d));
	d->ipc_peers.assign(d->peers.size(), hns_dist::IpcPeer());
	for (size_t i = 0; i < d->peers.size(); ++i) {
		IpcBlob b;
		memcpy(&b, (const char*)blobs + (size_t)d->peers[i].rank * HNS_DIST_IPC_BLOB_BYTES, sizeof(b));
		if (b.magic != kIpcMagic || (int)b.world != d->world || (int)b.rank != d->peers[i].rank) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_ipc: blob of the wrong rank or world");
		if (b.pid == (uint64_t)getpid()) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_ipc: ranks must be separate processes (use hns_dist_connect_local inside one)");
		const IpcBlob::PeerInfo* me = nullptr;
		for (uint32_t k = 0; k < b.n_peers && k < (uint32_t)kIpcMaxPeers; ++k)
			if (b.peer[k].rank == d->rank) me = &b.peer[k];
		if (!me) return fail(HNS_ERR_RUNTIME, "hns_dist_connect_ipc: a peer does not list this rank (plans disagree)");
		hns_dist::IpcPeer& q = d->ipc_peers[i];
		for (int t = 0; t < X_COUNT; ++t) {
			if (me->recv_voxels[t] != d->peers[i].send[t].voxels) return fail(HNS_ERR_RUNTIME, "hns_dist_connect_ipc: send/receive plans of two ranks disagree");
			q.recv_direct[t] = me->recv_direct[t], q.recv_voxels[t] = me->recv_voxels[t];
		}
		q.unit_bytes = b.unit_bytes, q.rbuf_off[0] = me->rbuf_off[0], q.rbuf_off[1] = me->rbuf_off[1];
		HNS_HIP(hipIpcOpenMemHandle(&q.opened[0], b.arena, hipIpcMemLazyEnablePeerAccess));
		HNS_HIP(hipIpcOpenMemHandle(&q.opened[1], b.tables, hipIpcMemLazyEnablePeerAccess));
		HNS_HIP(hipIpcOpenMemHandle(&q.opened[2], b.flags, hipIpcMemLazyEnablePeerAccess));
		q.arena = (char*)q.opened[0], q.tables = (char*)q.opened[1], q.flags = (uint32_t*)q.opened[2];
		for (int t = 0; t < X_COUNT; ++t) q.recv_leaf_off[t] = me->recv_leaf_off[t], q.recv_leaves[t] = me->recv_leaves[t];
	}
	d->ipc = true;
	if (mirror_wanted(d)) {
		std::vector<std::vector<int>> remote[X_COUNT];
		std::vector<char*> arenas;
		std::vector<uint64_t> units;
		std::vector<uint32_t*> fl;
		for (int t = 0; t < X_COUNT; ++t) remote[t].resize(d->peers.size());
		for (size_t i = 0; i < d->peers.size(); ++i) {
			const hns_dist::IpcPeer& q = d->ipc_peers[i];
			for (int t = 0; t < X_COUNT; ++t) {
				remote[t][i].resize(q.recv_leaves[t]);
				if (q.recv_leaves[t]) HNS_HIP(hipMemcpy(remote[t][i].data(), q.tables + q.recv_leaf_off[t], sizeof(int) * q.recv_leaves[t], hipMemcpyDeviceToHost));
			}
			arenas.push_back(q.arena), units.push_back(q.unit_bytes), fl.push_back(q.flags);
		}
		HNS_TRY(setup_mirror(d, remote, arenas, units, fl));
	}
	return HNS_OK;
}

int hns_dist_connect_local(hns_dist* const* ranks, int world) {
	if (!ranks || world < 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_local: bad arguments");
	for (int r = 0; r < world; ++r)
		if (!ranks[r] || !ranks[r]->gA || ranks[r]->world != world || ranks[r]->rank != r || ranks[r]->comm || ranks[r]->device != ranks[0]->device)
			return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_connect_local: ranks[r] must be rank r of this world, unconnected, all on one device");
	// Locally connected ranks run EVERYTHING on the caller's stream, boundary kernels and messages included. With a
	// communication stream per emulated rank (nine streams in an 8-rank test) the runtime multiplexed them onto its handful of
	// hardware queues and the device stalled for minutes at a time, sporadically even with one shared communication stream;
	// a single stream in host order cannot. The plan, launch ranges, kernels, pack / unpack and message buffers are the ones
	// the RCCL path uses; its two-stream overlap is exercised by the loopback transport (one rank, two streams).
	for (int r = 0; r < world; ++r) {
		ranks[r]->single_stream = true;
		ranks[r]->local_ranks.assign(ranks, ranks + world);
	}
	bool want = world > 1;
	for (int r = 0; r < world; ++r) want = want && mirror_wanted(ranks[r]);
	if (want) {
		for (int r = 0; r < world; ++r) HNS_TRY(ensure_flags(ranks[r]));
		for (int r = 0; r < world; ++r) {
			hns_dist* d = ranks[r];
			std::vector<std::vector<int>> remote[X_COUNT];
			std::vector<char*> arenas;
			std::vector<uint64_t> units;
			std::vector<uint32_t*> fl;
			for (int t = 0; t < X_COUNT; ++t) remote[t].resize(d->peers.size());
			for (size_t i = 0; i < d->peers.size(); ++i) {
				hns_dist* q = ranks[d->peers[i].rank];
				for (const Peer& c : q->peers)
					if (c.rank == d->rank)
						for (int t = 0; t < X_COUNT; ++t) remote[t][i] = c.recv[t].leaf;
				arenas.push_back((char*)q->arena), units.push_back((uint64_t)q->unit_bytes), fl.push_back(q->ipc_flags);
			}
			HNS_TRY(setup_mirror(d, remote, arenas, units, fl));
		}
	}
	return HNS_OK;
}

// ---- plan queries (also on HNS_DIST_PLAN_ONLY handles) ----
int hns_dist_local_leaves(const hns_dist* d, int64_t* out_global_ids) {
	if (!d || !out_global_ids) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_local_leaves: null argument");
	std::copy(d->local_global.begin(), d->local_global.end(), out_global_ids);
	return HNS_OK;
}

int hns_dist_peer_rank(const hns_dist* d, int peer) { return d && peer >= 0 && peer < (int)d->peers.size() ? d->peers[(size_t)peer].rank : -1; }

int hns_dist_peer_region(const hns_dist* d, int peer, int type, int is_send, int32_t* leaves, unsigned char* masks, uint64_t* n_leaves, uint64_t* n_voxels) {
	if (!d || peer < 0 || peer >= (int)d->peers.size() || type < 0 || type >= X_COUNT) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_peer_region: bad arguments");
	const Region& r = is_send ? d->peers[(size_t)peer].send[type] : d->peers[(size_t)peer].recv[type];
	if (n_leaves) *n_leaves = r.leaf.size();
	if (n_voxels) *n_voxels = (uint64_t)r.voxels;
	if (leaves) std::copy(r.leaf.begin(), r.leaf.end(), leaves);
	if (masks) std::copy(r.mask.begin(), r.mask.end(), masks);
	return HNS_OK;
}

uint64_t hns_dist_owned_leaves(const hns_dist* d) { return d ? (uint64_t)(d->nB + d->nI) : 0; }
// the first owned leaf when the owned leaves are a contiguous run of the caller's list (part_axis -1: every partition of rounds 1-4), else ~0
uint64_t hns_dist_first_owned_leaf(const hns_dist* d) {
	if (!d || d->nB + d->nI == 0) return 0;
	return d->part_axis < 0 ? (uint64_t)d->owned_global.front() : ~(uint64_t)0;
}
int hns_dist_owned_leaf_ids(const hns_dist* d, int64_t* out_global_ids) {
	if (!d || !out_global_ids) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_owned_leaf_ids: null argument");
	std::copy(d->owned_global.begin(), d->owned_global.end(), out_global_ids);
	return HNS_OK;
}
int hns_dist_partition_axis(const hns_dist* d) { return d ? d->part_axis : -1; }
// sweeps_per_exchange of the chained one-sided substep for a decomposition of this size: 2 where the ranks' owned ranges are swept in 16^3 blocks, else 1
int hns_dist_one_sided_sweeps(uint64_t n_leaves, int world) { return blocked_mirror_rule(2, world, (int64_t)n_leaves) ? 2 : 1; }

int hns_dist_info(const hns_dist* d, hns_dist_stats* out) {
	if (!d || !out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_info: null argument");
	memset(out, 0, sizeof(*out));
	out->world = d->world, out->rank = d->rank, out->sweeps_per_exchange = d->k;
	out->boundary_leaves = (uint64_t)d->nB, out->interior_leaves = (uint64_t)d->nI, out->ghost_leaves = (uint64_t)d->nG;
	out->peers = (int)d->peers.size();
	for (int t = 0; t < X_COUNT; ++t) {
		out->bytes_sent[t] = d->bytes_sent[t];
		for (const Peer& p : d->peers) out->region_voxels_sent[t] += (uint64_t)p.send[t].voxels;
	}
	out->messages_sent = d->messages_sent, out->exchanges = d->exchanges, out->packed_exchanges = d->packed_exchanges;
	out->chained = d->mirror && d->chain ? 1 : 0;
	for (const Peer& p : d->peers) {
		bool halo = false;
		for (int t = 1; t < X_COUNT; ++t) halo = halo || p.send[t].voxels || p.recv[t].voxels;
		out->halo_peers += halo ? 1 : 0;
	}
	return HNS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// upload / download: host arrays over the OWNED leaves in partition order (hns_dist_owned_leaf_ids; ascending global ids when part_axis < 0)
// ---------------------------------------------------------------------------------------------------------------

static int drain(hns_dist* d, hipStream_t st) {
	// a posted exchange whose data nobody will consume (new fields are coming): let it finish, then forget it
	if (d->pending.active) {
		HNS_HIP(hipStreamSynchronize(st));
		if (d->cs) HNS_HIP(hipStreamSynchronize(d->cs));
		d->pending.active = false;
	}
	d->phi_in_flight = false;
	d->u_ghosts_fresh = false;
	return HNS_OK;
}

int hns_dist_upload(hns_dist* d, const float* vel3, const float* const* scalars, void* stream) {
	if (!d || !vel3 || (d->n_scalars && !scalars)) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_upload: null argument");
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_upload: plan-only handle (there is no CPU fallback)");
	hipStream_t st = (hipStream_t)stream;
	HNS_TRY(drain(d, st));
	if (d->far_status) *d->far_status = 0;  // new fields: whatever an earlier back-trace did is history
	const int nO = d->nB + d->nI;
	if (nO == 0) return HNS_OK;
	for (int f = -1; f < d->n_scalars; ++f) {
		const int nc = f < 0 ? 3 : 1;
		const float* src = f < 0 ? vel3 : scalars[f];
		if (!src) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_upload: null field");
		HNS_HIP(hipMemcpyAsync(d->stage, src, sizeof(float) * 512 * (size_t)nO * nc, hipMemcpyHostToDevice, st));
		HNS_TRY(hns_dev_pack_leaves(d->stage, d->d_perm, (uint64_t)nO, f < 0 ? d->u : d->phi[(size_t)f], nc, st));  // field[local] = staged[perm[local]]
	}
	HNS_HIP(hipStreamSynchronize(st));
	return HNS_OK;
}

int hns_dist_download(hns_dist* d, float* vel3, float* const* scalars, float* pressure, void* stream) {
	if (!d) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_download: null argument");
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_download: plan-only handle");
	hipStream_t st = (hipStream_t)stream;
	const int nO = d->nB + d->nI;
	if (nO == 0) return HNS_OK;
	if (d->cs) HNS_HIP(hipStreamSynchronize(d->cs));  // the boundary leaves' values are written on the communication stream
	for (int f = -2; f < d->n_scalars; ++f) {
		const int nc = f == -1 ? 3 : 1;
		float* dst = f == -2 ? pressure : (f == -1 ? vel3 : (scalars ? scalars[f] : nullptr));
		if (!dst) continue;
		const float* src = f == -2 ? d->p_result : (f == -1 ? d->u : d->phi[(size_t)f]);
		HNS_TRY(hns_dev_unpack_leaves(src, d->d_perm, (uint64_t)nO, d->stage, nc, st));  // staged[perm[local]] = field[local]
		HNS_HIP(hipMemcpyAsync(dst, d->stage, sizeof(float) * 512 * (size_t)nO * nc, hipMemcpyDeviceToHost, st));
		HNS_HIP(hipStreamSynchronize(st));
	}
	return far_check(d);
}

// Diagnostics: one field of ALL local leaves, ghosts included, in local order [boundary | interior | ghosts], as the device holds
// it now. which: -2 = the last solve's p, -1 = velocity (3 floats per voxel), s >= 0 = scalar s.
int hns_dist_download_local(hns_dist* d, int which, float* out, void* stream) {
	if (!d || !out) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_download_local: null argument");
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_download_local: plan-only handle");
	if (which < -2 || which >= d->n_scalars) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_download_local: no such field");
	hipStream_t st = (hipStream_t)stream;
	HNS_HIP(hipStreamSynchronize(st));
	if (d->cs) HNS_HIP(hipStreamSynchronize(d->cs));
	const float* src = which == -2 ? d->p_result : (which == -1 ? d->u : d->phi[(size_t)which]);
	const size_t n = d->local_global.size();
	if (n == 0) return HNS_OK;
	if (!src) return fail(HNS_ERR_RUNTIME, "hns_dist_download_local: the field does not exist yet");
	HNS_HIP(hipMemcpy(out, src, sizeof(float) * 512 * n * (which == -1 ? 3 : 1), hipMemcpyDeviceToHost));
	return HNS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// exchange: post (pack, hand to the communication stream) / complete (wait, unpack)
// ---------------------------------------------------------------------------------------------------------------

namespace {

int halo_copy(bool pack, float* field, int ncomp, const Region& r, float* msg, hipStream_t st) {
	if (r.leaf.empty()) return HNS_OK;
	if (r.whole) return pack ? hns_dev_pack_leaves(field, r.d_leaf, r.leaf.size(), msg, ncomp, st) : hns_dev_unpack_leaves(msg, r.d_leaf, r.leaf.size(), field, ncomp, st);
	const dim3 grid((unsigned)r.leaf.size()), block(64);
	if (ncomp == 3) {
		if (pack)
			hipLaunchKernelGGL((k_halo_copy<3, true>), grid, block, 0, st, field, r.d_leaf, r.d_mask, r.d_off, msg);
		else
			hipLaunchKernelGGL((k_halo_copy<3, false>), grid, block, 0, st, field, r.d_leaf, r.d_mask, r.d_off, msg);
	} else {
		if (pack)
			hipLaunchKernelGGL((k_halo_copy<1, true>), grid, block, 0, st, field, r.d_leaf, r.d_mask, r.d_off, msg);
		else
			hipLaunchKernelGGL((k_halo_copy<1, false>), grid, block, 0, st, field, r.d_leaf, r.d_mask, r.d_off, msg);
	}
	return launch_status("hns_dist: halo pack/unpack");
}

// every field of exchange `x`, all peers: one launch per field where the combined tables exist, per peer otherwise
int halo_copy_exchange(hns_dist* d, bool pack, const Pending& x, hipStream_t st) {
	const hns_dist::AllPeers& a = pack ? d->all_send[x.type] : d->all_recv[x.type];
	if (a.d_leaf) {
		if (a.n == 0) return HNS_OK;
		PeerMsgs msgs;
		for (size_t pi = 0; pi < d->peers.size(); ++pi) {
			msgs.base[pi] = pack ? d->peers[pi].sbuf[x.parity] : d->peers[pi].rbuf[x.parity];
			msgs.voxels[pi] = (pack ? d->peers[pi].send[x.type] : d->peers[pi].recv[x.type]).voxels;
		}
		int before = 0;
		const dim3 grid((unsigned)a.n), block(64);
		for (auto& f : x.fields) {
			if (f.second == 3) {
				if (pack)
					hipLaunchKernelGGL((k_halo_copy_all<3, true>), grid, block, 0, st, f.first, a.d_leaf, a.d_mask, a.d_off, a.d_peer, msgs, before);
				else
					hipLaunchKernelGGL((k_halo_copy_all<3, false>), grid, block, 0, st, f.first, a.d_leaf, a.d_mask, a.d_off, a.d_peer, msgs, before);
			} else {
				if (pack)
					hipLaunchKernelGGL((k_halo_copy_all<1, true>), grid, block, 0, st, f.first, a.d_leaf, a.d_mask, a.d_off, a.d_peer, msgs, before);
				else
					hipLaunchKernelGGL((k_halo_copy_all<1, false>), grid, block, 0, st, f.first, a.d_leaf, a.d_mask, a.d_off, a.d_peer, msgs, before);
			}
			before += f.second;
		}
		return launch_status("hns_dist: halo pack/unpack");
	}
	for (Peer& p : d->peers) {
		float* msg = pack ? p.sbuf[x.parity] : p.rbuf[x.parity];
		const Region& r = pack ? p.send[x.type] : p.recv[x.type];
		if (r.direct >= 0) continue;
		for (auto& f : x.fields) {
			HNS_TRY(halo_copy(pack, f.first, f.second, r, msg, st));
			msg += (size_t)f.second * (size_t)r.voxels;
		}
	}
	return HNS_OK;
}

// where field `f` (ncomp components, `before` components of earlier fields ahead of it) of a message lives: in the field itself
// when the region is a run of whole consecutive leaves, in the message buffer otherwise
float* segment(const Region& r, float* field, int ncomp, float* buf, int before) {
	return r.direct >= 0 ? field + (size_t)r.direct * 512 * (size_t)ncomp : buf + (size_t)before * (size_t)r.voxels;
}

size_t message_floats(const Pending& x, const Region& r) {
	size_t c = 0;
	for (auto& f : x.fields) c += (size_t)f.second;
	return c * (size_t)r.voxels;
}

// received regions -> ghost voxels, on the communication stream; ev_done marks the end of the exchange
int unpack(hns_dist* d, Pending& x) {
	HNS_TRY(halo_copy_exchange(d, false, x, x.stream));
	if (!d->single_stream) HNS_HIP(hipEventRecord(d->ev_done[x.parity], x.stream));
	return HNS_OK;
}

// One exchange. post(): the compute stream `st` marks "everything the boundary kernel reads is ready", and the rank's
// communication stream takes over the whole boundary side of the step: `boundary(cs)` runs the kernel on the boundary leaves,
// the regions the peers read are packed, the messages travel, the received regions are unpacked into the ghost voxels. The
// caller then launches the interior kernel on `st`, which runs concurrently with all of that (an interior leaf touches no
// ghost and no ghost-facing leaf writes what it reads). complete(): `st` waits for the end of that chain.
// RCCL: sends and receives are one group on the communication stream. Local: the peers pull at complete().
// Round 6: `interior(st)` -- the same kernel over the interior leaves -- is handed in and enqueued HERE, right behind the boundary kernel and in front of the pack / transfer /
// unpack calls. Enqueued after them (rounds 2-5) it reached the device only once the host had issued the whole boundary chain, and by then that chain had run: the two
// streams never overlapped (profiles/r06_dist_exchanged_timeline_before.txt; the loop is bound by the HOST's ~10 runtime calls per exchange).
// in_line: the whole exchange -- `boundary(st)`, pack, transfer, unpack -- on the compute stream itself, in order, no events, complete on return (round 6: the pressure loop
// whose one launch over all owned leaves packs its own messages; also what locally connected ranks do).
template <class BoundaryFn, class InteriorFn>
int post(hns_dist* d, int type, std::vector<std::pair<float*, int>> fields, hipStream_t st, BoundaryFn boundary, InteriorFn interior, bool in_line = false) {
	if (d->pending.active) return fail(HNS_ERR_RUNTIME, "hns_dist: an exchange is already in flight");
	if (d->world == 1) {  // nobody to talk to: the boundary range is empty, keep everything on one stream
		HNS_TRY(boundary(st));
		return interior(st);
	}
	if (!d->comm && !d->loopback && !d->ipc && d->local_ranks.empty())
		return fail(HNS_ERR_RUNTIME, "hns_dist: not connected (call hns_dist_connect_rccl, hns_dist_connect_ipc or hns_dist_connect_local first)");
	Pending& x = d->pending;
	x.active = true, x.type = type, x.parity = d->parity, x.fields = std::move(fields);
	d->parity ^= 1;
	const bool one_stream = d->single_stream || in_line;
	const hipStream_t cs = one_stream ? st : d->cs;
	x.stream = cs;
	if (!one_stream) {
		HNS_HIP(hipEventRecord(d->ev_ready, st));
		HNS_HIP(hipStreamWaitEvent(cs, d->ev_ready, 0));
	}
	x.prepacked = false;
	HNS_TRY(boundary(cs));
	if (!one_stream) HNS_HIP(hipEventRecord(d->ev_bdone[x.parity], cs));
	HNS_TRY(interior(st));
	if (!x.prepacked) HNS_TRY(halo_copy_exchange(d, true, x, cs));
	else ++d->packed_exchanges;
	for (Peer& p : d->peers) {
		const size_t fl = message_floats(x, p.send[type]);
		if (fl) d->bytes_sent[type] += sizeof(float) * fl, ++d->messages_sent;
	}
	++d->exchanges;
	if (d->ipc) {
		const uint32_t seq = ++d->ipc_seq;
		IpcPeers ip;
		ip.n = (int)d->peers.size();
		for (int i = 0; i < ip.n; ++i) {
			ip.theirs_ready[i] = d->ipc_peers[(size_t)i].flags + d->rank;
			ip.theirs_landed[i] = d->ipc_peers[(size_t)i].flags + kFlagLanded + d->rank;
			ip.rank[i] = d->peers[(size_t)i].rank;
		}
		hipLaunchKernelGGL(k_ipc_ready, dim3(1), dim3(64), 0, cs, ip, (const uint32_t*)d->ipc_flags, seq, d->ipc_status);
		IpcSegs sg;
		sg.n = 0, sg.wg0[0] = 0;
		auto flush = [&]() {
			if (sg.n) hipLaunchKernelGGL(k_ipc_put<true>, dim3(sg.wg0[sg.n]), dim3(256), 0, cs, sg);
			sg.n = 0;
		};
		for (size_t i = 0; i < d->peers.size(); ++i) {
			Peer& p = d->peers[i];
			const hns_dist::IpcPeer& q = d->ipc_peers[i];
			const Region& rs = p.send[type];
			int before = 0;
			for (auto& f : x.fields) {
				const size_t fl = (size_t)rs.voxels * (size_t)f.second;
				if (fl) {
					if (sg.n == kIpcMaxSegs) flush();
					// the same field in the peer's memory: fields sit at the same multiples of the (peer's) unit
					const size_t unit_index = (size_t)((char*)f.first - (char*)d->arena) / d->unit_bytes;
					char* dst = q.recv_direct[type] >= 0 ? q.arena + unit_index * q.unit_bytes + sizeof(float) * 512 * (size_t)q.recv_direct[type] * (size_t)f.second
					                                     : q.tables + q.rbuf_off[x.parity] + sizeof(float) * (size_t)before * (size_t)q.recv_voxels[type];
					sg.dst[sg.n] = (float*)dst, sg.src[sg.n] = segment(rs, f.first, f.second, p.sbuf[x.parity], before), sg.floats[sg.n] = (unsigned)fl;
					sg.wg0[sg.n + 1] = sg.wg0[sg.n] + (unsigned)((fl + 4095) / 4096);
					++sg.n;
				}
				before += f.second;
			}
		}
		flush();
		hipLaunchKernelGGL(k_ipc_landed, dim3(1), dim3(64), 0, cs, ip, (const uint32_t*)d->ipc_flags, seq, d->ipc_status);
		HNS_TRY(launch_status("hns_dist: one-sided exchange"));
	} else if (d->comm) {
		HNS_NCCL(rccl().GroupStart());
		for (Peer& p : d->peers) {
			const Region &rs = p.send[type], &rr = p.recv[type];
			// (loopback over RCCL: the peer is this rank, the answer to a message is the message, cut to the smaller region)
			const int to = d->loopback ? 0 : p.rank;
			const size_t vs = d->loopback ? (size_t)std::min(rs.voxels, rr.voxels) : (size_t)rs.voxels, vr = d->loopback ? vs : (size_t)rr.voxels;
			int before = 0;
			for (auto& f : x.fields) {  // one send and one receive per field: either end may use the field itself or its buffer
				if (vs) HNS_NCCL(rccl().Send(segment(rs, f.first, f.second, p.sbuf[x.parity], before), vs * f.second, ncclFloat, to, d->comm, cs));
				if (vr) HNS_NCCL(rccl().Recv(segment(rr, f.first, f.second, p.rbuf[x.parity], before), vr * f.second, ncclFloat, to, d->comm, cs));
				before += f.second;
			}
		}
		HNS_NCCL(rccl().GroupEnd());
	} else if (d->loopback) {  // same streams, events and copy sizes as a real exchange, but the payload is this rank's own
		if (const int us = options().dist_wire_us.load()) hipLaunchKernelGGL(k_wire_delay, dim3(1), dim3(1), 0, cs, (long long)us * 100);
		// all messages of the exchange as ONE copy launch (round 6; a hipMemcpyAsync per peer and field cost the host 5-8 us each, two to fourteen of them per exchange):
		// what stands in for the one send / receive group of the RCCL path
		IpcSegs sg;
		sg.n = 0, sg.wg0[0] = 0;
		auto flush = [&]() {
			if (sg.n) hipLaunchKernelGGL(k_ipc_put<false>, dim3(sg.wg0[sg.n]), dim3(256), 0, cs, sg);
			sg.n = 0;
		};
		for (Peer& p : d->peers) {
			const Region &rs = p.send[type], &rr = p.recv[type];
			int before = 0;
			for (auto& f : x.fields) {
				const size_t nr = (size_t)std::min(rs.voxels, rr.voxels) * (size_t)f.second;
				if (nr) {
					if (sg.n == kIpcMaxSegs) flush();
					sg.dst[sg.n] = segment(rr, f.first, f.second, p.rbuf[x.parity], before), sg.src[sg.n] = segment(rs, f.first, f.second, p.sbuf[x.parity], before), sg.floats[sg.n] = (unsigned)nr;
					sg.wg0[sg.n + 1] = sg.wg0[sg.n] + (unsigned)((nr + 4095) / 4096);
					++sg.n;
				}
				before += f.second;
			}
		}
		flush();
		HNS_TRY(launch_status("hns_dist: loopback exchange"));
	} else {
		if (!d->single_stream) HNS_HIP(hipEventRecord(d->ev_post[x.parity], cs));  // packed: the peers may pull
		return HNS_OK;
	}
	if (in_line && !d->single_stream) {  // received regions -> ghost voxels behind the transfer on the same stream: nothing left to wait for
		HNS_TRY(halo_copy_exchange(d, false, x, cs));
		x.active = false;
		return HNS_OK;
	}
	return unpack(d, x);
}

template <class BoundaryFn>
int post(hns_dist* d, int type, std::vector<std::pair<float*, int>> fields, hipStream_t st, BoundaryFn boundary) {
	return post(d, type, std::move(fields), st, boundary, [](hipStream_t) { return (int)HNS_OK; });
}

// Make the posted exchange's data visible in the ghost voxels before anything else runs on the compute stream.
int complete(hns_dist* d, hipStream_t st) {
	Pending& x = d->pending;
	if (!x.active) return HNS_OK;
	if (!d->comm && !d->loopback && !d->ipc) {  // local transport: pull every peer's message out of its send buffer, once the peer has packed it
		for (Peer& p : d->peers) {
			// the peer packed this message into its buffer of the same parity when it posted the same exchange; it may already
			// have posted the NEXT one (other parity) -- never the one after, which its own complete() of this one precedes
			hns_dist* q = d->local_ranks[(size_t)p.rank];
			const Peer* back = nullptr;
			for (const Peer& c : q->peers)
				if (c.rank == d->rank) back = &c;
			const size_t nr = message_floats(x, p.recv[x.type]);
			if (!nr) continue;
			if (!back || message_floats(x, back->send[x.type]) != nr) return fail(HNS_ERR_RUNTIME, "hns_dist: send/receive plans of two ranks disagree");
			const Pending& y = q->pending;  // the peer's record of the same exchange (its fields are ITS device arrays)
			if (y.type != x.type || y.parity != x.parity || y.fields.size() != x.fields.size()) return fail(HNS_ERR_RUNTIME, "hns_dist: locally connected ranks are out of step");
			if (!d->single_stream) HNS_HIP(hipStreamWaitEvent(x.stream, q->ev_post[x.parity], 0));
			int before = 0;
			for (size_t fi = 0; fi < x.fields.size(); ++fi) {
				const int nc = x.fields[fi].second;
				HNS_HIP(hipMemcpyAsync(segment(p.recv[x.type], x.fields[fi].first, nc, p.rbuf[x.parity], before),
				                       segment(back->send[x.type], y.fields[fi].first, nc, back->sbuf[x.parity], before), sizeof(float) * (size_t)p.recv[x.type].voxels * nc,
				                       hipMemcpyDeviceToDevice, x.stream));
				before += nc;
			}
		}
		HNS_TRY(unpack(d, x));
	}
	if (!d->single_stream) HNS_HIP(hipStreamWaitEvent(st, d->ev_done[x.parity], 0));
	x.active = false;
	return HNS_OK;
}

// Between two blocks of the exchanged pressure loop that sweep owned leaves only: the compute stream needs the boundary KERNEL of the posted exchange (interior tiles read the
// boundary leaves' new p), not its messages -- the ghost voxels are read by the next boundary kernel alone, which follows the unpack in stream order on the communication
// stream. So the compute stream waits for ev_bdone, the exchange is forgotten, and no cross-stream edge is left on the chain boundary sweep -> transfer -> unpack -> next
// boundary sweep. (The last exchange of a solve is completed in full by the phase behind it.) Two-stream transports only; false = the caller must complete() in full.
bool complete_boundary_only(hns_dist* d, hipStream_t st) {
	Pending& x = d->pending;
	if (!x.active) return true;
	if (d->single_stream || !(d->comm || d->loopback || d->ipc)) return false;
	if (hipStreamWaitEvent(st, d->ev_bdone[x.parity], 0) != hipSuccess) return false;
	x.active = false;
	return true;
}

// ---------------------------------------------------------------------------------------------------------------
// the core substep as a sequence of phases; a phase ends where an exchange has been posted
// ---------------------------------------------------------------------------------------------------------------

float omega_compute(float vs) { return 2.0f / (1.0f + sinf(static_cast<float>(3.14159) * vs)); }  // reference HNanoSolver.cu:257

struct Step {
	hns_dist* d;
	int iterations;
	float dt;
	hipStream_t st;
	// pressure loop cursor
	int it = 0;
	float *src = nullptr, *dst = nullptr;
	// the whole Compute_Sim substep (reference HNanoSolver.cu:150-356) instead of its core: combustion parameters, the positions of
	// fuel / waste / temperature / flame / collision_sdf among the rank's scalars, collision on, vorticity confinement on
	const hns_combustion_params* prm = nullptr;
	int fi[5] = {-1, -1, -1, -1, -1};
	bool coll = false, vort = false;

	bool full() const { return prm != nullptr; }
	int n_phases() const {
		const int blocks = (iterations + d->k - 1) / d->k;
		if (full()) return 1 + (coll ? 1 : 0) + 1 + (vort ? 1 : 0) + 1 + 1 + blocks + 1 + 1;  // open | [collision] | advect_vector | [vorticity] | divergence | combustion | blocks | gradient | advect_scalars
		return 1 + 1 + 1 + blocks + 1 + 1;  // open | advect_vector | divergence | pressure blocks | gradient | advect_scalars
	}
	const float* sdf() const { return coll ? d->phi[(size_t)fi[4]] : nullptr; }

	// Do both launch ranges of the split sweep take two iterations in ONE launch each (result in dst for both)? Asked of the library's own
	// plan, so that whatever hns_rbgs_iterate does with `2` is what this loop assumes.
	// Round 6: a SMALL rank (up to 16,384 owned leaves: BASELINE config 5 in 8 ranks has 8,243 each) runs its short phases -- the sweeps of the pressure loop, the divergence,
	// the gradient subtraction -- as ONE launch over all owned leaves with the exchange behind it on the compute stream (post(..., in_line)): at that size the boundary chain
	// (boundary kernel -> pack -> transfer -> unpack, each a latency-bound launch, plus two cross-stream event edges) is longer than the interior kernel it was meant to hide
	// under. Large ranks (a 256^3 slab: 32,768 leaves) keep the boundary / interior split on two streams. A rank decides for itself: the messages are the same either way.
	bool in_line_rank() const {
		return !d->single_stream && (d->comm || d->loopback || d->ipc) && options().dist_unsplit.load() != 0 && d->nB + d->nI <= 16384;
	}

	bool split_blocked() const {
		if (d->k < 2) return false;
		for (hns_grid* g : {d->gB, d->gI}) {
			if (!g->n_active) continue;
			int launches = 0, per = 0;
			if (hns_grid_rbgs_plan(g, 2, nullptr, 0, &launches, &per) != HNS_OK || launches != 1) return false;
		}
		return true;
	}

	int advect_scalars(hns_grid* g, float inv_dx, hipStream_t s) const {
		if (!d->n_scalars || !g->n_active) return HNS_OK;
		std::vector<const float*> in(d->phi.begin(), d->phi.end());
		return hns_dev_advect_scalars(g, d->u, in.data(), d->phi_next.data(), d->n_scalars, nullptr, 0, dt, inv_dx, s);
	}

	// One chained launch (hns_flags.hpp: PhaseMirror): `launch` runs the kernel over the owned leaves with the arguments `m`.
	template <class Launch>
	int chained(const PhaseMirror& m, Launch launch, bool gate = false) {
		if ((gate || options().dist_mirror.load() == 2) && m.n_peers) {
			// "guarded": ONE wave waits for the peers' previous launch in front of this one, so that no boundary workgroup ever
			// spins. For ranks that share a GPU (tests, bench.py --share-one-gpu): there the boundary waves of several processes
			// waiting inside their kernels can occupy every wave slot of the device, and the process they all wait for is never
			// scheduled (four 16k-leaf plume ranks: every bounded wait ran out). ~5 us per launch.
			PhaseMirror w = m;
			w.seq = m.seq - 1u;
			hipLaunchKernelGGL(k_sweep_wait, dim3(1), dim3(64), 0, st, w);
		}
		if (d->gO->n_active) {
			HNS_TRY(launch());
			// (locally connected ranks share ONE stream: a rank's flag must not wait for its next launch, which sits behind the
			// peers' launches that wait for the flag)
			if (d->single_stream && m.n_peers) hipLaunchKernelGGL(k_sweep_signal, dim3(1), dim3(64), 0, st, m);
		} else if (m.n_peers) {  // a rank without leaves still takes part in the chain of flags
			hipLaunchKernelGGL(k_sweep_signal, dim3(1), dim3(64), 0, st, m);
		}
		return launch_status("hns_dist: chained launch");
	}

	// one block of up to k sweeps with the halo of p exchanged behind it; all but the last sweep the ghost leaves too
	int sor_block_exchanged(int b) {
		hns_dist* D = d;
		typedef std::vector<std::pair<float*, int>> Fields;
		if (b == 0) it = 0, src = d->p_a, dst = d->p_b;  // never warm-started (reference HNanoSolver.cu:113): the first sweep reads no p
		const int n = std::min(d->k, iterations - it);
		// what the previous phase posted: in full in front of the first block (the divergence's ghosts) and wherever this block starts with sweeps over the ghost leaves;
		// between blocks that sweep owned leaves only, the boundary kernel alone (complete_boundary_only)
		// the one launch over all owned leaves that packs its own messages (below) where the owned range is swept in 16^3 blocks and the plan has pack tables for both region types
		const bool unsplit = in_line_rank() && d->pack_ok[X_P] && d->pack_ok[X_D1] && hns_rbgs_block_packable(d->gO);
		const int tail = unsplit ? std::min(n, 2) : ((n >= 2 && split_blocked()) ? 2 : 1);
		if (b == 0 || n > tail || !complete_boundary_only(d, st)) HNS_TRY(complete(d, st));
		if (b == 0 && d->timing && d->tev_used + 2 <= d->tev.size()) HNS_HIP(hipEventRecord(d->tev[d->tev_used], st));
		// Round 4: the sweeps of the block that the exchange follows are TWO iterations in one temporally blocked launch per range
		// (hns_sorblock.hip over a launch range: the ghost leaves are tile sources, 2K = 4 voxels deep, and are not swept; the X_P region
		// of a plan with k >= 2 reaches 2k >= 4 voxels, its div region 2k - 1 >= 3), where the library's plan for the ranges says so.
		// With k = 2 that is the whole pressure loop: no sweep ever touches a ghost leaf.
		if (n > tail) {  // the sweeps over owned + ghost leaves as ONE solve of n - tail iterations: the library picks the form (two iterations per launch where that pays)
			int in_b = 0;
			HNS_TRY(hns_rbgs_iterate(d->gA, d->div, src, dst, d->voxel_size, omega_compute(d->voxel_size), n - tail, &in_b, st, it == 0));
			if (in_b) std::swap(src, dst);
			it += n - tail;
		}
		const bool last = it + tail == iterations, zero = it == 0;
		float *s0 = src, *d0 = dst;
		const float vs = d->voxel_size;
		auto part = [=](hns_grid* g, hipStream_t s) {
			return g->n_active ? hns_rbgs_iterate(g, D->div, s0, d0, vs, omega_compute(vs), tail, nullptr, s, zero) : (int)HNS_OK;
		};
		const int xt = last ? X_D1 : X_P;
		// Round 6: ONE launch over all owned leaves that packs the peers' messages as it stores (PackMirror), then the transfer and the unpack behind it on the compute stream.
		// The boundary / interior split (below) buys overlap of the transfer with the interior sweep, and pays for it: a 16^3 block that straddles the boundary layer is swept by
		// both launches (config 5, rank 4 of 8: 14 + 14 us against 19 for the one launch), two cross-stream event edges per exchange, and twice the runtime calls -- traced, the
		// split loop's chain boundary sweep -> transfer -> unpack -> next boundary sweep alone took longer than this whole sequence (profiles/r06_dist_exchanged_notes.txt).
		if (unsplit) {
			HNS_TRY(post(d, xt, Fields{{dst, 1}}, st, [=](hipStream_t s) -> int {
				PackMirror m = D->pack_type[xt];
				for (size_t pi = 0; pi < D->peers.size(); ++pi) m.msg[pi] = D->peers[pi].sbuf[D->pending.parity];
				bool done = false;
				HNS_TRY(hns_rbgs_block_pack_launch(D->gO, D->div, s0, d0, vs, omega_compute(vs), zero, &m, s, &done, tail));
				if (!done) return fail(HNS_ERR_RUNTIME, "hns_dist: the owned range is not swept in 16^3 blocks after all");
				D->pending.prepacked = true;
				return HNS_OK;
			}, [](hipStream_t) { return (int)HNS_OK; }, true));
			std::swap(src, dst);
			it += tail;
			if (last) d->p_result = src;
			return HNS_OK;
		}
		HNS_TRY(post(d, xt, Fields{{dst, 1}}, st, [=](hipStream_t s) -> int {
			// two iterations in one blocked launch: the boundary sweep writes the peers' messages as it stores (PackMirror; option "dist_pack")
			if (tail == 2 && D->pack_ok[xt]) {
				PackMirror m = D->pack_type[xt];
				for (size_t pi = 0; pi < D->peers.size(); ++pi) m.msg[pi] = D->peers[pi].sbuf[D->pending.parity];
				bool done = false;
				HNS_TRY(hns_rbgs_block_pack_launch(D->gB, D->div, s0, d0, vs, omega_compute(vs), zero, &m, s, &done, 2));
				if (done) {
					D->pending.prepacked = true;
					return HNS_OK;
				}
			}
			return part(D->gB, s);
		}, [=](hipStream_t s) { return part(D->gI, s); }));
		std::swap(src, dst);
		it += tail;
		if (last) d->p_result = src;
		return HNS_OK;
	}

	// The full substep. Every kernel boundary a stencil crosses is an exchange (post / complete), whatever the transport: the chained
	// and mirroring forms of the core substep are not used here. Pointwise kernels run over the owned leaves, local [0, nB + nI).
	int run_full(int ph) {
		const float inv_dx = 1.0f / d->voxel_size;
		const int blocks = (iterations + d->k - 1) / d->k;
		typedef std::vector<std::pair<float*, int>> Fields;
		hns_dist* D = d;
		const float* sd = sdf();
		const int cl = coll ? 1 : 0;
		const uint64_t n_owned = (uint64_t)(d->nB + d->nI) * 512u;
		auto nothing = [](hipStream_t) { return HNS_OK; };
		const float dtv = dt;
		if (ph == 0) {  // the advection inputs: phi unless the previous substep already posted it, and u -- which collision rewrites first
			Fields f;
			if (!coll && !d->u_ghosts_fresh) {
				if (d->phi_in_flight) HNS_TRY(complete(d, st));
				f.emplace_back(d->u, 3);
			}
			if (!d->phi_in_flight)
				for (float* p : d->phi) f.emplace_back(p, 1);
			d->phi_in_flight = false;
			if (f.empty()) return HNS_OK;
			return post(d, X_ADV, f, st, nothing);
		}
		{  // (the blocks of the exchanged pressure loop complete what is in flight themselves: sor_block_exchanged)
			const int qb = ph - 1 - (coll ? 1 : 0) - (vort ? 2 : 1) - 2;
			if (!(qb >= 0 && qb < blocks)) HNS_TRY(complete(d, st));
		}
		if (coll && ph == 1) {  // enforceCollisionBoundaries (HNanoSolver.cu:153-157) reads the ghost voxels of the SDF (its normal): they have arrived now
			HNS_TRY(hns_dev_enforce_collision_boundaries(d->gO, d->u, sd, d->voxel_size, st));
			return post(d, X_ADV, Fields{{d->u, 3}}, st, nothing);
		}
		int q = ph - 1 - (coll ? 1 : 0);
		if (q == 0) {  // advect_vector (:162-170); vorticity confinement reads it up to factor_scale + 1 voxels away: whole leaves travel then
			return post(d, vort ? X_ADV : X_D1, Fields{{d->adv, 3}}, st, [=](hipStream_t s) { return hns_dev_advect_vector(D->gB, D->u, D->adv, sd, cl, dtv, inv_dx, s); },
			            [=](hipStream_t s) { return hns_dev_advect_vector(D->gI, D->u, D->adv, sd, cl, dtv, inv_dx, s); });
		}
		if (vort && q == 1) {  // :172-176, out of place (the reference's in-place launch races)
			const float scale = prm->vorticityScale, fs = prm->factorScale;
			HNS_TRY(post(d, X_D1, Fields{{d->tmp, 3}}, st, [=](hipStream_t s) { return hns_dev_vorticity_confinement(D->gB, D->adv, D->tmp, dtv, inv_dx, scale, fs, s); },
			             [=](hipStream_t s) { return hns_dev_vorticity_confinement(D->gI, D->adv, D->tmp, dtv, inv_dx, scale, fs, s); }));
			std::swap(d->adv, d->tmp);
			return HNS_OK;
		}
		q -= vort ? 2 : 1;
		if (q == 0) {  // divergence (:181-188) + what combustion adds to it (:211-221, k_combustion_div: fuel and waste only)
			const float ex = prm->expansionRate;
			const float *fuel = d->phi[(size_t)fi[0]], *waste = d->phi[(size_t)fi[1]];
			const uint64_t nb = (uint64_t)d->nB * 512u, ni = (uint64_t)d->nI * 512u;
			HNS_TRY(post(d, X_DIV, Fields{{d->div, 1}}, st, [=](hipStream_t s) {
				HNS_TRY(hns_dev_divergence(D->gB, D->adv, D->div, inv_dx, s));
				return nb ? hns_combustion_div(fuel, waste, D->div, ex, nb, s) : HNS_OK;
			}));
			HNS_TRY(hns_dev_divergence(d->gI, d->adv, d->div, inv_dx, st));
			return ni ? hns_combustion_div(fuel + nb, waste + nb, d->div + nb, ex, ni, st) : HNS_OK;
		}
		if (q == 1) {  // the rest of combustion, buoyancy with the NEW temperature (:226-234), outputs become inputs (:239-246): pointwise, owned voxels
			if (n_owned) {
				HNS_TRY(hns_combustion_fields(d->phi[(size_t)fi[0]], d->phi[(size_t)fi[1]], d->phi[(size_t)fi[2]], d->phi[(size_t)fi[3]], d->phi_next[(size_t)fi[0]],
				                              d->phi_next[(size_t)fi[1]], d->phi_next[(size_t)fi[2]], d->phi_next[(size_t)fi[3]], prm->temperatureRelease, n_owned, st));
				HNS_TRY(hns_dev_temperature_buoyancy(d->adv, d->phi_next[(size_t)fi[2]], d->adv, dt, prm->ambientTemp, prm->buoyancyStrength, n_owned, st));
			}
			Fields f;
			for (int c = 0; c < 4; ++c) {
				std::swap(d->phi[(size_t)fi[c]], d->phi_next[(size_t)fi[c]]);
				f.emplace_back(d->phi[(size_t)fi[c]], 1);
			}
			return post(d, X_ADV, f, st, nothing);  // advect_scalars reads their ghosts; hidden under the pressure solve
		}
		q -= 2;
		if (q < blocks) return sor_block_exchanged(q);
		q -= blocks;
		if (q == 0) {  // gradient subtraction (:278-289) [and collision, :292-296] -> u, whose ghosts the scalar advection reads
			if (d->timing && d->tev_used + 2 <= d->tev.size()) {
				HNS_HIP(hipEventRecord(d->tev[d->tev_used + 1], st));
				d->tev_used += 2;
				d->timed_sweeps += iterations;
			}
			const float vs = d->voxel_size;
			HNS_TRY(post(d, X_ADV, Fields{{d->u, 3}}, st, [=](hipStream_t s) {
				HNS_TRY(hns_dev_subtract_pressure_gradient(D->gB, D->adv, D->p_result, D->u, sd, cl, inv_dx, s));
				return cl ? hns_dev_enforce_collision_boundaries(D->gB, D->u, sd, vs, s) : HNS_OK;
			}));
			HNS_TRY(hns_dev_subtract_pressure_gradient(d->gI, d->adv, d->p_result, d->u, sd, cl, inv_dx, st));
			return cl ? hns_dev_enforce_collision_boundaries(d->gI, d->u, sd, vs, st) : HNS_OK;
		}
		// advect every float field except collision_sdf with the projected velocity (:321-356), and post them for the next substep
		d->u_ghosts_fresh = !coll;  // (with collision the next substep rewrites u before it advects)
		std::vector<const float*> in;
		std::vector<float*> out;
		std::vector<int> which;
		for (int sidx = 0; sidx < d->n_scalars; ++sidx)
			if (sidx != fi[4]) in.push_back(d->phi[(size_t)sidx]), out.push_back(d->phi_next[(size_t)sidx]), which.push_back(sidx);
		Fields f;
		for (float* p : out) f.emplace_back(p, 1);
		const int ns = (int)in.size();
		if (ns) {
			HNS_TRY(post(d, X_ADV, f, st, [=](hipStream_t s) {
				return D->gB->n_active ? hns_dev_advect_scalars(D->gB, D->u, in.data(), const_cast<float* const*>(out.data()), ns, sd, cl, dtv, inv_dx, s) : HNS_OK;
			}));
			if (d->gI->n_active) HNS_TRY(hns_dev_advect_scalars(d->gI, d->u, in.data(), out.data(), ns, sd, cl, dt, inv_dx, st));
		}
		for (int sidx : which) std::swap(d->phi[(size_t)sidx], d->phi_next[(size_t)sidx]);
		d->phi_in_flight = ns > 0 && d->world > 1;
		return HNS_OK;
	}

	int run(int ph) {
		if (full()) return run_full(ph);
		const float inv_dx = 1.0f / d->voxel_size;
		const int blocks = (iterations + d->k - 1) / d->k;
		typedef std::vector<std::pair<float*, int>> Fields;
		typedef std::vector<std::pair<const float*, int>> Outs;
		hns_dist* D = d;
		auto nothing = [](hipStream_t) { return HNS_OK; };
		if (ph == 0) {  // the advection inputs: phi was posted by the previous substep unless new fields were uploaded
			if (d->phi_in_flight) return HNS_OK;
			Fields f;
			if (!d->u_ghosts_fresh) f.emplace_back(d->u, 3);
			for (float* p : d->phi) f.emplace_back(p, 1);
			if (f.empty()) return HNS_OK;
			return post(d, X_ADV, f, st, nothing);
		}
		if (!(ph >= 3 && ph < 3 + blocks && !d->mirror)) HNS_TRY(complete(d, st));  // (the blocks of the exchanged pressure loop do it themselves: sor_block_exchanged)
		if (ph == 1) {
			if (d->chain) {
				const PhaseMirror m = phase_args(d, X_D1, Outs{{d->adv, 3}});
				return chained(m, [&] {  // gate | the kernel as it is | copy of the boundary leaves' reach-1 voxels into the peers' ghosts
					HNS_TRY(hns_dev_advect_vector(d->gO, d->u, d->adv, nullptr, 0, dt, inv_dx, st));
					if (d->nB && m.n_peers)
						hipLaunchKernelGGL(k_chain_mirror<3>, dim3((unsigned)d->nB), dim3(64), 0, st, m, 1, (const float*)d->adv, (const float*)nullptr, (const float*)nullptr,
						                   (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
					return HNS_OK;
				}, true);
			}
			const float dtv = dt;
			return post(d, X_D1, Fields{{d->adv, 3}}, st, [=](hipStream_t s) { return hns_dev_advect_vector(D->gB, D->u, D->adv, nullptr, 0, dtv, inv_dx, s); },
			            [=](hipStream_t s) { return hns_dev_advect_vector(D->gI, D->u, D->adv, nullptr, 0, dtv, inv_dx, s); });
		}
		if (ph == 2) {
			if (d->chain) {
				const PhaseMirror m = phase_args(d, X_DIV, Outs{{d->div, 1}});
				return chained(m, [&] { return hns_chain_divergence(d->gO, d->adv, d->div, inv_dx, &m, st); });
			}
			if (in_line_rank()) return post(d, X_DIV, Fields{{d->div, 1}}, st, [=](hipStream_t s) { return hns_dev_divergence(D->gO, D->adv, D->div, inv_dx, s); }, nothing, true);
			return post(d, X_DIV, Fields{{d->div, 1}}, st, [=](hipStream_t s) { return hns_dev_divergence(D->gB, D->adv, D->div, inv_dx, s); },
			            [=](hipStream_t s) { return hns_dev_divergence(D->gI, D->adv, D->div, inv_dx, s); });
		}
		if (ph < 3 + blocks) {  // one block of up to k sweeps; all but the last sweep the ghost leaves too
			const int b = ph - 3;
			if (!d->mirror) return sor_block_exchanged(b);
			if (b == 0) {
				it = 0, src = d->p_a, dst = d->p_b;  // never warm-started (reference HNanoSolver.cu:113): the first sweep reads no p
				if (d->timing && d->tev_used + 2 <= d->tev.size()) HNS_HIP(hipEventRecord(d->tev[d->tev_used], st));
			}
			{  // ONE chained launch of the temporally blocked form: two iterations, or the odd one left over (every rank alike: blocked_mirror)
				const int its = std::min(2, iterations - it);
				const PhaseMirror m = phase_args(d, X_P, Outs{{dst, 1}});
				const bool zero = it == 0;
				HNS_TRY(chained(m, [&] { return hns_rbgs_block_mirror_launch(d->gO, d->div, src, dst, d->voxel_size, omega_compute(d->voxel_size), zero, &m, st, its); }));
				std::swap(src, dst);
				it += its;
				if (it == iterations) d->p_result = src;
				return launch_status("hns_dist: blocked mirror sweep");
			}
		}
		if (ph == 3 + blocks) {
			if (d->mirror && !d->chain && d->mir.n_peers) {  // the gradient reads what the peers' last sweep wrote into the ghost voxels
				// (here and not behind the last sweep: locally connected ranks share one stream, and a rank's wait must not sit in
				// front of the sweeps it waits for)
				PhaseMirror m = d->mir;
				m.seq = d->sweep_seq;
				hipLaunchKernelGGL(k_sweep_wait, dim3(1), dim3(64), 0, st, m);
			}
			if (d->timing && d->tev_used + 2 <= d->tev.size()) {  // the timed region ends when the last refresh of p has landed (complete() above)
				HNS_HIP(hipEventRecord(d->tev[d->tev_used + 1], st));
				d->tev_used += 2;
				d->timed_sweeps += iterations;
			}
			if (d->chain) {  // (its boundary workgroups wait for the peers' last sweep themselves)
				const PhaseMirror m = phase_args(d, X_ADV, Outs{{d->u, 3}});
				return chained(m, [&] { return hns_chain_subtract_pressure_gradient(d->gO, d->adv, d->p_result, d->u, inv_dx, &m, st); });
			}
			if (in_line_rank())
				return post(d, X_ADV, Fields{{d->u, 3}}, st, [=](hipStream_t s) { return hns_dev_subtract_pressure_gradient(D->gO, D->adv, D->p_result, D->u, nullptr, 0, inv_dx, s); }, nothing, true);
			return post(d, X_ADV, Fields{{d->u, 3}}, st, [=](hipStream_t s) { return hns_dev_subtract_pressure_gradient(D->gB, D->adv, D->p_result, D->u, nullptr, 0, inv_dx, s); },
			            [=](hipStream_t s) { return hns_dev_subtract_pressure_gradient(D->gI, D->adv, D->p_result, D->u, nullptr, 0, inv_dx, s); });
		}
		// last phase: advect the scalars, and already post them for the advection that opens the next substep
		d->u_ghosts_fresh = true;
		if (d->chain) {
			if (d->n_scalars) {
				Outs outs;
				for (float* p : d->phi_next) outs.emplace_back(p, 1);
				const PhaseMirror m = phase_args(d, X_ADV, outs);
				std::vector<const float*> in(d->phi.begin(), d->phi.end());
				HNS_TRY(chained(m, [&] {
					HNS_TRY(hns_dev_advect_scalars(d->gO, d->u, in.data(), d->phi_next.data(), d->n_scalars, nullptr, 0, dt, inv_dx, st));
					const float* f[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
					for (int s = 0; s < d->n_scalars && s < 8; ++s) f[s] = d->phi_next[(size_t)s];
					if (d->nB && m.n_peers)
						hipLaunchKernelGGL(k_chain_mirror<1>, dim3((unsigned)d->nB), dim3(64), 0, st, m, d->n_scalars, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
					return HNS_OK;
				}, true));
				std::swap(d->phi, d->phi_next);
			}
			d->phi_in_flight = true;  // (here: the peers' ghost copies of phi are already being written, nothing to open the next substep with)
			return HNS_OK;
		}
		Fields f;
		for (float* p : d->phi_next) f.emplace_back(p, 1);  // the boundary leaves' new values travel while the interior is advected
		const Step self = *this;  // (the scalars' arrays as they are now: they are swapped below)
		if (d->n_scalars) HNS_TRY(post(d, X_ADV, f, st, [=](hipStream_t s) { return self.advect_scalars(D->gB, inv_dx, s); }, [=](hipStream_t s) { return self.advect_scalars(D->gI, inv_dx, s); }));
		std::swap(d->phi, d->phi_next);
		d->phi_in_flight = d->n_scalars > 0 && d->world > 1;
		return HNS_OK;
	}
};

int check_step(const hns_dist* d, int iterations, float dt) {
	if (!d) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_core_substep: null handle");
	if (dt < 0.0f) return fail(HNS_ERR_INVALID_ARGUMENT, "dt (time step) cannot be negative.");
	if (iterations <= 0) return fail(HNS_ERR_INVALID_ARGUMENT, "Number of pressure iterations must be positive.");
	return far_check(d);  // (raised by an earlier substep: kernels are asynchronous; hns_dist_synchronize / hns_dist_download report it too)
}

}  // namespace

extern "C" {

// One core substep (advect_vector -> divergence -> iterations x RB-SOR -> gradient subtraction -> advect_scalars) of this
// rank, asynchronous on `stream` (plus the rank's communication stream). RCCL transport, or world == 1.
int hns_dist_core_substep(hns_dist* d, int iterations, float dt, void* stream) {
	HNS_TRY(check_step(d, iterations, dt));
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_core_substep: plan-only handle (there is no CPU fallback)");
	if (!d->local_ranks.empty() && d->world > 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_core_substep: locally connected ranks step together (hns_dist_local_core_substep)");
	if (d->ipc_status && *(volatile int*)d->ipc_status) return fail(HNS_ERR_RUNTIME, "hns_dist: a peer did not answer within 20 s (one-sided transport); results are invalid");
	memset(d->bytes_sent, 0, sizeof(d->bytes_sent));
	d->messages_sent = d->exchanges = d->packed_exchanges = 0;
	Step s{d, iterations, dt, (hipStream_t)stream};
	for (int ph = 0, n = s.n_phases(); ph < n; ++ph) HNS_TRY(s.run(ph));
	return HNS_OK;
}

// The same for ranks connected with hns_dist_connect_local: all ranks advance phase by phase on `stream`.
int hns_dist_local_core_substep(hns_dist* const* ranks, int world, int iterations, float dt, void* stream) {
	if (!ranks || world < 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_local_core_substep: bad arguments");
	std::vector<Step> steps;
	for (int r = 0; r < world; ++r) {
		HNS_TRY(check_step(ranks[r], iterations, dt));
		if ((int)ranks[r]->local_ranks.size() != world) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_local_core_substep: ranks are not locally connected");
		memset(ranks[r]->bytes_sent, 0, sizeof(ranks[r]->bytes_sent));
		ranks[r]->messages_sent = ranks[r]->exchanges = ranks[r]->packed_exchanges = 0;
		steps.push_back(Step{ranks[r], iterations, dt, (hipStream_t)stream});
	}
	// a message may be the sender's field itself (whole-leaf regions are not packed): every rank takes delivery of the previous
	// phase's exchange before any rank's next kernel overwrites what was sent
	for (int ph = 0, n = steps[0].n_phases(); ph < n; ++ph) {
		if (ph > 0)
			for (Step& s : steps) HNS_TRY(complete(s.d, s.st));
		for (Step& s : steps) HNS_TRY(s.run(ph));
	}
	return HNS_OK;
}

// ---- the whole Compute_Sim substep, partitioned (reference HNanoSolver.cu:150-356; single GPU: hns_sim_substep) ----
// field_index: positions of fuel, waste, temperature, flame and collision_sdf (-1: none) among the rank's scalars (hns_dist_upload
// order). Every float field except collision_sdf is advected. Owned results equal hns_sim_substep's on the whole domain bit for bit.
static int sim_step_args(const hns_dist* d, const hns_combustion_params* params, const int* field_index, int has_collision, Step& s) {
	if (!params || !field_index) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: null argument");
	const char* required[4] = {"fuel", "waste", "temperature", "flame"};
	for (int c = 0; c < 4; ++c) {
		if (field_index[c] < 0 || field_index[c] >= d->n_scalars) {
			set_error("Missing required input field for combustion: %s", required[c]);  // HNanoSolver.cu:193-201
			return HNS_ERR_RUNTIME;
		}
		for (int e = 0; e < c; ++e)
			if (field_index[e] == field_index[c]) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: two combustion fields share one scalar");
	}
	if (field_index[4] >= d->n_scalars) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: collision_sdf index out of range");
	for (int c = 0; c < 4; ++c)
		if (field_index[4] >= 0 && field_index[4] == field_index[c]) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: collision_sdf shares a scalar with a combustion field");
	// vorticity confinement reads u* up to (int)factor_scale + 1 voxels from a voxel: it must stay inside the one-leaf ghost layer
	if ((int)params->factorScale > 6 || (int)params->factorScale < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: factor_scale must be within 0..6 on a partitioned domain");
	s.prm = params;
	for (int c = 0; c < 5; ++c) s.fi[c] = field_index[c];
	s.coll = has_collision && field_index[4] >= 0;  // HNanoSolver.cu:66-75 (collision_sdf itself is never advected, used or not: :327)
	s.vort = (int)params->factorScale != 0;  // (int)factor_scale == 0: the kernel is a bit-exact copy (hns_api.hip: Substep::part_a)
	return HNS_OK;
}

int hns_dist_sim_substep(hns_dist* d, int iterations, float dt, const hns_combustion_params* params, const int* field_index, int has_collision, void* stream) {
	HNS_TRY(check_step(d, iterations, dt));
	if (!d->gA) return fail(HNS_ERR_NO_DEVICE, "hns_dist_sim_substep: plan-only handle (there is no CPU fallback)");
	if (!d->local_ranks.empty() && d->world > 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_sim_substep: locally connected ranks step together (hns_dist_local_sim_substep)");
	if (d->ipc_status && *(volatile int*)d->ipc_status) return fail(HNS_ERR_RUNTIME, "hns_dist: a peer did not answer within 20 s (one-sided transport); results are invalid");
	memset(d->bytes_sent, 0, sizeof(d->bytes_sent));
	d->messages_sent = d->exchanges = d->packed_exchanges = 0;
	Step s{d, iterations, dt, (hipStream_t)stream};
	HNS_TRY(sim_step_args(d, params, field_index, has_collision, s));
	for (int ph = 0, n = s.n_phases(); ph < n; ++ph) HNS_TRY(s.run(ph));
	return HNS_OK;
}

int hns_dist_local_sim_substep(hns_dist* const* ranks, int world, int iterations, float dt, const hns_combustion_params* params, const int* field_index, int has_collision,
                               void* stream) {
	if (!ranks || world < 1) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_local_sim_substep: bad arguments");
	std::vector<Step> steps;
	for (int r = 0; r < world; ++r) {
		HNS_TRY(check_step(ranks[r], iterations, dt));
		if ((int)ranks[r]->local_ranks.size() != world) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_local_sim_substep: ranks are not locally connected");
		memset(ranks[r]->bytes_sent, 0, sizeof(ranks[r]->bytes_sent));
		ranks[r]->messages_sent = ranks[r]->exchanges = ranks[r]->packed_exchanges = 0;
		steps.push_back(Step{ranks[r], iterations, dt, (hipStream_t)stream});
		HNS_TRY(sim_step_args(ranks[r], params, field_index, has_collision, steps.back()));
	}
	for (int ph = 0, n = steps[0].n_phases(); ph < n; ++ph) {
		// (before phase 0 too: with collision the phase rewrites u and posts it again, and a rank must have taken delivery of the
		// exchange the previous substep left in flight before a peer posts the next one)
		for (Step& s : steps) HNS_TRY(complete(s.d, s.st));
		for (Step& s : steps) HNS_TRY(s.run(ph));
	}
	return HNS_OK;
}

int hns_dist_timing(hns_dist* d, int max_solves) {
	if (!d || max_solves < 0) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_timing: bad arguments");
	while (d->tev.size() < (size_t)max_solves * 2) {
		hipEvent_t e;
		HNS_HIP(hipEventCreate(&e));
		d->tev.push_back(e);
	}
	d->timing = max_solves > 0;
	d->tev_used = 0;
	d->timed_sweeps = 0;
	return HNS_OK;
}

int hns_dist_pressure_time(hns_dist* d, float* total_ms, long long* sweeps) {
	if (!d || !total_ms || !sweeps) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_pressure_time: null argument");
	double tot = 0.0;
	for (size_t i = 0; i + 1 < d->tev_used; i += 2) {
		HNS_HIP(hipEventSynchronize(d->tev[i + 1]));
		float ms = 0.0f;
		HNS_HIP(hipEventElapsedTime(&ms, d->tev[i], d->tev[i + 1]));
		tot += ms;
	}
	*total_ms = (float)tot;
	*sweeps = d->timed_sweeps;
	return HNS_OK;
}

int hns_dist_synchronize(hns_dist* d, void* stream) {
	if (!d) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dist_synchronize: null handle");
	HNS_HIP(hipStreamSynchronize((hipStream_t)stream));
	if (d->cs) HNS_HIP(hipStreamSynchronize(d->cs));
	if (d->ipc_status && *(volatile int*)d->ipc_status) return fail(HNS_ERR_RUNTIME, "hns_dist: a peer did not answer within 20 s (one-sided transport); results are invalid");
	return far_check(d);
}

}  // extern "C"
