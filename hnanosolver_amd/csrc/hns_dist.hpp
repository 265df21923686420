// hns_dist.hpp -- the leaf-partitioned multi-GPU core substep (the reference is single-GPU, so this is new
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
//
// Round 6: three translation units around this header --
//   hns_dist_plan.hip       which leaves a rank owns, its halo regions and tables; create / destroy, plan queries, upload / download
//   hns_dist_transport.hip  RCCL (bound at run time), the loopback stand-ins, hipIpc-mapped peers, locally connected ranks; the flags and tables of the chained substep
//   hns_dist_substep.hip    the exchange (post / complete), its kernels, and the core / full substep as a sequence of phases
#pragma once
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
// ships its own librccl.so.1 -- keeps exactly one copy instead of two interposing each other. (hns_dist_transport.hip)
namespace hnsd {
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
Rccl& rccl();
int need_rccl(const char* who);
}  // namespace hnsd

#define HNS_NCCL(call)                                                                                         \
	do {                                                                                                       \
		ncclResult_t r__ = (call);                                                                             \
		if (r__ != ncclSuccess) {                                                                              \
			hns::set_error("%s failed: %s (%s:%d)", #call, hnsd::rccl().GetErrorString(r__), __FILE__, __LINE__);    \
			return HNS_ERR_HIP;                                                                                \
		}                                                                                                      \
	} while (0)

namespace hnsd {

constexpr int kMaxBatchPeers = 16;  // peers whose regions one pack / unpack launch serves (k_halo_copy_all)
constexpr int kIpcMaxSegs = 32, kIpcMaxPeers = hns::kMirrorMaxPeers;  // one-sided transport: segments per put launch, peers a rank can map

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
inline Mask512 reach_mask(int j, int D) {
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

}  // namespace hnsd

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
	std::vector<hnsd::Peer> peers;
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
	hnsd::Pending pending;
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
	// boundary values into the peers' ghost voxels (hns_sorblock.hip: k_rbgs_block_xy<., PhaseMirror>)
	bool mirror = false;
	void* mir_tables = nullptr;
	hns::PhaseMirror mir;            // everything but the region tables, the output arrays and the launch number
	struct MirTables {          // per halo region type (X_ADV, X_D1, X_DIV, X_P)
		const int* first = nullptr;
		const int2* entry = nullptr;
		const unsigned char* mask = nullptr;  // null: whole leaves
	} mir_type[4];
	// the blocked boundary sweep of the exchanged pressure loop packs its own messages (hns_flags.hpp: PackMirror): tables per region type (X_D1, X_P), one allocation
	void* pack_tables = nullptr;
	hns::PackMirror pack_type[4];
	bool pack_ok[4] = {false, false, false, false};
	bool chain = false;         // ... and every other kernel of the substep delivers its own halo too (no communication stream at all)
	uint32_t sweep_seq = 0;
	// statistics of the last substep
	uint64_t bytes_sent[hnsd::X_COUNT] = {0, 0, 0, 0}, messages_sent = 0, exchanges = 0, packed_exchanges = 0;
	// hipEvent bracketing of the pressure loop (communication included)
	bool timing = false;
	std::vector<hipEvent_t> tev;
	size_t tev_used = 0;
	long long timed_sweeps = 0;
};

namespace hnsd {

inline int far_check(const hns_dist* d) {
	if (d->far_status && *(volatile int*)d->far_status)
		return hns::fail(HNS_ERR_RUNTIME, "hns_dist: an advection back-trace reached beyond the one-leaf ghost layer of this rank (|u| dt / dx above ~8 voxels at a partition "
		                             "boundary): the owned result can differ from the single-domain run. Use a smaller time step (or fewer ranks); upload the fields again to clear this.");
	return HNS_OK;
}

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }


// Round 4: sweeps_per_exchange = 2 mirrors as well -- its sweeps are the temporally blocked form, two iterations per chained launch
// (hns_sorblock.hip: k_rbgs_block<2, 2, ., true, PhaseMirror>), whose mirror region is the plan's reach-4 region of p. All ranks must
// take the same path (they count launches alike), so the decision uses only what every rank knows: the smallest owned range must be
// swept in 16^3 blocks (more than 600 leaves, hns_rbgs_block_shape) and the option must say so.
// Does a rank of `world` ranks over `n_global` leaves take the chained blocked sweep (two iterations per chained launch) when it runs with
// sweeps_per_exchange = `k` over the ipc / local transports? From what every rank knows alike (all ranks must count launches alike) and the options
// as they are NOW: hns_dist_create asks once and stores the answer (hns_dist::blocked), which is what create, connect and the substep use -- an option
// changed between create and connect no longer leaves a rank on two paths at once.
inline bool blocked_mirror_rule(int k, int world, int64_t n_global) {
	return k == 2 && hns::options().sor_block_lb.load() != 1 && world > 0 && n_global / world > 600 && n_global <= 2000000;
}
inline bool blocked_mirror(const hns_dist* d) { return d->blocked; }
// (rounds 2-5 also chained ranks of one-leaf blocks, sweeps_per_exchange = 1, through a mirroring one-iteration kernel; such ranks -- 600 leaves and fewer -- take the exchanged path now)
inline bool mirror_wanted(const hns_dist* d) { return blocked_mirror(d) && d->world > 1 && hns::options().dist_mirror.load() != 0; }

// hns_dist_plan.hip
int partition_order(const int32_t* origins, int64_t n, int world, bool leaf_order, std::vector<int64_t>& order);
int build_plan(hns_dist* d, const int32_t* origins, int64_t n, int world, int rank, int64_t g0);
// hns_dist_transport.hip
int ensure_flags(hns_dist* d);
int ensure_comm_stream(hns_dist* d);

}  // namespace hnsd
