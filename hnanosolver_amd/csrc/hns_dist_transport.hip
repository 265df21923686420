// hns_dist_transport.hip -- multi-GPU: RCCL bound at run time, the loopback stand-ins, hipIpc-mapped peers, locally connected ranks; flags and tables of the chained substep (see hns_dist.hpp)
#include "hns_dist.hpp"

using namespace hns;
using namespace hnsd;

namespace hnsd {
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

// ---- flags and the chained substep (ipc and local transports) ----
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

// The communication stream exists only where a second stream is used: RCCL and loopback transports. It outranks the compute
// stream: its short kernels (boundary leaves, pack, unpack) must not queue behind the thousands of waves of the interior
// kernel they run next to.
int ensure_comm_stream(hns_dist* d) {
	if (d->cs) return HNS_OK;
	int lo_prio = 0, hi_prio = 0;
	(void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
	HNS_HIP(hipStreamCreateWithPriority(&d->cs, hipStreamNonBlocking, hi_prio));
	d->cs_owner = std::shared_ptr<void>((void*)d->cs, [](void* s) { (void)hipStreamDestroy((hipStream_t)s); });
	return HNS_OK;
}


}  // namespace hnsd

extern "C" {

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


}  // extern "C"
