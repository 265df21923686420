// hns_dist_plan.hip -- multi-GPU: which leaves a rank owns, its halo regions and tables; create / destroy, plan queries, upload / download (see hns_dist.hpp)
#include "hns_dist.hpp"

using namespace hns;
using namespace hnsd;

namespace hnsd {

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

}  // namespace hnsd

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
