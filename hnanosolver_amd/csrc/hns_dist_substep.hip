// hns_dist_substep.hip -- multi-GPU: the exchange (post / complete) and its kernels, and the core / full substep of a rank as a sequence of phases (see hns_dist.hpp)
#include "hns_dist.hpp"

namespace hns {
using hnsd::kMaxBatchPeers;

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
using hnsd::kIpcMaxSegs;
using hnsd::kIpcMaxPeers;

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
using namespace hnsd;

// ---------------------------------------------------------------------------------------------------------------
// exchange: post (pack, hand to the communication stream) / complete (wait, unpack)
// ---------------------------------------------------------------------------------------------------------------

namespace {

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
		const int u = options().dist_unsplit.load();  // 0 never | 1 by size | 2 always
		return !d->single_stream && (d->comm || d->loopback || d->ipc) && u != 0 && (u == 2 || d->nB + d->nI <= 16384);
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
			// two iterations in one blocked launch: the boundary sweep writes the peers' messages as it stores (PackMirror)
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
