// hns_pointwise.hip -- element-wise and small-stencil side kernels of Compute_Sim (combustion, buoyancy, vorticity
// confinement, collision) and the leaf pack/unpack used by the halo exchange.
#include <cstdlib>
#include <cstring>

#include "hns_device.hpp"
#include "hns_digest.hpp"

namespace hns {

// ---------------------------------------------------------------------------------------------------------------
// element-wise kernels
// ---------------------------------------------------------------------------------------------------------------

// combustion_oxygen (reference Kernel.cu:923-966)
__global__ __launch_bounds__(256) void k_combustion_oxygen(const float* __restrict__ fuel_in, const float* __restrict__ waste_in, const float* __restrict__ temp_in,
                                                           float* __restrict__ div, const float* __restrict__ flame_in, float* __restrict__ fuel_out,
                                                           float* __restrict__ waste_out, float* __restrict__ temp_out, float* __restrict__ flame_out,
                                                           const float temp_gain, const float expansion, const uint64_t n, const int update_div) {
	for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
		float f = fuel_in[v];
		const float w = waste_in[v], tmp = temp_in[v], fl = flame_in[v];
		if (f < 0.001f) f = 0.0f;       // Kernel.cu:936-938
		const float oxy = 1.0f - f - w;  // :941
		if (oxy < 0.0f) {                // :942-949: no oxygen left, the voxel passes through
			fuel_out[v] = f, waste_out[v] = w, temp_out[v] = tmp, flame_out[v] = fl;
			continue;
		}
		const float burn = fminf(oxy, f);  // :952
		fuel_out[v] = f - burn;
		waste_out[v] = w + burn * 2.0f;
		temp_out[v] = tmp + burn * temp_gain;
		if (update_div) div[v] += burn * expansion;
		flame_out[v] = fmaxf(fl, fminf(1.0f, burn * 10.0f));
	}
}

// The divergence update of combustion_oxygen alone. It depends on fuel and waste only, and it is all the pressure solve
// waits for: the cook pipeline (hns_api.hip) runs it as soon as those two fields are on the device and lets the solve
// overlap the upload of the others; k_combustion_oxygen then runs with update_div = 0. Same expressions, same result.
__global__ __launch_bounds__(256) void k_combustion_div(const float* __restrict__ fuelData, const float* __restrict__ wasteData,
                                                        float* __restrict__ divergenceData, const float expansion, const uint64_t n) {
	for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
		float fuel = fuelData[idx];
		const float waste = wasteData[idx];
		if (fuel < 0.001f) fuel = 0.0f;
		const float oxygen = 1.0f - fuel - waste;
		if (oxygen < 0.0f) continue;
		const float burn = fminf(oxygen, fuel);
		divergenceData[idx] += burn * expansion;
	}
}

// temperature_buoyancy (reference Kernel.cu:831-847): vel + (0, max(0,(T-Tamb)*k), 0) * dt. The x and z results are
// vel + 0*dt; they are passed through the same add so that the stored bits equal the reference's (-0 + 0 = +0).
__global__ __launch_bounds__(256) void k_temperature_buoyancy(const float* u, const float* __restrict__ temp, float* out, const float dt,
                                                              const float ambient, const float strength, const uint64_t n) {
	for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
		const f3 v = ld3(u, (int)idx);
		const float t = temp[idx];
		if (t <= ambient) {
			st3(out, (int)idx, v);
			continue;
		}
		const float tempDiff = t - ambient;
		const f3 r = {v.x + dt * 0.0f, v.y + dt * fmaxf(0.0f, tempDiff * strength), v.z + dt * 0.0f};
		st3(out, (int)idx, r);
	}
}

// whole-leaf gather/scatter for the halo exchange: a leaf payload is 512*ncomp floats (ncomp 1 = float field, 3 = Vec3f
// field), moved as float4 by 128 threads
__global__ __launch_bounds__(128) void k_pack_leaves(const float* __restrict__ field, const int* __restrict__ ids, float* __restrict__ packed,
                                                     const int ncomp) {
	const int l = ids[blockIdx.x];
	const float4* src = reinterpret_cast<const float4*>(field + (size_t)l * 512 * ncomp);
	float4* dst = reinterpret_cast<float4*>(packed + (size_t)blockIdx.x * 512 * ncomp);
	for (int i = threadIdx.x; i < 128 * ncomp; i += 128) dst[i] = src[i];
}
__global__ __launch_bounds__(128) void k_unpack_leaves(const float* __restrict__ packed, const int* __restrict__ ids, float* __restrict__ field,
                                                       const int ncomp) {
	const int l = ids[blockIdx.x];
	const float4* src = reinterpret_cast<const float4*>(packed + (size_t)blockIdx.x * 512 * ncomp);
	float4* dst = reinterpret_cast<float4*>(field + (size_t)l * 512 * ncomp);
	for (int i = threadIdx.x; i < 128 * ncomp; i += 128) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------------------
// vorticityConfinement (reference Kernel.cu:970-1024 + Utils.cuh:226-243), out of place
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ f3 curl_at(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ u, int i, int j, int k,
                                      float factor) {
	const f3 pX = ld3z(u, tap_index(g, s_nbr, org, i + 1, j, k)), mX = ld3z(u, tap_index(g, s_nbr, org, i - 1, j, k));
	const f3 pY = ld3z(u, tap_index(g, s_nbr, org, i, j + 1, k)), mY = ld3z(u, tap_index(g, s_nbr, org, i, j - 1, k));
	const f3 pZ = ld3z(u, tap_index(g, s_nbr, org, i, j, k + 1)), mZ = ld3z(u, tap_index(g, s_nbr, org, i, j, k - 1));
	f3 w;
	w.x = ((pY.z - mY.z) - (pZ.y - mZ.y)) * factor;
	w.y = ((pZ.x - mZ.x) - (pX.z - mX.z)) * factor;
	w.z = ((pX.y - mX.y) - (pY.x - mY.x)) * factor;
	return w;
}

__device__ __forceinline__ float curl_mag(const GridDev& g, const int* s_nbr, const int4 org, const float* __restrict__ u, int i, int j, int k,
                                          float factor) {
	const f3 w = curl_at(g, s_nbr, org, u, i, j, k, factor);
	return sqrtf(w.x * w.x + w.y * w.y + w.z * w.z);
}

__global__ __launch_bounds__(512) void k_vorticity(const GridDev g, const float* __restrict__ u, float* __restrict__ out, const float dt,
                                                   const float inv_dx, const float scale, const int fs) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
	const float factor = 0.5f * inv_dx;
	const f3 w = curl_at(g, s_nbr, L.org, u, ci, cj, ck, factor);
	const float m_pX = curl_mag(g, s_nbr, L.org, u, ci + fs, cj, ck, factor), m_mX = curl_mag(g, s_nbr, L.org, u, ci - fs, cj, ck, factor);
	const float m_pY = curl_mag(g, s_nbr, L.org, u, ci, cj + fs, ck, factor), m_mY = curl_mag(g, s_nbr, L.org, u, ci, cj - fs, ck, factor);
	const float m_pZ = curl_mag(g, s_nbr, L.org, u, ci, cj, ck + fs, factor), m_mZ = curl_mag(g, s_nbr, L.org, u, ci, cj, ck - fs, factor);
	const float grad_x = (m_pX - m_mX) * 0.5f * inv_dx;
	const float grad_y = (m_pY - m_mY) * 0.5f * inv_dx;
	const float grad_z = (m_pZ - m_mZ) * 0.5f * inv_dx;
	const float gradLen = sqrtf(grad_x * grad_x + grad_y * grad_y + grad_z * grad_z) + 1e-5f;
	const float Nx = grad_x / gradLen, Ny = grad_y / gradLen, Nz = grad_z / gradLen;
	const f3 v = ld3(u, idx);
	const f3 r = {v.x + dt * (scale * (Ny * w.z - Nz * w.y)), v.y + dt * (scale * (Nz * w.x - Nx * w.z)), v.z + dt * (scale * (Nx * w.y - Ny * w.x))};
	st3(out, idx, r);
}

// enforceCollisionBoundaries (reference Kernel.cu:77-116), in place
__global__ __launch_bounds__(512) void k_enforce_collision(const GridDev g, float* u, const float* __restrict__ sdf, const float inv_dx) {
	__shared__ int s_nbr[27];
	const LeafCtx L = stage_leaf(g, s_nbr, blockIdx.x);
	const int n = threadIdx.x;
	const int idx = L.leaf * 512 + n;
	const float sv = sdf[idx];
	if (sv < 0.0f) {
		const f3 z = {0.0f, 0.0f, 0.0f};
		st3(u, idx, z);
		return;
	}
	const float margin = 0.1f;
	if (sv < margin) {
		const int ci = L.org.x + (n >> 6), cj = L.org.y + ((n >> 3) & 7), ck = L.org.z + (n & 7);
		const f3 nrm = sdf_normal(g, s_nbr, L.org, sdf, ci, cj, ck, inv_dx);
		st3(u, idx, no_slip_blend(ld3(u, idx), nrm, 1.0f - (sv / margin)));
	}
}

}  // namespace hns

using namespace hns;

extern "C" {



int hns_dev_combustion_oxygen(const float* fuel, const float* waste, const float* temperature, float* divergence, const float* flame,
                              float* out_fuel, float* out_waste, float* out_temperature, float* out_flame, float temp_gain, float expansion,
                              uint64_t n, void* stream) {
	NULLCHK(!fuel || !waste || !temperature || !divergence || !flame || !out_fuel || !out_waste || !out_temperature || !out_flame,
	        "hns_dev_combustion_oxygen");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_combustion_oxygen, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, fuel, waste, temperature, divergence, flame,
	                   out_fuel, out_waste, out_temperature, out_flame, temp_gain, expansion, n, 1);
	return launch_status("hns_dev_combustion_oxygen");
}

// the two halves of combustion_oxygen for the cook pipeline (see k_combustion_div)
// ---- field digest (hns_compute_sim_resident, CHECKED fields; hns_digest.hpp) ----
// sum over the 16-byte pieces of a field of hns_digest_piece(piece number, its two 8-byte words), mod 2^64: order-independent, so the device takes it of the
// buffer a host array was downloaded from in one pass at memory speed, and the host takes the same number of the array on as many threads as it likes
__global__ __launch_bounds__(256) void k_field_digest(const uint4* __restrict__ a, const uint64_t n_pieces, const float* __restrict__ tail, const int n_tail, unsigned long long* __restrict__ out) {
	unsigned long long sum = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pieces; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint4 v = a[i];
		sum += hns_digest_piece(i, (uint64_t)v.x | (uint64_t)v.y << 32, (uint64_t)v.z | (uint64_t)v.w << 32);
	}
	if (n_tail && blockIdx.x == 0 && threadIdx.x == 0) {  // (a field whose length is not a multiple of four floats: zero-padded last piece)
		uint32_t w[4] = {0, 0, 0, 0};
		for (int k = 0; k < n_tail; ++k) w[k] = __float_as_uint(tail[k]);
		sum += hns_digest_piece(n_pieces, (uint64_t)w[0] | (uint64_t)w[1] << 32, (uint64_t)w[2] | (uint64_t)w[3] << 32);
	}
	for (int d = 32; d; d >>= 1) sum += __shfl_down(sum, d, 64);
	__shared__ unsigned long long s_part[4];
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

int hns_field_digest(const float* field, uint64_t count, unsigned long long* d_out, void* stream) {
	const uint64_t n_pieces = count / 4;
	const unsigned blocks = (unsigned)std::min<uint64_t>(2048, std::max<uint64_t>(1, (n_pieces + 255) / 256));
	hipLaunchKernelGGL(k_field_digest, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(field), n_pieces, field + 4 * n_pieces, (int)(count & 3), d_out);
	return launch_status("hns_field_digest");
}

int hns_combustion_div(const float* fuel, const float* waste, float* divergence, float expansion, uint64_t n, void* stream) {
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_combustion_div, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, fuel, waste, divergence, expansion, n);
	return launch_status("hns_combustion_div");
}
int hns_combustion_fields(const float* fuel, const float* waste, const float* temperature, const float* flame, float* out_fuel, float* out_waste,
                          float* out_temperature, float* out_flame, float temp_gain, uint64_t n, void* stream) {
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_combustion_oxygen, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, fuel, waste, temperature, (float*)nullptr, flame,
	                   out_fuel, out_waste, out_temperature, out_flame, temp_gain, 0.0f, n, 0);
	return launch_status("hns_combustion_fields");
}

int hns_dev_temperature_buoyancy(const float* vel3, const float* temperature, float* out3, float dt, float ambient, float strength, uint64_t n,
                                 void* stream) {
	NULLCHK(!vel3 || !temperature || !out3, "hns_dev_temperature_buoyancy");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_temperature_buoyancy, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, vel3, temperature, out3, dt, ambient, strength,
	                   n);
	return launch_status("hns_dev_temperature_buoyancy");
}

int hns_dev_vorticity_confinement(hns_grid* g, const float* vel3, float* out3, float dt, float inv_dx, float confinement_scale, float factor_scale,
                                  void* stream) {
	if (int rc = check_grid(g, "hns_dev_vorticity_confinement")) return rc;
	NULLCHK(!vel3 || !out3, "hns_dev_vorticity_confinement");
	if (vel3 == out3) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_vorticity_confinement: output must not alias input");
	if (g->n_active == 0) return HNS_OK;
	const int fs = (int)factor_scale;  // nanovdb::Coord(factorScale,0,0) truncates (Kernel.cu:998)
	hipLaunchKernelGGL(k_vorticity, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), vel3, out3, dt, inv_dx, confinement_scale,
	                   fs);
	return launch_status("hns_dev_vorticity_confinement");
}

int hns_dev_enforce_collision_boundaries(hns_grid* g, float* vel3, const float* sdf, float voxel_size, void* stream) {
	if (int rc = check_grid(g, "hns_dev_enforce_collision_boundaries")) return rc;
	NULLCHK(!vel3, "hns_dev_enforce_collision_boundaries");
	if (!sdf || g->n_active == 0) return HNS_OK;  // Kernel.cu:83
	hipLaunchKernelGGL(k_enforce_collision, dim3((unsigned)g->n_active), dim3(512), 0, (hipStream_t)stream, g->dev(), vel3, sdf, 1.0f / voxel_size);
	return launch_status("hns_dev_enforce_collision_boundaries");
}

int hns_dev_pack_leaves(const float* field, const int32_t* leaf_ids, uint64_t n, float* packed, int ncomp, void* stream) {
	NULLCHK((!field || !leaf_ids || !packed) && n, "hns_dev_pack_leaves");
	if (ncomp != 1 && ncomp != 3) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_pack_leaves: ncomp must be 1 or 3");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_pack_leaves, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, field, leaf_ids, packed, ncomp);
	return launch_status("hns_dev_pack_leaves");
}

int hns_dev_unpack_leaves(const float* packed, const int32_t* leaf_ids, uint64_t n, float* field, int ncomp, void* stream) {
	NULLCHK((!field || !leaf_ids || !packed) && n, "hns_dev_unpack_leaves");
	if (ncomp != 1 && ncomp != 3) return fail(HNS_ERR_INVALID_ARGUMENT, "hns_dev_unpack_leaves: ncomp must be 1 or 3");
	if (n == 0) return HNS_OK;
	hipLaunchKernelGGL(k_unpack_leaves, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream, packed, leaf_ids, field, ncomp);
	return launch_status("hns_dev_unpack_leaves");
}

}  // extern "C"
