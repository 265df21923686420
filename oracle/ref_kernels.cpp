// ref_kernels.cpp -- launch driver for the REFERENCE's own kernel bodies (src/Cuda/Kernel.cu), run on the host.
//
// TEST INFRASTRUCTURE ONLY. This file contains no reference code. oracle/Makefile compiles the reference's
// src/Cuda/Kernel.cu WHERE IT LIES into oracle/_ref/Kernel.o (git-ignored) with g++ and links it with this driver
// and ref_samplers.cpp into oracle/_ref/libhns_ref.so. What that build uses, and nothing else:
//   * the reference's own files: src/Cuda/Kernel.cu, Kernels.cuh, Utils.cuh, src/Utils/Stencils.hpp, vendored NanoVDB;
//   * NVIDIA's own CUDA runtime headers, which THIS IMAGE ships (the copy bundled with the triton wheel:
//     <site-packages>/triton/backends/nvidia/include/{cuda_runtime.h, device_launch_parameters.h, ...}); located at
//     build time, never copied, no header of ours stands in for them;
//   * command-line macros for what only nvcc provides: the include guard of nanovdb/tools/cuda/PointsToGrid.cuh
//     (Kernel.cu:5 includes it, no kernel uses it, it needs CUB), and the device intrinsics as the IEEE operations they
//     are defined to be: __fmaf_rn=fmaf, __fmul_rn(a,b)=a*b, __fsub_rn(a,b)=a-b, __fsqrt_rn=sqrtf,
//     __float2int_rd(x)=(int)floorf(x) (with -ffp-contract=off nothing else fuses), __shared__=static,
//     __syncthreads() = a call of refk_barrier() below.
// A __global__ function compiled this way is an ordinary function that reads the launch coordinates from the
// globals device_launch_parameters.h declares (threadIdx, blockIdx, blockDim, gridDim). This driver defines those
// four objects and is the launch: it sets them and calls the kernel once per thread, blocks and threads in order.
//   * 1-D kernels (the Compute_Sim path, HNanoSolver.cu:150-356: blockSize 256): a plain double loop.
//   * the 8x8x8 "_opt" kernels (PressureProjection.cu:44-66) stage a tile in __shared__ memory and meet at
//     __syncthreads(): the 512 threads of a block run as 512 ucontext fibers on one OS thread that share the
//     (static) tile; a barrier hands control back to the scheduler, which resumes every fiber in turn.
// The kernels live in another translation unit and are reached through function pointers, so every call reloads
// the launch coordinates (Kernel.cu's side declares them const).
//
// What it pins: every live kernel body of Kernel.cu, bit for bit, for oracle/hns_oracle.c (tests/test_ref_kernels.py)
// and, through tests/golden/ref_kernels_v1.npz (generator committed), for the HIP kernels on the GPU box.
// What it does not pin: nvcc's own instruction selection (FMA contraction).
#include <ucontext.h>

#include <cstdint>
#include <cstdlib>
#include <vector>

#include "Cuda/Kernels.cuh"  // the reference's declarations, where they lie

// ---- the launch coordinates (declared `extern const` by NVIDIA's device_launch_parameters.h on Kernel.cu's side) ----
uint3 threadIdx;
uint3 blockIdx;
dim3 blockDim;
dim3 gridDim;

namespace {

using Grid = nanovdb::NanoGrid<nanovdb::ValueOnIndex>;
using nanovdb::Coord;
using nanovdb::Vec3f;

template <class Call>
void launch_1d(size_t n, Call&& call) {
	constexpr unsigned kBlock = 256;  // HNanoSolver.cu:141, PressureProjection.cu:110, Advection.cu:70
	blockDim = dim3(kBlock, 1, 1);
	gridDim = dim3((unsigned)((n + kBlock - 1) / kBlock), 1, 1);
	threadIdx = uint3{0, 0, 0};
	blockIdx = uint3{0, 0, 0};
	for (unsigned b = 0; b < gridDim.x; ++b) {
		blockIdx.x = b;
		for (unsigned t = 0; t < kBlock; ++t) {
			threadIdx.x = t;
			call();
		}
	}
}

// ---- 512 fibers per block for the kernels that use __shared__ + __syncthreads ----
struct Fiber {
	ucontext_t ctx;
	uint3 tid;
	bool done;
};
ucontext_t g_sched;
Fiber* g_cur = nullptr;
void (*g_body)(void*) = nullptr;
void* g_body_arg = nullptr;

void fiber_main() {
	g_body(g_body_arg);
	g_cur->done = true;
	swapcontext(&g_cur->ctx, &g_sched);
}

template <class Call>
void launch_leaf(unsigned n_leaves, Call&& call) {
	constexpr size_t kStack = 64 << 10;
	static std::vector<char> stacks(512 * kStack);
	static std::vector<Fiber> fibers(512);
	blockDim = dim3(8, 8, 8);  // PressureProjection.cu:44
	gridDim = dim3(n_leaves, 1, 1);
	g_body = [](void* p) { (*static_cast<Call*>(p))(); };
	g_body_arg = &call;
	for (unsigned b = 0; b < n_leaves; ++b) {
		blockIdx = uint3{b, 0, 0};
		for (unsigned t = 0; t < 512; ++t) {
			Fiber& f = fibers[t];
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = stacks.data() + t * kStack;
			f.ctx.uc_stack.ss_size = kStack;
			f.ctx.uc_link = nullptr;
			makecontext(&f.ctx, fiber_main, 0);
			f.tid = uint3{t & 7u, (t >> 3) & 7u, t >> 6};
			f.done = false;
		}
		unsigned running = 512;
		while (running) {  // one pass = "every thread up to its next barrier"
			for (unsigned t = 0; t < 512; ++t) {
				Fiber& f = fibers[t];
				if (f.done) continue;
				g_cur = &f;
				threadIdx = f.tid;
				swapcontext(&g_sched, &f.ctx);
				if (f.done) --running;
			}
		}
	}
}

}  // namespace

// What -D'__syncthreads()=...' expands to inside Kernel.cu (C++ linkage on purpose: declared there at block scope).
void refk_barrier() { swapcontext(&g_cur->ctx, &g_sched); }

extern "C" {

// `grid` = the NanoGrid<ValueOnIndex>* of ref_samplers.cpp's ref_grid_nanogrid(); coords = N x 3 int32 in value order.

void refk_advect_vector(const void* grid, const int32_t* coords, const float* vel3, float* out3, const float* sdf, int has_collision,
                        uint64_t n, float dt, float inv_dx) {
	auto* k = &advect_vector;
	launch_1d(n, [&] {
		k((const Grid*)grid, (const Coord*)coords, (const Vec3f*)vel3, (Vec3f*)out3, sdf, has_collision != 0, n, dt, inv_dx);
	});
}

void refk_advect_scalar(const void* grid, const int32_t* coords, const float* vel3, const float* in, float* out, const float* sdf,
                        int has_collision, uint64_t n, float dt, float inv_dx) {
	auto* k = &advect_scalar;
	launch_1d(n, [&] { k((const Grid*)grid, (const Coord*)coords, (const Vec3f*)vel3, in, out, sdf, has_collision != 0, n, dt, inv_dx); });
}

void refk_advect_scalars(const void* grid, const int32_t* coords, const float* vel3, float** in, float** out, int n_scalars,
                         const float* sdf, int has_collision, uint64_t n, float dt, float inv_dx) {
	auto* k = &advect_scalars;
	launch_1d(n, [&] {
		k((const Grid*)grid, (const Coord*)coords, (const Vec3f*)vel3, in, out, n_scalars, sdf, has_collision != 0, n, dt, inv_dx);
	});
}

void refk_divergence(const void* grid, const int32_t* coords, const float* vel3, float* out_div, float inv_dx, uint64_t n) {
	auto* k = &divergence;
	launch_1d(n, [&] { k((const Grid*)grid, (const Coord*)coords, (const Vec3f*)vel3, out_div, inv_dx, n); });
}

void refk_rbgs(const void* grid, const int32_t* coords, const float* div, float* p, float dx, uint64_t n, int color, float omega) {
	auto* k = &redBlackGaussSeidelUpdate;
	launch_1d(n, [&] { k((const Grid*)grid, (const Coord*)coords, div, p, dx, n, color, omega); });
}

void refk_subtract_pressure_gradient(const void* grid, const int32_t* coords, uint64_t n, const float* vel3, const float* p, float* out3,
                                     const float* sdf, int has_collision, float inv_dx) {
	auto* k = &subtractPressureGradient;
	launch_1d(n, [&] {
		k((const Grid*)grid, (const Coord*)coords, n, (const Vec3f*)vel3, p, (Vec3f*)out3, sdf, has_collision != 0, inv_dx);
	});
}

void refk_temperature_buoyancy(const float* vel3, const float* temp, float* out3, float dt, float ambient, float strength, uint64_t n) {
	auto* k = &temperature_buoyancy;
	launch_1d(n, [&] { k((const Vec3f*)vel3, temp, (Vec3f*)out3, dt, ambient, strength, n); });
}

void refk_combustion_oxygen(const float* fuel, const float* waste, const float* temperature, float* divergence_io, const float* flame,
                            float* out_fuel, float* out_waste, float* out_temperature, float* out_flame, float temp_gain,
                            float expansion, uint64_t n) {
	auto* k = &combustion_oxygen;
	launch_1d(n, [&] {
		k(fuel, waste, temperature, divergence_io, flame, out_fuel, out_waste, out_temperature, out_flame, temp_gain, expansion, n);
	});
}

// The reference launches it with out == in (HNanoSolver.cu:174); both pointers are the caller's here.
void refk_vorticity_confinement(const void* grid, const int32_t* coords, const float* vel3, float* out3, float dt, float inv_dx,
                                float confinement_scale, float factor_scale, uint64_t n) {
	auto* k = &vorticityConfinement;
	launch_1d(n, [&] {
		k((const Grid*)grid, (const Coord*)coords, (const Vec3f*)vel3, (Vec3f*)out3, dt, inv_dx, confinement_scale, factor_scale, n);
	});
}

void refk_enforce_collision_boundaries(const void* grid, const int32_t* coords, float* vel3, const float* sdf, float voxel_size,
                                       uint64_t n) {
	auto* k = &enforceCollisionBoundaries;
	launch_1d(n, [&] { k((const Grid*)grid, (const Coord*)coords, (Vec3f*)vel3, sdf, voxel_size, n); });
}

// ---- the leaf-per-block forms ProjectNonDivergent launches (PressureProjection.cu:48-65) ----

void refk_divergence_opt(const void* grid, const float* vel3, float* out_div, float inv_dx, int n_leaves) {
	auto* k = &divergence_opt;
	launch_leaf((unsigned)n_leaves, [&] { k((const Grid*)grid, (const Vec3f*)vel3, out_div, inv_dx, n_leaves); });
}

void refk_rbgs_opt(const void* grid, const float* div, float* p, float dx, uint64_t n, int color, float omega, int n_leaves) {
	auto* k = &redBlackGaussSeidelUpdate_opt;
	launch_leaf((unsigned)n_leaves, [&] { k((const Grid*)grid, div, p, dx, n, color, omega); });
}

void refk_subtract_pressure_gradient_opt(const void* grid, const float* vel3, const float* p, float* out3, float inv_dx,
                                         uint64_t n_leaves) {
	auto* k = &subtractPressureGradient_opt;
	launch_leaf((unsigned)n_leaves, [&] { k((const Grid*)grid, (const Vec3f*)vel3, p, (Vec3f*)out3, inv_dx, n_leaves); });
}

}  // extern "C"
