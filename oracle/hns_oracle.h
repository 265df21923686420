/*
 * hns_oracle.h -- CPU ORACLE for the HNanoSolver substep hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path (hnanosolver_amd/, libhns.so) never links or calls it.
 *
 * It restates, operation for operation, the arithmetic of the reference's
 * device kernels (paths below are relative to the reference checkout):
 *   src/Cuda/Kernel.cu         -- all kernels on the path
 *   src/Utils/Stencils.hpp     -- IndexOffsetSampler / IndexSampler<T,0|1>
 *   src/Cuda/Utils.cuh:226-243 -- computeVorticityMag
 *   src/Cuda/HNanoSolver.cu:9-372          -- Compute() launch order
 *   src/Cuda/PressureProjection.cu:9-125   -- pressure_projection_idx / divergence
 *   src/Cuda/Advection.cu:13-166           -- advect_index_grid(_v)
 *
 * Pinning (see oracle/README.md and DESIGN.md "Oracle"):
 *   - topology + samplers are checked against the reference's own
 *     Stencils.hpp + vendored NanoVDB built into oracle/_ref (real reference
 *     code, compiled by oracle/Makefile from /root/reference);
 *   - sampler known answers from Tests/IndexGrid.cpp:212-223 and the index
 *     known answer from externals/nanovdb/unittest/TestNanoVDB.cu:311-355;
 *   - the kernel bodies: the reference's src/Cuda/Kernel.cu itself, compiled
 *     where it lies for the host into oracle/_ref/libhns_refk.so (g++, the
 *     CUDA runtime headers the image ships, command-line macros for the nvcc
 *     intrinsics; oracle/ref_kernels.cpp is the launch). tests/test_ref_kernels.py
 *     holds every function below to it BIT FOR BIT: each kernel alone on
 *     random sparse grids and fields, the "_opt" forms, and the three host
 *     drivers against the reference's launch order. The committed goldens
 *     tests/golden/kernel_goldens_v1.npz are that build's outputs.
 *     Not pinned: nvcc's own FMA contraction (measured on the reference's
 *     own code instead: `make ref_fma`, <= 1e-5 relative L-inf).
 *
 * Layout: flat leaf-dense arrays, index = leaf*512 + (x<<6 | y<<3 | z)
 * (src/Utils/GridBuilder.hpp:160-163), Vec3f as AoS float[3].
 * Build with -ffp-contract=off: every fused multiply-add below is explicit and
 * marks a place where the reference itself calls __fmaf_rn / fmaf.
 */
#ifndef HNS_ORACLE_H
#define HNS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_grid orc_grid;

typedef struct {
	float expansionRate, temperatureRelease, buoyancyStrength, ambientTemp, vorticityScale, factorScale;
} orc_combustion_params; /* src/Cuda/Kernels.cuh:6-13 */

/* ---- topology (IndexOffsetSampler<0>, Stencils.hpp:51-71) ---- */
orc_grid* orc_grid_create(const int32_t* leaf_origins_xyz, int64_t n_leaves); /* leaf l = l-th origin; NULL on duplicate/misaligned */
void orc_grid_destroy(orc_grid*);
/* element read by orc_advect_scalars for out-of-domain taps; 0 (default) is the reference's behaviour (Kernel.cu:133,192,225).
 * Only the partitioned multi-rank tests move it (to the local copy of global element 0). */
void orc_grid_set_outside_element(orc_grid*, uint64_t element_index);
int64_t orc_grid_leaf_count(const orc_grid*);
int64_t orc_grid_voxel_count(const orc_grid*);
uint64_t orc_offset(const orc_grid*, int32_t i, int32_t j, int32_t k); /* 1-based value index, 0 = outside */
void orc_coords(const orc_grid*, int32_t* out_xyz);                    /* N x 3, the array the reference calls d_coords */
/* 1 -> Vec3f trilinear uses fmaf(w, b-a, a) (device branch, Stencils.hpp:131-135); 0 -> a+(b-a)*w (host branch :137) */
void orc_set_vec3_lerp_fma(int on);
void orc_set_threads(int n); /* OpenMP threads used by every kernel below (0 = all cores) */
int orc_get_threads(void);

/* ---- samplers, batch form (Stencils.hpp:74-173) ---- */
void orc_sample_nearest_f(const orc_grid*, const float* data, const int32_t* ijk, int64_t n, float* out);
void orc_sample_trilinear_f(const orc_grid*, const float* data, const float* xyz, int64_t n, float* out);
void orc_sample_trilinear_v(const orc_grid*, const float* data3, const float* xyz, int64_t n, float* out3);

/* ---- kernels (Kernel.cu) ---- */
void orc_advect_vector(const orc_grid*, const float* vel3, float* out3, const float* sdf, int has_collision, float dt, float inv_dx);
void orc_advect_scalar(const orc_grid*, const float* vel3, const float* in, float* out, const float* sdf, int has_collision, float dt,
                       float inv_dx);
void orc_advect_scalars(const orc_grid*, const float* vel3, const float* const* in, float* const* out, int n_scalars, const float* sdf,
                        int has_collision, float dt, float inv_dx);
void orc_divergence(const orc_grid*, const float* vel3, float* out_div, float inv_dx);
void orc_rbgs(const orc_grid*, const float* div, float* p, float dx, int color, float omega);
void orc_subtract_pressure_gradient(const orc_grid*, const float* vel3, const float* p, float* out3, const float* sdf, int has_collision,
                                    float inv_dx);
void orc_combustion_oxygen(const float* fuel, const float* waste, const float* temperature, float* divergence, const float* flame,
                           float* out_fuel, float* out_waste, float* out_temperature, float* out_flame, float temp_gain, float expansion,
                           int64_t n);
void orc_temperature_buoyancy(const float* vel3, const float* temp, float* out3, float dt, float ambient, float strength, int64_t n);
/* reads `vel3`, writes `out3`; the reference launches it in place (a race when factorScale >= 1, HNanoSolver.cu:174) */
void orc_vorticity_confinement(const orc_grid*, const float* vel3, float* out3, float dt, float inv_dx, float confinement_scale,
                               float factor_scale);
void orc_enforce_collision_boundaries(const orc_grid*, float* vel3, const float* sdf, float voxel_size);

/* ---- host drivers ---- */
float orc_omega_compute(float voxel_size); /* HNanoSolver.cu:257, float sinf */
float orc_omega_project(float voxel_size); /* PressureProjection.cu:53, double sin */

/* Compute_Sim (HNanoSolver.cu:9-372). names[i]/fields[i]: float blocks in insertion order; results overwrite in place.
 * returns 0, or <0 with the same refusal conditions as the reference's exceptions. */
int orc_compute_sim(const orc_grid*, float* vel3, const char* const* names, float* const* fields, int n_fields, int iterations, float dt,
                    float voxel_size, const orc_combustion_params*, int has_collision);
int orc_project_non_divergent(const orc_grid*, float* vel3, int64_t iterations, float voxel_size); /* PressureProjection.cu:9-78 */
int orc_divergence_op(const orc_grid*, const float* vel3, float* out_div, float voxel_size);       /* PressureProjection.cu:81-125 */
int orc_advect_index_grid(const orc_grid*, const float* vel3, float* const* fields, int n_fields, float dt, float voxel_size);
int orc_advect_index_grid_velocity(const orc_grid*, float* vel3, float dt, float voxel_size);

#ifdef __cplusplus
}
#endif
#endif
