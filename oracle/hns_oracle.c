/*
 * hns_oracle.c -- CPU ORACLE (test infrastructure only; see hns_oracle.h).
 *
 * Every function cites the reference lines it restates. Floating-point
 * expressions keep the reference's association and operand order; build with
 * -ffp-contract=off so that only the explicit fmaf() calls fuse.
 */
#include "hns_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* topology                                                            */
/* ------------------------------------------------------------------ */

struct orc_grid {
	int64_t n_leaves;
	int32_t* origins; /* n_leaves x 3 */
	uint32_t mask;    /* table size - 1 */
	int32_t* table;   /* open addressing, -1 = empty, else leaf index */
	uint64_t oob;     /* element read for out-of-domain taps by advect_scalars (0 in the reference) */
};

static int g_vec3_fma = 1;
static int g_threads = 0;

void orc_set_vec3_lerp_fma(int on) { g_vec3_fma = on ? 1 : 0; }
void orc_set_threads(int n) { g_threads = n; }
int orc_get_threads(void) {
#ifdef _OPENMP
	return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
	return 1;
#endif
}
#ifdef _OPENMP
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(orc_get_threads())")
#else
#define ORC_PAR
#endif

static inline uint32_t hash3(int32_t x, int32_t y, int32_t z) {
	uint64_t h = (uint64_t)(uint32_t)x * 0x9E3779B97F4A7C15ULL;
	h ^= ((uint64_t)(uint32_t)y + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL;
	h ^= ((uint64_t)(uint32_t)z + 0x165667B1ULL) * 0xD6E8FEB86659FD93ULL;
	h ^= h >> 29;
	h *= 0xBF58476D1CE4E5B9ULL;
	h ^= h >> 32;
	return (uint32_t)h;
}

static int64_t find_leaf(const orc_grid* g, int32_t ox, int32_t oy, int32_t oz) {
	uint32_t s = hash3(ox, oy, oz) & g->mask;
	for (;;) {
		const int32_t l = g->table[s];
		if (l < 0) return -1;
		const int32_t* o = g->origins + 3 * (int64_t)l;
		if (o[0] == ox && o[1] == oy && o[2] == oz) return l;
		s = (s + 1) & g->mask;
	}
}

orc_grid* orc_grid_create(const int32_t* leaf_origins_xyz, int64_t n_leaves) {
	if (n_leaves < 0 || n_leaves > 0x3fffffff) return NULL;
	orc_grid* g = (orc_grid*)calloc(1, sizeof(orc_grid));
	if (!g) return NULL;
	g->n_leaves = n_leaves;
	g->origins = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(n_leaves > 0 ? n_leaves : 1));
	uint32_t size = 16;
	while ((int64_t)size < 2 * n_leaves + 2) size <<= 1;
	g->mask = size - 1;
	g->table = (int32_t*)malloc(sizeof(int32_t) * size);
	if (!g->origins || !g->table) {
		orc_grid_destroy(g);
		return NULL;
	}
	memset(g->table, 0xff, sizeof(int32_t) * size);
	for (int64_t l = 0; l < n_leaves; ++l) {
		const int32_t ox = leaf_origins_xyz[3 * l], oy = leaf_origins_xyz[3 * l + 1], oz = leaf_origins_xyz[3 * l + 2];
		if ((ox & 7) || (oy & 7) || (oz & 7)) {
			orc_grid_destroy(g);
			return NULL;
		}
		g->origins[3 * l] = ox;
		g->origins[3 * l + 1] = oy;
		g->origins[3 * l + 2] = oz;
		if (find_leaf(g, ox, oy, oz) >= 0) { /* duplicate leaf */
			orc_grid_destroy(g);
			return NULL;
		}
		uint32_t s = hash3(ox, oy, oz) & g->mask;
		while (g->table[s] >= 0) s = (s + 1) & g->mask;
		g->table[s] = (int32_t)l;
	}
	return g;
}

void orc_grid_set_outside_element(orc_grid* g, uint64_t element_index) { g->oob = element_index; }

void orc_grid_destroy(orc_grid* g) {
	if (!g) return;
	free(g->origins);
	free(g->table);
	free(g);
}

int64_t orc_grid_leaf_count(const orc_grid* g) { return g->n_leaves; }
int64_t orc_grid_voxel_count(const orc_grid* g) { return g->n_leaves * 512; }

/* Accessor with a one-leaf cache, standing in for ReadAccessor<ValueOnIndex,0,1,2>
 * (Stencils.hpp:53; the cache changes speed only, never the value). */
typedef struct {
	const orc_grid* g;
	int32_t ox, oy, oz;
	int64_t leaf;
	int valid;
} acc_t;

static inline acc_t acc_make(const orc_grid* g) {
	acc_t a;
	a.g = g;
	a.ox = a.oy = a.oz = 0;
	a.leaf = -1;
	a.valid = 0;
	return a;
}

/* IndexOffsetSampler<0>::offset (Stencils.hpp:59-61) -> LeafData<ValueOnIndex>::getValue
 * (NanoVDB.h:4219-4228): for the leaf-dense domain this is mOffset + n = leaf*512 + n + 1; 0 when the leaf is absent. */
static inline uint64_t acc_offset(acc_t* a, int32_t i, int32_t j, int32_t k) {
	const int32_t ox = i & ~7, oy = j & ~7, oz = k & ~7;
	if (!a->valid || ox != a->ox || oy != a->oy || oz != a->oz) {
		a->leaf = find_leaf(a->g, ox, oy, oz);
		a->ox = ox;
		a->oy = oy;
		a->oz = oz;
		a->valid = 1;
	}
	if (a->leaf < 0) return 0;
	return (uint64_t)a->leaf * 512u + (uint64_t)(((i & 7) << 6) | ((j & 7) << 3) | (k & 7)) + 1u;
}

uint64_t orc_offset(const orc_grid* g, int32_t i, int32_t j, int32_t k) {
	acc_t a = acc_make(g);
	return acc_offset(&a, i, j, k);
}

void orc_coords(const orc_grid* g, int32_t* out) {
	/* leaf.offsetToGlobalCoord(n) (GridBuilder.hpp:160-163): n = x<<6 | y<<3 | z */
	for (int64_t l = 0; l < g->n_leaves; ++l)
		for (int n = 0; n < 512; ++n) {
			int32_t* c = out + 3 * (l * 512 + n);
			c[0] = g->origins[3 * l] + (n >> 6);
			c[1] = g->origins[3 * l + 1] + ((n >> 3) & 7);
			c[2] = g->origins[3 * l + 2] + (n & 7);
		}
}

/* ------------------------------------------------------------------ */
/* samplers                                                            */
/* ------------------------------------------------------------------ */

typedef struct {
	float v[3];
} vec3;

static inline vec3 v3(float a, float b, float c) {
	vec3 r = {{a, b, c}};
	return r;
}
static inline vec3 v3_load(const float* d, uint64_t idx) { return v3(d[3 * idx], d[3 * idx + 1], d[3 * idx + 2]); }
static inline void v3_store(float* d, uint64_t idx, vec3 a) {
	d[3 * idx] = a.v[0];
	d[3 * idx + 1] = a.v[1];
	d[3 * idx + 2] = a.v[2];
}
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
static inline vec3 v3_sub(vec3 a, vec3 b) { return v3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
static inline vec3 v3_scale(vec3 a, float s) { return v3(s * a.v[0], s * a.v[1], s * a.v[2]); } /* Math.h:649 */

/* IndexSampler<float,0>::operator() (Stencils.hpp:81-89) */
static inline float nearest_f(acc_t* a, const float* d, int32_t i, int32_t j, int32_t k) {
	const uint64_t off = acc_offset(a, i, j, k);
	return off == 0 ? 0.0f : d[off - 1];
}
static inline vec3 nearest_v(acc_t* a, const float* d, int32_t i, int32_t j, int32_t k) {
	const uint64_t off = acc_offset(a, i, j, k);
	return off == 0 ? v3(0.0f, 0.0f, 0.0f) : v3_load(d, off - 1);
}

/* Floor (Stencils.hpp:25-43): __float2int_rd, then xyz -= float(ijk) */
static inline int32_t floor_frac(float* x) {
	const int32_t i = (int32_t)floorf(*x);
	*x -= (float)i;
	return i;
}

/* float lerp: val_a + ValueT(weight) * (val_b - val_a) (Stencils.hpp:140) */
static inline float lerp_f(float a, float b, float w) { return a + w * (b - a); }
/* Vec3f lerp: device fmaf(weight, b-a, a) (Stencils.hpp:131-135, :20-22); host a + (b-a)*w (:137) */
static inline vec3 lerp_v(vec3 a, vec3 b, float w) {
	vec3 r;
	if (g_vec3_fma) {
		for (int c = 0; c < 3; ++c) r.v[c] = fmaf(w, b.v[c] - a.v[c], a.v[c]);
	} else {
		for (int c = 0; c < 3; ++c) r.v[c] = a.v[c] + w * (b.v[c] - a.v[c]);
	}
	return r;
}

/* TrilinearSampler<float>::sample (Stencils.hpp:117-153); stencil order :104-114 */
static inline float trilinear_f(acc_t* a, const float* d, float x, float y, float z) {
	const int32_t i = floor_frac(&x), j = floor_frac(&y), k = floor_frac(&z);
	const float v000 = nearest_f(a, d, i, j, k);
	const float v001 = nearest_f(a, d, i, j, k + 1);
	const float v010 = nearest_f(a, d, i, j + 1, k);
	const float v011 = nearest_f(a, d, i, j + 1, k + 1);
	const float v100 = nearest_f(a, d, i + 1, j, k);
	const float v101 = nearest_f(a, d, i + 1, j, k + 1);
	const float v110 = nearest_f(a, d, i + 1, j + 1, k);
	const float v111 = nearest_f(a, d, i + 1, j + 1, k + 1);
	const float z0 = lerp_f(v000, v001, z);
	const float z1 = lerp_f(v010, v011, z);
	const float z2 = lerp_f(v100, v101, z);
	const float z3 = lerp_f(v110, v111, z);
	const float y0 = lerp_f(z0, z1, y);
	const float y1 = lerp_f(z2, z3, y);
	return lerp_f(y0, y1, x);
}

static inline vec3 trilinear_v(acc_t* a, const float* d, float x, float y, float z) {
	const int32_t i = floor_frac(&x), j = floor_frac(&y), k = floor_frac(&z);
	const vec3 v000 = nearest_v(a, d, i, j, k);
	const vec3 v001 = nearest_v(a, d, i, j, k + 1);
	const vec3 v010 = nearest_v(a, d, i, j + 1, k);
	const vec3 v011 = nearest_v(a, d, i, j + 1, k + 1);
	const vec3 v100 = nearest_v(a, d, i + 1, j, k);
	const vec3 v101 = nearest_v(a, d, i + 1, j, k + 1);
	const vec3 v110 = nearest_v(a, d, i + 1, j + 1, k);
	const vec3 v111 = nearest_v(a, d, i + 1, j + 1, k + 1);
	const vec3 z0 = lerp_v(v000, v001, z);
	const vec3 z1 = lerp_v(v010, v011, z);
	const vec3 z2 = lerp_v(v100, v101, z);
	const vec3 z3 = lerp_v(v110, v111, z);
	const vec3 y0 = lerp_v(z0, z1, y);
	const vec3 y1 = lerp_v(z2, z3, y);
	return lerp_v(y0, y1, x);
}

void orc_sample_nearest_f(const orc_grid* g, const float* data, const int32_t* ijk, int64_t n, float* out) {
	acc_t a = acc_make(g);
	for (int64_t t = 0; t < n; ++t) out[t] = nearest_f(&a, data, ijk[3 * t], ijk[3 * t + 1], ijk[3 * t + 2]);
}
void orc_sample_trilinear_f(const orc_grid* g, const float* data, const float* xyz, int64_t n, float* out) {
	acc_t a = acc_make(g);
	for (int64_t t = 0; t < n; ++t) out[t] = trilinear_f(&a, data, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]);
}
void orc_sample_trilinear_v(const orc_grid* g, const float* data3, const float* xyz, int64_t n, float* out3) {
	acc_t a = acc_make(g);
	for (int64_t t = 0; t < n; ++t) v3_store(out3, (uint64_t)t, trilinear_v(&a, data3, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]));
}

/* ------------------------------------------------------------------ */
/* collision helpers (Kernel.cu:8-74)                                  */
/* ------------------------------------------------------------------ */

/* gradientSDF with Vec3T = Coord (Kernel.cu:15-28): nearest taps, (0.5f*inv) factored */
static inline vec3 gradient_sdf(acc_t* a, const float* sdf, int32_t i, int32_t j, int32_t k, float inv_dx) {
	const float right = nearest_f(a, sdf, i + 1, j, k);
	const float left = nearest_f(a, sdf, i - 1, j, k);
	const float top = nearest_f(a, sdf, i, j + 1, k);
	const float bottom = nearest_f(a, sdf, i, j - 1, k);
	const float front = nearest_f(a, sdf, i, j, k + 1);
	const float back = nearest_f(a, sdf, i, j, k - 1);
	return v3_scale(v3(right - left, top - bottom, front - back), 0.5f * inv_dx);
}

/* getSDFNormal (Kernel.cu:40-47): g / len == (1/len) * g (Math.h:650) */
static inline vec3 sdf_normal(acc_t* a, const float* sdf, int32_t i, int32_t j, int32_t k, float eps) {
	const vec3 g = gradient_sdf(a, sdf, i, j, k, eps);
	const float len = sqrtf(g.v[0] * g.v[0] + g.v[1] * g.v[1] + g.v[2] * g.v[2]);
	return len > 1e-6f ? v3_scale(g, 1.0f / len) : v3(0.0f, 0.0f, 0.0f);
}

/* applyNoSlipBoundary (Kernel.cu:57-74) */
static inline vec3 no_slip(vec3 v, vec3 n) {
	const float vdotn = v.v[0] * n.v[0] + v.v[1] * n.v[1] + v.v[2] * n.v[2];
	const vec3 v_normal = v3_scale(n, vdotn);
	return v3_sub(v, v_normal);
}

/* ------------------------------------------------------------------ */
/* kernels                                                             */
/* ------------------------------------------------------------------ */

#define FOR_EACH_VOXEL(g)                                            \
	ORC_PAR                                                          \
	for (int64_t leaf__ = 0; leaf__ < (g)->n_leaves; ++leaf__) {     \
		acc_t acc = acc_make(g);                                     \
		const int32_t* org__ = (g)->origins + 3 * leaf__;            \
		for (int n__ = 0; n__ < 512; ++n__) {                        \
			const uint64_t idx = (uint64_t)leaf__ * 512u + (uint64_t)n__; \
			const int32_t ci = org__[0] + (n__ >> 6), cj = org__[1] + ((n__ >> 3) & 7), ck = org__[2] + (n__ & 7);
#define END_FOR_EACH_VOXEL \
	}                      \
	}

/* advect_vector (Kernel.cu:354-453) */
void orc_advect_vector(const orc_grid* g, const float* vel, float* out, const float* sdf, int has_collision, float dt, float inv_dx) {
	const float scaled_dt = dt * inv_dx;
	FOR_EACH_VOXEL(g)
	const vec3 pos = v3((float)ci, (float)cj, (float)ck);
	const vec3 velOrig = nearest_v(&acc, vel, ci, cj, ck);
	vec3 backPos = v3_sub(pos, v3_scale(velOrig, scaled_dt));
	if (has_collision && sdf) {
		if (trilinear_f(&acc, sdf, backPos.v[0], backPos.v[1], backPos.v[2]) < 0.0f) backPos = pos;
	}
	const vec3 velForward = trilinear_v(&acc, vel, backPos.v[0], backPos.v[1], backPos.v[2]);
	vec3 fwdPos2 = v3_add(backPos, v3_scale(velForward, scaled_dt));
	if (has_collision && sdf) {
		if (trilinear_f(&acc, sdf, fwdPos2.v[0], fwdPos2.v[1], fwdPos2.v[2]) < 0.0f) fwdPos2 = backPos;
	}
	const vec3 velBackward = trilinear_v(&acc, vel, fwdPos2.v[0], fwdPos2.v[1], fwdPos2.v[2]);
	const vec3 errorVec = v3_sub(velOrig, velBackward);
	vec3 velCorr = v3_add(velForward, v3_scale(errorVec, 0.5f));
	vec3 minVel = velOrig, maxVel = velOrig;
	for (int dim = 0; dim < 3; ++dim)
		for (int offset = -1; offset <= 1; offset += 2) {
			int32_t nc[3] = {ci, cj, ck};
			nc[dim] += offset;
			const vec3 nv = nearest_v(&acc, vel, nc[0], nc[1], nc[2]);
			for (int c = 0; c < 3; ++c) {
				minVel.v[c] = fminf(minVel.v[c], nv.v[c]);
				maxVel.v[c] = fmaxf(maxVel.v[c], nv.v[c]);
			}
		}
	for (int c = 0; c < 3; ++c) {
		minVel.v[c] = fminf(minVel.v[c], velForward.v[c]);
		maxVel.v[c] = fmaxf(maxVel.v[c], velForward.v[c]);
		velCorr.v[c] = fmaxf(minVel.v[c], fminf(velCorr.v[c], maxVel.v[c]));
	}
	if (has_collision && sdf) { /* Kernel.cu:433-450 */
		const float sdf_value = nearest_f(&acc, sdf, ci, cj, ck);
		if (sdf_value < 0.0f) {
			velCorr = v3(0.0f, 0.0f, 0.0f);
		} else if (sdf_value < 0.1f) {
			const vec3 normal = sdf_normal(&acc, sdf, ci, cj, ck, inv_dx);
			const float blend = 1.0f - (sdf_value / 1.5f);
			const vec3 ns = no_slip(velCorr, normal);
			velCorr = v3_add(v3_scale(velCorr, 1.0f - blend), v3_scale(ns, blend));
		}
	}
	v3_store(out, idx, velCorr);
	END_FOR_EACH_VOXEL
}

/* advect_scalar (Kernel.cu:269-352) */
void orc_advect_scalar(const orc_grid* g, const float* vel, const float* in, float* out, const float* sdf, int has_collision, float dt,
                       float inv_dx) {
	const float scaled_dt = dt * inv_dx;
	FOR_EACH_VOXEL(g)
	const vec3 posCell = v3((float)ci, (float)cj, (float)ck);
	const float phiOrig = nearest_f(&acc, in, ci, cj, ck);
	const vec3 velCenter = nearest_v(&acc, vel, ci, cj, ck);
	vec3 backPos = v3_sub(posCell, v3_scale(velCenter, scaled_dt));
	if (has_collision && sdf) {
		if (trilinear_f(&acc, sdf, backPos.v[0], backPos.v[1], backPos.v[2]) < 0.0f) backPos = posCell;
	}
	const float phiForward = trilinear_f(&acc, in, backPos.v[0], backPos.v[1], backPos.v[2]);
	const vec3 velF = trilinear_v(&acc, vel, backPos.v[0], backPos.v[1], backPos.v[2]);
	vec3 fwdPos2 = v3_add(backPos, v3_scale(velF, scaled_dt));
	if (has_collision && sdf) {
		if (trilinear_f(&acc, sdf, fwdPos2.v[0], fwdPos2.v[1], fwdPos2.v[2]) < 0.0f) fwdPos2 = backPos;
	}
	const float phiBackward = trilinear_f(&acc, in, fwdPos2.v[0], fwdPos2.v[1], fwdPos2.v[2]);
	const float error = phiOrig - phiBackward;
	float phiCorr = phiForward + 0.5f * error;
	float minVal = phiOrig, maxVal = phiOrig;
	for (int dim = 0; dim < 3; ++dim)
		for (int offset = -1; offset <= 1; offset += 2) {
			int32_t nc[3] = {ci, cj, ck};
			nc[dim] += offset;
			const float nv = nearest_f(&acc, in, nc[0], nc[1], nc[2]);
			minVal = fminf(minVal, nv);
			maxVal = fmaxf(maxVal, nv);
		}
	minVal = fminf(minVal, phiForward);
	maxVal = fmaxf(maxVal, phiForward);
	phiCorr = fmaxf(minVal, fminf(phiCorr, maxVal));
	out[idx] = phiCorr;
	END_FOR_EACH_VOXEL
}

/* advect_scalars (Kernel.cu:118-266) */
typedef struct {
	uint64_t indices[8];
	float weights[8];
} interp_t;

/* setupInterpolation lambda (Kernel.cu:163-196); out-of-domain taps read element 0 (:192) */
static inline interp_t setup_interp(acc_t* a, vec3 pos) {
	interp_t d;
	const float x = pos.v[0], y = pos.v[1], z = pos.v[2];
	const int32_t i0 = (int32_t)floorf(x), i1 = i0 + 1;
	const int32_t j0 = (int32_t)floorf(y), j1 = j0 + 1;
	const int32_t k0 = (int32_t)floorf(z), k1 = k0 + 1;
	const float tx = x - (float)i0, ty = y - (float)j0, tz = z - (float)k0;
	const float itx = 1.0f - tx, ity = 1.0f - ty, itz = 1.0f - tz;
	const float w00 = itx * ity, w10 = tx * ity, w01 = itx * ty, w11 = tx * ty;
	d.weights[0] = w00 * itz;
	d.weights[1] = w10 * itz;
	d.weights[2] = w01 * itz;
	d.weights[3] = w11 * itz;
	d.weights[4] = w00 * tz;
	d.weights[5] = w10 * tz;
	d.weights[6] = w01 * tz;
	d.weights[7] = w11 * tz;
	const int32_t c[8][3] = {{i0, j0, k0}, {i1, j0, k0}, {i0, j1, k0}, {i1, j1, k0}, {i0, j0, k1}, {i1, j0, k1}, {i0, j1, k1}, {i1, j1, k1}};
	for (int t = 0; t < 8; ++t) {
		const uint64_t off = acc_offset(a, c[t][0], c[t][1], c[t][2]);
		d.indices[t] = off == 0 ? a->g->oob : off - 1;
	}
	return d;
}

void orc_advect_scalars(const orc_grid* g, const float* vel, const float* const* in, float* const* out, int n_scalars, const float* sdf,
                        int has_collision, float dt, float inv_dx) {
	const float scaled_dt = dt * inv_dx;
	static const int offs[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}}; /* Kernel.cu:219 */
	FOR_EACH_VOXEL(g)
	uint64_t origIndex = acc_offset(&acc, ci, cj, ck);
	origIndex = origIndex == 0 ? g->oob : origIndex - 1;
	const vec3 posCell = v3((float)ci, (float)cj, (float)ck);
	const vec3 velCenter = v3_load(vel, origIndex);
	vec3 backPos = v3_sub(posCell, v3_scale(velCenter, scaled_dt));
	if (has_collision && sdf) { /* the test is made twice (Kernel.cu:142-155); the second is a no-op once the first fired */
		if (trilinear_f(&acc, sdf, backPos.v[0], backPos.v[1], backPos.v[2]) < 0.0f) backPos = posCell;
		const int inCollision = trilinear_f(&acc, sdf, backPos.v[0], backPos.v[1], backPos.v[2]) < 0.0f;
		backPos = inCollision ? posCell : backPos;
	}
	const interp_t back = setup_interp(&acc, backPos);
	vec3 velF = v3(0.0f, 0.0f, 0.0f);
	for (int j = 0; j < 8; ++j) velF = v3_add(velF, v3_scale(v3_load(vel, back.indices[j]), back.weights[j])); /* :201-206 */
	vec3 fwdPos2 = v3_add(backPos, v3_scale(velF, scaled_dt));
	if (has_collision && sdf) {
		const int fwdIn = trilinear_f(&acc, sdf, fwdPos2.v[0], fwdPos2.v[1], fwdPos2.v[2]) < 0.0f;
		fwdPos2 = fwdIn ? backPos : fwdPos2;
	}
	const interp_t fwd = setup_interp(&acc, fwdPos2);
	uint64_t nbrIdx[6];
	for (int n = 0; n < 6; ++n) {
		const uint64_t off = acc_offset(&acc, ci + offs[n][0], cj + offs[n][1], ck + offs[n][2]);
		nbrIdx[n] = off == 0 ? g->oob : off - 1;
	}
	for (int s = 0; s < n_scalars; ++s) {
		const float* inData = in[s];
		const float phiOrig = inData[origIndex];
		float phiForward = 0.0f, phiBackward = 0.0f;
		for (int j = 0; j < 8; ++j) {
			phiForward = fmaf(inData[back.indices[j]], back.weights[j], phiForward);
			phiBackward = fmaf(inData[fwd.indices[j]], fwd.weights[j], phiBackward);
		}
		const float error = phiOrig - phiBackward;
		const float phiCorr = fmaf(0.5f, error, phiForward);
		float minVal = phiOrig, maxVal = phiOrig;
		for (int n = 0; n < 6; ++n) {
			const float val = inData[nbrIdx[n]];
			minVal = fminf(minVal, val);
			maxVal = fmaxf(maxVal, val);
		}
		minVal = fminf(minVal, phiForward);
		maxVal = fmaxf(maxVal, phiForward);
		out[s][idx] = fmaxf(minVal, fminf(phiCorr, maxVal));
	}
	END_FOR_EACH_VOXEL
}

/* divergence (Kernel.cu:499-519); divergence_opt (:455-496) evaluates 0.5f*(c+n), the same value */
void orc_divergence(const orc_grid* g, const float* vel, float* out_div, float inv_dx) {
	FOR_EACH_VOXEL(g)
	const vec3 center = v3_load(vel, idx);
	const float xp = (center.v[0] + nearest_v(&acc, vel, ci + 1, cj, ck).v[0]) * 0.5f;
	const float xm = (center.v[0] + nearest_v(&acc, vel, ci - 1, cj, ck).v[0]) * 0.5f;
	const float yp = (center.v[1] + nearest_v(&acc, vel, ci, cj + 1, ck).v[1]) * 0.5f;
	const float ym = (center.v[1] + nearest_v(&acc, vel, ci, cj - 1, ck).v[1]) * 0.5f;
	const float zp = (center.v[2] + nearest_v(&acc, vel, ci, cj, ck + 1).v[2]) * 0.5f;
	const float zm = (center.v[2] + nearest_v(&acc, vel, ci, cj, ck - 1).v[2]) * 0.5f;
	out_div[idx] = (xp - xm + yp - ym + zp - zm) * inv_dx;
	END_FOR_EACH_VOXEL
}

/* redBlackGaussSeidelUpdate (Kernel.cu:591-623) == _opt (:521-588). One colour; in place is race-free because a
 * voxel of one colour only reads voxels of the other colour. */
void orc_rbgs(const orc_grid* g, const float* div, float* p, float dx, int color, float omega) {
	const float dx2 = dx * dx;
	const float inv6 = 0.166666667f;
	FOR_EACH_VOXEL(g)
	if (((ci + cj + ck) & 1) != color) continue;
	const float pxp1 = nearest_f(&acc, p, ci + 1, cj, ck);
	const float pxm1 = nearest_f(&acc, p, ci - 1, cj, ck);
	const float pyp1 = nearest_f(&acc, p, ci, cj + 1, ck);
	const float pym1 = nearest_f(&acc, p, ci, cj - 1, ck);
	const float pzp1 = nearest_f(&acc, p, ci, cj, ck + 1);
	const float pzm1 = nearest_f(&acc, p, ci, cj, ck - 1);
	const float divVal = div[idx];
	const float pOld = p[idx];
	const float pGS = ((pxp1 + pxm1 + pyp1 + pym1 + pzp1 + pzm1) - divVal * dx2) * inv6;
	p[idx] = pOld + omega * (pGS - pOld);
	END_FOR_EACH_VOXEL
}

/* subtractPressureGradient (Kernel.cu:765-829); _opt (:694-762) computes (d*0.5f)*inv, the same value */
void orc_subtract_pressure_gradient(const orc_grid* g, const float* vel, const float* p, float* out, const float* sdf, int has_collision,
                                    float inv_dx) {
	FOR_EACH_VOXEL(g)
	const vec3 u_star = v3_load(vel, idx);
	const float p_xp = nearest_f(&acc, p, ci + 1, cj, ck);
	const float p_xm = nearest_f(&acc, p, ci - 1, cj, ck);
	const float p_yp = nearest_f(&acc, p, ci, cj + 1, ck);
	const float p_ym = nearest_f(&acc, p, ci, cj - 1, ck);
	const float p_zp = nearest_f(&acc, p, ci, cj, ck + 1);
	const float p_zm = nearest_f(&acc, p, ci, cj, ck - 1);
	const vec3 grad = v3_scale(v3_scale(v3(p_xp - p_xm, p_yp - p_ym, p_zp - p_zm), 0.5f), inv_dx);
	vec3 u_final = v3_sub(u_star, grad);
	if (has_collision && sdf) { /* Kernel.cu:809-826 */
		const float sdf_value = nearest_f(&acc, sdf, ci, cj, ck);
		if (sdf_value < 0.0f) {
			u_final = v3(0.0f, 0.0f, 0.0f);
		} else if (sdf_value < 0.1f) {
			const vec3 normal = sdf_normal(&acc, sdf, ci, cj, ck, inv_dx);
			const float blend = 1.0f - (sdf_value / 0.1f);
			const vec3 ns = no_slip(u_final, normal);
			u_final = v3_add(v3_scale(u_final, 1.0f - blend), v3_scale(ns, blend));
		}
	}
	v3_store(out, idx, u_final);
	END_FOR_EACH_VOXEL
}

/* combustion_oxygen (Kernel.cu:923-966) */
void orc_combustion_oxygen(const float* fuelData, const float* wasteData, const float* temperatureData, float* divergenceData,
                           const float* flameData, float* outFuel, float* outWaste, float* outTemperature, float* outFlame,
                           float temp_gain, float expansion, int64_t n) {
	ORC_PAR
	for (int64_t idx = 0; idx < n; ++idx) {
		float fuel = fuelData[idx];
		const float waste = wasteData[idx];
		const float temperature = temperatureData[idx];
		const float flame = flameData[idx];
		if (fuel < 0.001f) fuel = 0.0f;
		const float oxygen = 1.0f - fuel - waste;
		if (oxygen < 0.0f) {
			outFuel[idx] = fuel;
			outWaste[idx] = waste;
			outTemperature[idx] = temperature;
			outFlame[idx] = flame;
			continue;
		}
		const float burn = fminf(oxygen, fuel);
		outFuel[idx] = fuel - burn;
		outWaste[idx] = waste + burn * 2.0f;
		outTemperature[idx] = temperature + burn * temp_gain;
		divergenceData[idx] += burn * expansion;
		outFlame[idx] = fmaxf(flame, fminf(1.0f, burn * 10.0f));
	}
}

/* temperature_buoyancy (Kernel.cu:831-847) */
void orc_temperature_buoyancy(const float* vel, const float* temp, float* out, float dt, float ambient, float strength, int64_t n) {
	ORC_PAR
	for (int64_t idx = 0; idx < n; ++idx) {
		const vec3 v = v3_load(vel, (uint64_t)idx);
		const float t = temp[idx];
		if (t <= ambient) {
			v3_store(out, (uint64_t)idx, v);
			continue;
		}
		const float tempDiff = t - ambient;
		const vec3 force = v3(0.0f, fmaxf(0.0f, tempDiff * strength), 0.0f);
		v3_store(out, (uint64_t)idx, v3_add(v, v3_scale(force, dt)));
	}
}

/* computeVorticityMag (Utils.cuh:226-243) */
static inline float vorticity_mag(acc_t* a, const float* vel, int32_t i, int32_t j, int32_t k, float factor) {
	const vec3 u_pX = nearest_v(a, vel, i + 1, j, k), u_mX = nearest_v(a, vel, i - 1, j, k);
	const vec3 u_pY = nearest_v(a, vel, i, j + 1, k), u_mY = nearest_v(a, vel, i, j - 1, k);
	const vec3 u_pZ = nearest_v(a, vel, i, j, k + 1), u_mZ = nearest_v(a, vel, i, j, k - 1);
	const float ox = ((u_pY.v[2] - u_mY.v[2]) - (u_pZ.v[1] - u_mZ.v[1])) * factor;
	const float oy = ((u_pZ.v[0] - u_mZ.v[0]) - (u_pX.v[2] - u_mX.v[2])) * factor;
	const float oz = ((u_pX.v[1] - u_mX.v[1]) - (u_pY.v[0] - u_mY.v[0])) * factor;
	return sqrtf(ox * ox + oy * oy + oz * oz);
}

/* vorticityConfinement (Kernel.cu:970-1024). nanovdb::Coord(factorScale,0,0) truncates the float to int (:998-1007). */
void orc_vorticity_confinement(const orc_grid* g, const float* vel, float* out, float dt, float inv_dx, float confinementScale,
                               float factorScale) {
	const float factor = (float)(0.5 * (double)inv_dx); /* `0.5 * inv_dx` is a double product (Kernel.cu:982) */
	const int32_t fs = (int32_t)factorScale;
	FOR_EACH_VOXEL(g)
	const vec3 u_pX = nearest_v(&acc, vel, ci + 1, cj, ck), u_mX = nearest_v(&acc, vel, ci - 1, cj, ck);
	const vec3 u_pY = nearest_v(&acc, vel, ci, cj + 1, ck), u_mY = nearest_v(&acc, vel, ci, cj - 1, ck);
	const vec3 u_pZ = nearest_v(&acc, vel, ci, cj, ck + 1), u_mZ = nearest_v(&acc, vel, ci, cj, ck - 1);
	const float omega_x = ((u_pY.v[2] - u_mY.v[2]) - (u_pZ.v[1] - u_mZ.v[1])) * factor;
	const float omega_y = ((u_pZ.v[0] - u_mZ.v[0]) - (u_pX.v[2] - u_mX.v[2])) * factor;
	const float omega_z = ((u_pX.v[1] - u_mX.v[1]) - (u_pY.v[0] - u_mY.v[0])) * factor;
	const float m_pX = vorticity_mag(&acc, vel, ci + fs, cj, ck, factor), m_mX = vorticity_mag(&acc, vel, ci - fs, cj, ck, factor);
	const float m_pY = vorticity_mag(&acc, vel, ci, cj + fs, ck, factor), m_mY = vorticity_mag(&acc, vel, ci, cj - fs, ck, factor);
	const float m_pZ = vorticity_mag(&acc, vel, ci, cj, ck + fs, factor), m_mZ = vorticity_mag(&acc, vel, ci, cj, ck - fs, factor);
	const float grad_x = (m_pX - m_mX) * 0.5f * inv_dx;
	const float grad_y = (m_pY - m_mY) * 0.5f * inv_dx;
	const float grad_z = (m_pZ - m_mZ) * 0.5f * inv_dx;
	const float gradLen = sqrtf(grad_x * grad_x + grad_y * grad_y + grad_z * grad_z) + 1e-5f;
	const float Nx = grad_x / gradLen, Ny = grad_y / gradLen, Nz = grad_z / gradLen;
	const vec3 force = v3(confinementScale * (Ny * omega_z - Nz * omega_y), confinementScale * (Nz * omega_x - Nx * omega_z),
	                      confinementScale * (Nx * omega_y - Ny * omega_x));
	v3_store(out, idx, v3_add(v3_load(vel, idx), v3_scale(force, dt)));
	END_FOR_EACH_VOXEL
}

/* enforceCollisionBoundaries (Kernel.cu:77-116) */
void orc_enforce_collision_boundaries(const orc_grid* g, float* vel, const float* sdf, float voxelSize) {
	if (!sdf) return;
	FOR_EACH_VOXEL(g)
	const float sdf_value = nearest_f(&acc, sdf, ci, cj, ck);
	if (sdf_value < 0.0f) {
		v3_store(vel, idx, v3(0.0f, 0.0f, 0.0f));
		continue;
	}
	const float collisionMargin = (float)0.1;
	if (sdf_value < collisionMargin) {
		const vec3 normal = sdf_normal(&acc, sdf, ci, cj, ck, 1.0f / voxelSize);
		const float blend = 1.0f - (sdf_value / collisionMargin);
		const vec3 velocity = v3_load(vel, idx);
		const vec3 modified = no_slip(velocity, normal);
		v3_store(vel, idx, v3_add(v3_scale(velocity, 1.0f - blend), v3_scale(modified, blend)));
	}
	END_FOR_EACH_VOXEL
}

/* ------------------------------------------------------------------ */
/* host drivers                                                        */
/* ------------------------------------------------------------------ */

float orc_omega_compute(float voxelSize) { return 2.0f / (1.0f + sinf((float)3.14159 * voxelSize)); } /* HNanoSolver.cu:257 */
float orc_omega_project(float voxelSize) {                                                           /* PressureProjection.cu:53 */
	return (float)(2.0f / (1.0f + sin(3.14159 * (double)voxelSize)));
}

static int find_name(const char* const* names, int n, const char* key) {
	for (int i = 0; i < n; ++i)
		if (strcmp(names[i], key) == 0) return i;
	return -1;
}

/* Compute (HNanoSolver.cu:9-372) */
int orc_compute_sim(const orc_grid* g, float* vel, const char* const* names, float* const* fields, int n_fields, int iterations, float dt,
                    float voxelSize, const orc_combustion_params* params, int hasCollision) {
	if (voxelSize <= 0.0f) return -1; /* :12-14 */
	if (dt < 0.0f) return -2;         /* :15-17 */
	if (iterations <= 0) return -3;   /* :18-20 */
	if (!g) return -4;                /* :21-23 */
	const int64_t N = orc_grid_voxel_count(g);
	if (N == 0) return 0; /* :26-28 */
	if (!vel) return -5;
	if (n_fields <= 0) return -6; /* :61-63 */
	const float inv_dx = 1.0f / voxelSize;
	const int i_sdf = hasCollision ? find_name(names, n_fields, "collision_sdf") : -1; /* :66-75 */
	const int hasCollisionData = i_sdf >= 0 && fields[i_sdf] != NULL;
	const int i_fuel = find_name(names, n_fields, "fuel"), i_waste = find_name(names, n_fields, "waste");
	const int i_temp = find_name(names, n_fields, "temperature"), i_flame = find_name(names, n_fields, "flame");
	if (i_fuel < 0 || i_waste < 0 || i_temp < 0 || i_flame < 0) return -7; /* :193-201 */

	const size_t fb = sizeof(float) * (size_t)N;
	float* d_velocity = (float*)malloc(3 * fb);
	float* d_advected = (float*)calloc((size_t)N * 3, sizeof(float));
	float* d_tmpvel = (float*)malloc(3 * fb);
	float* d_div = (float*)calloc((size_t)N, sizeof(float));
	float* d_p = (float*)calloc((size_t)N, sizeof(float));
	float* d_sdf = NULL;
	float** d_in = (float**)calloc((size_t)n_fields, sizeof(float*));
	float** d_out = (float**)calloc((size_t)n_fields, sizeof(float*));
	memcpy(d_velocity, vel, 3 * fb);
	if (hasCollisionData) {
		d_sdf = (float*)malloc(fb);
		memcpy(d_sdf, fields[i_sdf], fb);
	}
	for (int i = 0; i < n_fields; ++i) {
		d_in[i] = (float*)malloc(fb);
		memcpy(d_in[i], fields[i], fb);
		d_out[i] = (float*)calloc((size_t)N, sizeof(float)); /* :115-117 */
	}

	if (hasCollisionData) orc_enforce_collision_boundaries(g, d_velocity, d_sdf, voxelSize);        /* :153-157 */
	orc_advect_vector(g, d_velocity, d_advected, d_sdf, hasCollisionData, dt, inv_dx);                /* :162-170 */
	orc_vorticity_confinement(g, d_advected, d_tmpvel, dt, inv_dx, params->vorticityScale, params->factorScale); /* :172-176 */
	memcpy(d_advected, d_tmpvel, 3 * fb);
	orc_divergence(g, d_advected, d_div, inv_dx);                                                     /* :181-188 */
	orc_combustion_oxygen(d_in[i_fuel], d_in[i_waste], d_in[i_temp], d_div, d_in[i_flame], d_out[i_fuel], d_out[i_waste], d_out[i_temp],
	                      d_out[i_flame], params->temperatureRelease, params->expansionRate, N);     /* :211-221 */
	orc_temperature_buoyancy(d_advected, d_out[i_temp], d_advected, dt, params->ambientTemp, params->buoyancyStrength, N); /* :226-234 */
	{ /* :239-246 move outputs -> inputs, fresh zeroed outputs */
		const int comb[4] = {i_fuel, i_waste, i_temp, i_flame};
		for (int c = 0; c < 4; ++c) {
			free(d_in[comb[c]]);
			d_in[comb[c]] = d_out[comb[c]];
			d_out[comb[c]] = (float*)calloc((size_t)N, sizeof(float));
		}
	}
	{ /* :256-272 */
		const float omega = orc_omega_compute(voxelSize);
		for (int it = 0; it < iterations; ++it) {
			orc_rbgs(g, d_div, d_p, voxelSize, 0, omega);
			orc_rbgs(g, d_div, d_p, voxelSize, 1, omega);
		}
	}
	orc_subtract_pressure_gradient(g, d_advected, d_p, d_velocity, d_sdf, hasCollisionData, inv_dx); /* :278-289 */
	if (hasCollisionData) orc_enforce_collision_boundaries(g, d_velocity, d_sdf, voxelSize);         /* :292-296 */
	{ /* :321-356 */
		const float** ins = (const float**)calloc((size_t)n_fields, sizeof(float*));
		float** outs = (float**)calloc((size_t)n_fields, sizeof(float*));
		int S = 0;
		for (int i = 0; i < n_fields; ++i) {
			if (strcmp(names[i], "collision_sdf") == 0) continue;
			ins[S] = d_in[i];
			outs[S] = d_out[i];
			++S;
		}
		orc_advect_scalars(g, d_velocity, ins, outs, S, d_sdf, hasCollisionData, dt, inv_dx);
		free(ins);
		free(outs);
	}
	memcpy(vel, d_velocity, 3 * fb);                                   /* :361 */
	for (int i = 0; i < n_fields; ++i) memcpy(fields[i], d_out[i], fb); /* :364-369 (collision_sdf comes back zeroed) */

	for (int i = 0; i < n_fields; ++i) {
		free(d_in[i]);
		free(d_out[i]);
	}
	free(d_in);
	free(d_out);
	free(d_sdf);
	free(d_p);
	free(d_div);
	free(d_tmpvel);
	free(d_advected);
	free(d_velocity);
	return 0;
}

/* pressure_projection_idx (PressureProjection.cu:9-78) */
int orc_project_non_divergent(const orc_grid* g, float* vel, int64_t iterations, float voxelSize) {
	if (!g || !vel) return -5;
	const int64_t N = orc_grid_voxel_count(g);
	if (N == 0) return 0;
	float* d_div = (float*)calloc((size_t)N, sizeof(float));
	float* d_p = (float*)calloc((size_t)N, sizeof(float));
	orc_divergence(g, vel, d_div, 1.0f / voxelSize); /* :48 */
	const float omega = orc_omega_project(voxelSize);
	for (int64_t it = 0; it < iterations; ++it) { /* :54-59 */
		orc_rbgs(g, d_div, d_p, voxelSize, 0, omega);
		orc_rbgs(g, d_div, d_p, voxelSize, 1, omega);
	}
	orc_subtract_pressure_gradient(g, vel, d_p, vel, NULL, 0, 1.0f / voxelSize); /* :64, in place: each voxel reads only its own u */
	free(d_p);
	free(d_div);
	return 0;
}

/* divergence (PressureProjection.cu:81-125) */
int orc_divergence_op(const orc_grid* g, const float* vel, float* out_div, float voxelSize) {
	if (!g || !vel || !out_div) return -5;
	orc_divergence(g, vel, out_div, 1.0f / voxelSize);
	return 0;
}

/* advect_index_grid (Advection.cu:13-112): every float block through advect_scalar, no collision */
int orc_advect_index_grid(const orc_grid* g, const float* vel, float* const* fields, int n_fields, float dt, float voxelSize) {
	if (!g || !vel) return -5;
	if (n_fields <= 0) return -6;
	const int64_t N = orc_grid_voxel_count(g);
	float* tmp = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
	for (int i = 0; i < n_fields; ++i) {
		orc_advect_scalar(g, vel, fields[i], tmp, NULL, 0, dt, 1.0f / voxelSize);
		memcpy(fields[i], tmp, sizeof(float) * (size_t)N);
	}
	free(tmp);
	return 0;
}

/* advect_index_grid_v (Advection.cu:114-166) */
int orc_advect_index_grid_velocity(const orc_grid* g, float* vel, float dt, float voxelSize) {
	if (!g || !vel) return -5;
	const int64_t N = orc_grid_voxel_count(g);
	float* tmp = (float*)calloc((size_t)(N > 0 ? N : 1) * 3, sizeof(float));
	orc_advect_vector(g, vel, tmp, NULL, 0, dt, 1.0f / voxelSize);
	memcpy(vel, tmp, sizeof(float) * 3 * (size_t)N);
	free(tmp);
	return 0;
}
