// valu_rate.hip -- what a wave64 pays per VALU instruction on gfx950: v_add_f32 / v_mul_f32 against v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_mov_b32, as independent streams (throughput) with 1..8 waves per SIMD. Decides whether hand-packed arithmetic (two voxels per
// instruction) is a lever for the issue-bound kernels (k_rbgs_block lean form, the advection lerps).
//   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int N_IT = 2048, U = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, unsigned long long* cyc) {
	float x[2 * U];
#pragma unroll
	for (int i = 0; i < 2 * U; ++i) x[i] = a * (threadIdx.x + i);
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < N_IT; ++it) {
#pragma unroll
		for (int i = 0; i < U; ++i) {
			if (MODE == 0) {  // 2U scalar adds
				asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2 * i]) : "v"(b));
				asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2 * i + 1]) : "v"(b));
			} else if (MODE == 1) {  // U packed adds = the same 2U additions
				f2 v = {x[2 * i], x[2 * i + 1]}; f2 w = {b, b};
				asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));
				x[2 * i] = v.x, x[2 * i + 1] = v.y;
			} else if (MODE == 2) {  // 2U scalar muls
				asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[2 * i]) : "v"(b));
				asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[2 * i + 1]) : "v"(b));
			} else if (MODE == 3) {
				f2 v = {x[2 * i], x[2 * i + 1]}; f2 w = {b, b};
				asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(w));
				x[2 * i] = v.x, x[2 * i + 1] = v.y;
			} else if (MODE == 4) {  // U packed moves (two dwords each)
				f2 v = {x[2 * i], x[2 * i + 1]};
				asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(v));
				x[2 * i] = v.x, x[2 * i + 1] = v.y;
			} else if (MODE == 5) {  // 2U scalar moves
				float t;
				asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(x[2 * i]));
				asm volatile("v_mov_b32 %0, %1" : "=v"(x[2 * i]) : "v"(x[2 * i + 1]));
				x[2 * i + 1] = t;
			} else if (MODE == 6) {  // 2U scalar fma
				asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * i]) : "v"(b));
				asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[2 * i + 1]) : "v"(b));
			} else if (MODE == 7) {  // U packed fma
				f2 v = {x[2 * i], x[2 * i + 1]}; f2 w = {b, b};
				asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
				x[2 * i] = v.x, x[2 * i + 1] = v.y;
			} else if (MODE == 8) {  // U dependent-chain scalar adds on ONE register (latency)
				asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
				asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(b));
			} else if (MODE == 9) {  // dependent chain, packed
				f2 v = {x[0], x[1]}; f2 w = {b, b};
				asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));
				asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));
				x[0] = v.x, x[1] = v.y;
			} else if (MODE == 10) {  // min3
				asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x[2 * i]) : "v"(b));
				asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x[2 * i + 1]) : "v"(b));
			} else if (MODE == 11) {  // v_cndmask with vcc
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[2 * i]) : "v"(b));
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[2 * i + 1]) : "v"(b));
			}
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float s = 0;
#pragma unroll
	for (int i = 0; i < 2 * U; ++i) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
	const int threads = 256;  // 4 waves = one per SIMD
	const int blocks = 256 * waves_per_simd;
	float* out; unsigned long long* cyc;
	hipMalloc(&out, sizeof(float) * threads * blocks); hipMalloc(&cyc, 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, threads>>>(out, 1.0f, 1.0001f, cyc);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<MODE><<<blocks, threads>>>(out, 1.0f, 1.0001f, cyc);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
	const double instr_per_wave = (double)N_IT * U * ((MODE == 1 || MODE == 3 || MODE == 4 || MODE == 7) ? 1 : 2);
	// wall per instruction per SIMD in ns and in cycles of the wave's own counter
	printf("%-28s waves/SIMD %d: %8.1f us, %.2f ns per instruction per SIMD, wave-clock %.2f ticks per instruction of ONE wave (100 MHz ticks if s_memrealtime)\n", name, waves_per_simd,
	       ms * 1e3, ms * 1e6 / (instr_per_wave * waves_per_simd), (double)c / instr_per_wave);
	hipFree(out); hipFree(cyc);
}

int main() {
	for (int w : {1, 2, 4, 8}) {
		run<0>("v_add_f32 x2", w);      run<1>("v_pk_add_f32 (same work)", w);
		run<2>("v_mul_f32 x2", w);      run<3>("v_pk_mul_f32 (same work)", w);
		run<6>("v_fma_f32 x2", w);      run<7>("v_pk_fma_f32 (same work)", w);
		run<5>("v_mov_b32 x2", w);      run<4>("v_pk_mov_b32", w);
		run<10>("v_min3_f32 x2", w);    run<11>("v_cndmask_b32 x2", w);
		run<8>("dependent v_add_f32 x2", w); run<9>("dependent v_pk_add_f32 x2", w);
	}
	return 0;
}
