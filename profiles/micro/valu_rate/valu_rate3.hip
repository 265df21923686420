// valu_rate3.hip -- is the VCC form of v_cndmask_b32 really five times dearer than the SGPR-pair form (valu_rate2: 9.2 against 1.8 ns)? The pairs a compiler
// emits: v_cmp -> vcc, v_cndmask ..., vcc against v_cmp_e64 -> s[n:n+1], v_cndmask_e64 ..., s[n:n+1]; and the select alone with VCC written once, by SALU / by VALU.
//   hipcc --offload-arch=gfx950 -O2 valu_rate3.hip -o valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N_IT = 2048, U = 16;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
	float x[U];
#pragma unroll
	for (int i = 0; i < U; ++i) x[i] = a * (threadIdx.x + i);
	if (MODE == 4) asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
	if (MODE == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(x[0]), "v"(b) : "vcc");
	for (int it = 0; it < N_IT; ++it) {
#pragma unroll
		for (int i = 0; i < U; ++i) {
			if (MODE == 0) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
			if (MODE == 1) asm volatile("v_cmp_gt_f32_e64 s[40:41], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[40:41]" : "+v"(x[i]) : "v"(b) : "s40", "s41");
			if (MODE == 2) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
			if (MODE == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b) : "vcc");
			if (MODE == 4 || MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
			if (MODE == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_nop 3\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
			if (MODE == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b));
			if (MODE == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_mov_b32 %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
		}
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < U; ++i) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int n_instr) {
	const int w = 8, threads = 256, blocks = 256 * w;
	float* out; (void)hipMalloc(&out, sizeof(float) * threads * blocks);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	k<MODE><<<blocks, threads>>>(out, 1.0f, 1.0001f); (void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	k<MODE><<<blocks, threads>>>(out, 1.0f, 1.0001f);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	printf("%-64s %.2f ns per group of %d per SIMD (8 waves per SIMD)\n", name, ms * 1e6 / ((double)N_IT * U * w), n_instr);
	(void)hipFree(out);
}
int main() {
	run<0>("v_cmp -> vcc ; v_cndmask_e32 vcc", 2);
	run<1>("v_cmp_e64 -> s[40:41] ; v_cndmask_e64 s[40:41]", 2);
	run<2>("v_cmp -> vcc ; v_cndmask_e64 vcc", 2);
	run<3>("v_cmp -> vcc ; v_add_f32", 2);
	run<4>("v_cndmask_e32 vcc (vcc written once by s_mov)", 1);
	run<5>("v_cndmask_e32 vcc (vcc written once by v_cmp)", 1);
	run<7>("v_cndmask_e64 vcc (never written)", 1);
	run<6>("v_cmp -> vcc ; s_nop 3 ; v_cndmask_e32 vcc", 3);
	run<8>("v_cmp -> vcc ; v_mov ; v_cndmask_e32 vcc", 3);
	return 0;
}
