// What a cross-stream dependency costs on the stream that carries the big kernels: the gap between two ~40 us kernels
// on stream A with, in between: nothing | an event record | a wait on an already-signalled event of stream B |
// hipStreamWaitValue32 on a value already written | record + wait (one exchange of hns_dist). hipEvents around N repetitions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(float* p, long long ticks) {
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}
__global__ void k_small(float* p) { if (p && threadIdx.x == 0) p[blockIdx.x] += 1.0f; }

int main() {
	hipStream_t a, b;
	int lo, hi;
	CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
	hipEvent_t t0, t1, er, ed;
	CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
	CK(hipEventCreateWithFlags(&er, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ed, hipEventDisableTiming));
	float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
	uint32_t* sig = nullptr;
	int can = 0;
	hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
	hipError_t se = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
	printf("CanUseStreamWaitValue %d, signal memory: %s\n", can, hipGetErrorString(se));
	if (se == hipSuccess) CK(hipMemset(sig, 0, 8));
	const int N = 200;
	const long long T = 4000;  // 40 us at 100 MHz
	for (int mode = 0; mode < 8; ++mode) {
		if ((mode == 4 || mode == 5) && (se != hipSuccess)) continue;
		float best = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			uint32_t seq = 0;
			if (sig) { CK(hipDeviceSynchronize()); CK(hipMemset(sig, 0, 8)); }
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(t0, a));
			for (int i = 0; i < N; ++i) {
				hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, a, buf, T);
				switch (mode) {
				case 0: break;
				case 1: CK(hipEventRecord(er, a)); break;
				case 2: CK(hipEventRecord(er, a)); CK(hipStreamWaitEvent(b, er, 0)); hipLaunchKernelGGL(k_small, dim3(16), dim3(64), 0, b, buf + 1024); break;
				case 3:  // one exchange as hns_dist does it: ready -> side chain -> done, waited for after the NEXT big kernel
					CK(hipEventRecord(er, a)); CK(hipStreamWaitEvent(b, er, 0));
					hipLaunchKernelGGL(k_small, dim3(16), dim3(64), 0, b, buf + 1024);
					CK(hipEventRecord(ed, b));
					hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, a, buf, T);
					CK(hipStreamWaitEvent(a, ed, 0));
					break;
				case 4:  // wait on a value that is already there
					CK(hipStreamWaitValue32(a, sig, 0, hipStreamWaitValueGte, 0xFFFFFFFFu)); break;
				case 5:  // the exchange with memory values instead of events
					++seq;
					CK(hipStreamWriteValue32(a, sig, seq, 0)); CK(hipStreamWaitValue32(b, sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
					hipLaunchKernelGGL(k_small, dim3(16), dim3(64), 0, b, buf + 1024);
					CK(hipStreamWriteValue32(b, sig + 1, seq, 0));
					hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, a, buf, T);
					CK(hipStreamWaitValue32(a, sig + 1, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
					break;
				case 6: hipLaunchKernelGGL(k_small, dim3(16), dim3(64), 0, a, buf + 1024); break;  // a small kernel on the same stream
				case 7:  // wait on an event recorded long ago on b
					if (i == 0) { CK(hipEventRecord(ed, b)); }
					CK(hipStreamWaitEvent(a, ed, 0)); break;
				}
			}
			CK(hipEventRecord(t1, a));
			CK(hipEventSynchronize(t1));
			CK(hipDeviceSynchronize());
			float ms; CK(hipEventElapsedTime(&ms, t0, t1));
			const int big = mode == 3 || mode == 5 ? 2 * N : N;
			const float per = 1e3f * ms / big - 40.0f;
			if (per < best) best = per;
		}
		const char* names[] = {"nothing", "event record", "record + other stream waits + small kernel there", "hns_dist exchange (2 big kernels, record/wait/record/wait)",
		                       "hipStreamWaitValue32 already satisfied", "exchange with Write/WaitValue32", "small kernel on the same stream", "wait on an old event of the other stream"};
		printf("mode %d  %-62s  %6.2f us over 40 per big kernel\n", mode, names[mode], best);
	}
	return 0;
}
