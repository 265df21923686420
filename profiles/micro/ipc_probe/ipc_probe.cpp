// Feasibility probe for a one-sided halo transport: two PROCESSES on one device, device memory shared with hipIpc*, a kernel of
// process B writes a payload into A's buffer and raises a flag there, a kernel of A that is already running waits for the flag
// (bounded spin) and checks the payload; then a ping-pong of flags between the two processes' kernels (round-trip latency).
// fork() happens before either child touches HIP.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)

__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ bool wait_ge(const uint32_t* flag, uint32_t v, long long limit_ticks) {
	const long long t0 = wall_clock64();
	while (ld_sys(flag) < v) {
		if (wall_clock64() - t0 > limit_ticks) return false;
		__builtin_amdgcn_s_sleep(4);
	}
	return true;
}

// A: wait for flag[0] >= 1, then sum the payload
__global__ void k_wait_and_sum(const uint32_t* flag, const float* buf, int n, float* out, int* timed_out) {
	__shared__ float red[256];
	__shared__ int ok;
	if (threadIdx.x == 0) ok = wait_ge(flag, 1, 300000000LL) ? 1 : 0;  // 3 s at 100 MHz
	__syncthreads();
	if (!ok) { if (threadIdx.x == 0) *timed_out = 1; return; }
	float s = 0.0f;
	for (int i = threadIdx.x; i < n; i += blockDim.x) s += __builtin_nontemporal_load(buf + i);
	red[threadIdx.x] = s;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
	if (threadIdx.x == 0) *out = red[0];
}
// B: write the payload into A's buffer, then raise A's flag (last block)
__global__ void k_put(float* remote, int n, uint32_t* remote_flag, unsigned* local_count) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) remote[i] = 1.0f;
	__threadfence_system();
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned c = atomicAdd(local_count, 1u);
		if (c == gridDim.x - 1) { __threadfence_system(); st_sys(remote_flag, 1u); }
	}
}
// ping-pong: `me` raises the other's flag to 2k+me and waits for its own to reach the answer, `rounds` times
__global__ void k_pingpong(uint32_t* mine, uint32_t* theirs, int me, int rounds, long long* ticks, int* timed_out) {
	const long long t0 = wall_clock64();
	for (int k = 1; k <= rounds; ++k) {
		if (me == 0) {
			st_sys(theirs, (uint32_t)(10 + k));
			if (!wait_ge(mine, (uint32_t)(10 + k), 300000000LL)) { *timed_out = 1; return; }
		} else {
			if (!wait_ge(mine, (uint32_t)(10 + k), 300000000LL)) { *timed_out = 1; return; }
			st_sys(theirs, (uint32_t)(10 + k));
		}
	}
	*ticks = wall_clock64() - t0;
}

struct Handles { hipIpcMemHandle_t buf, flag; int fine; };

static void xwrite(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) _exit(4); }
static void xread(int fd, void* p, size_t n) { size_t got = 0; while (got < n) { ssize_t r = read(fd, (char*)p + got, n - got); if (r <= 0) _exit(5); got += (size_t)r; } }

static int child(int me, int rd, int wr) {
	const int N = 1 << 18;
	float* buf; uint32_t* flag; int fine = 1;
	CK(hipSetDevice(0));
	CK(hipMalloc(&buf, N * sizeof(float)));
	if (hipExtMallocWithFlags((void**)&flag, 4096, hipDeviceMallocFinegrained) != hipSuccess) { fine = 0; (void)hipGetLastError(); CK(hipMalloc(&flag, 4096)); }
	CK(hipMemset(buf, 0, N * sizeof(float)));
	CK(hipMemset(flag, 0, 4096));
	CK(hipDeviceSynchronize());
	Handles mine, theirs;
	mine.fine = fine;
	CK(hipIpcGetMemHandle(&mine.buf, buf));
	hipError_t fe = hipIpcGetMemHandle(&mine.flag, flag);
	if (fe != hipSuccess && fine) {  // fine-grained memory not exportable: fall back to plain device memory for the flags
		printf("[%d] hipIpcGetMemHandle(fine-grained) -> %s; using hipMalloc flags\n", me, hipGetErrorString(fe));
		(void)hipGetLastError();
		CK(hipMalloc(&flag, 4096)); CK(hipMemset(flag, 0, 4096)); CK(hipDeviceSynchronize());
		mine.fine = 0;
		CK(hipIpcGetMemHandle(&mine.flag, flag));
	} else if (fe != hipSuccess) CK(fe);
	xwrite(wr, &mine, sizeof(mine));
	xread(rd, &theirs, sizeof(theirs));
	float* rbuf; uint32_t* rflag;
	CK(hipIpcOpenMemHandle((void**)&rbuf, theirs.buf, hipIpcMemLazyEnablePeerAccess));
	CK(hipIpcOpenMemHandle((void**)&rflag, theirs.flag, hipIpcMemLazyEnablePeerAccess));
	printf("[%d] handles open (flags fine-grained: mine %d theirs %d)\n", me, mine.fine, theirs.fine); fflush(stdout);
	int* timed_out; float* out; long long* ticks; unsigned* count;
	CK(hipHostMalloc(&timed_out, 64)); *timed_out = 0;
	CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&count, 64)); CK(hipMemset(count, 0, 64)); CK(hipMemset(ticks, 0, 64));
	if (me == 0) {
		hipLaunchKernelGGL(k_wait_and_sum, dim3(1), dim3(256), 0, 0, flag, buf, N, out, timed_out);
		char go = 1; xwrite(wr, &go, 1);  // the waiting kernel is in flight: B may put
		CK(hipDeviceSynchronize());
		float s = 0; CK(hipMemcpy(&s, out, 4, hipMemcpyDeviceToHost));
		printf("[0] payload sum %.0f (want %d), timed out %d\n", s, N, *timed_out); fflush(stdout);
	} else {
		char go; xread(rd, &go, 1);
		usleep(20000);
		hipLaunchKernelGGL(k_put, dim3(64), dim3(256), 0, 0, rbuf, N, rflag, count);
		CK(hipDeviceSynchronize());
	}
	// ping-pong on flag word 1
	const int rounds = 1000;
	hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(1), 0, 0, flag + 1, rflag + 1, me, rounds, ticks, timed_out);
	CK(hipDeviceSynchronize());
	long long t = 0; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
	printf("[%d] ping-pong: %d round trips, %.2f us each, timed out %d\n", me, rounds, t / 100.0 / rounds, *timed_out); fflush(stdout);
	CK(hipIpcCloseMemHandle(rbuf)); CK(hipIpcCloseMemHandle(rflag));
	return *timed_out ? 6 : 0;
}

int main() {
	int ab[2], ba[2];
	if (pipe(ab) || pipe(ba)) return 1;
	pid_t a = fork();
	if (a == 0) _exit(child(0, ba[0], ab[1]));
	pid_t b = fork();
	if (b == 0) _exit(child(1, ab[0], ba[1]));
	int sa = 0, sb = 0;
	waitpid(a, &sa, 0); waitpid(b, &sb, 0);
	printf("exit codes %d %d\n", WEXITSTATUS(sa), WEXITSTATUS(sb));
	return WEXITSTATUS(sa) | WEXITSTATUS(sb);
}
