// tools/probes/shift64_waw.hip -- on gfx950, can the high dword of a v_lshrrev_b64 result land AFTER a
// following 32-bit VALU write to the same register (a write-after-write hazard that is not interlocked)?
//   seq A:  v_lshrrev_b64 v[lo:hi], sh, v[x:y]  ;  v_and_b32 hi, 15, lo        (hi is rewritten at once)
//   seq B:  the same with the v_and result in a third register (no reuse)
// Both should give (data >> sh) & 15.  Run with 1..16 waves per workgroup, many iterations, count lanes
// whose result differs from the plain C++ computation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(1024) probe(const uint64_t *in, int iters, unsigned long long *badA, unsigned long long *badB)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = in[tid & 4095u];
	unsigned long long na = 0, nb = 0;
	uint32_t st = tid & 15u;
	for (int it = 0; it < iters; it++) {
		const uint32_t sh = st << 2;
		uint32_t ra, rb;
		asm volatile("v_lshrrev_b64 v[100:101], %1, %2\n\tv_and_b32 v101, 15, v100\n\tv_mov_b32 %0, v101"
		             : "=v"(ra) : "v"(sh), "v"(v) : "v100", "v101");
		asm volatile("v_lshrrev_b64 v[102:103], %1, %2\n\tv_and_b32 %0, 15, v102"
		             : "=&v"(rb) : "v"(sh), "v"(v) : "v102", "v103");
		const uint32_t want = (uint32_t)(v >> sh) & 15u;
		na += ra != want;
		nb += rb != want;
		st = (want + it) & 15u;
		v = v * 6364136223846793005ull + 1442695040888963407ull;
	}
	if (na) atomicAdd(badA, na);
	if (nb) atomicAdd(badB, nb);
}

int main()
{
	uint64_t h[4096];
	for (int i = 0; i < 4096; i++) h[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
	uint64_t *d; unsigned long long *bad;
	hipMalloc(&d, sizeof h); hipMalloc(&bad, 16);
	hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
	for (int waves : {1, 4, 8, 16}) {
		hipMemset(bad, 0, 16);
		probe<<<1024, waves * 64>>>(d, 20000, bad, bad + 1);
		hipError_t e = hipDeviceSynchronize();
		unsigned long long r[2];
		hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
		printf("waves/WG %2d: %s  reuse-high-half mismatches %llu   separate-register mismatches %llu\n", waves, hipGetErrorString(e), r[0], r[1]);
	}
	return 0;
}
