// tools/probes/shift64_war.hip -- does v_lshrrev_b64 on gfx950 still read a source register AFTER a following
// instruction has overwritten it (write-after-read not interlocked)?  Three sequences, results compared with C++:
//   A: v_lshrrev_b64 d, sh, v ; v_mov_b32 sh, junk           (shift amount rewritten at once)
//   B: v_lshrrev_b64 d, sh, v ; v_mov_b32 v.lo, junk ; v_mov_b32 v.hi, junk   (data rewritten at once)
//   C: v_lshrrev_b64 d, sh, v                                  (control)
// each followed by a few dependent VALU ops, as in the walk.  1..16 wavefronts per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(1024) probe(const uint64_t *in, int iters, unsigned long long *bad)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t v = in[tid & 4095u];
	unsigned long long na = 0, nb = 0, nc = 0;
	uint32_t st = tid & 15u;
	for (int it = 0; it < iters; it++) {
		const uint32_t sh = st << 2;
		const uint32_t want = (uint32_t)(v >> sh) & 15u;
		uint32_t ra, rb, rc;
		asm volatile("v_mov_b32 v100, %1\n\tv_lshrrev_b64 v[102:103], v100, %2\n\tv_mov_b32 v100, 0x3c\n\tv_and_b32 %0, 15, v102"
		             : "=&v"(ra) : "v"(sh), "v"(v) : "v100", "v102", "v103");
		{
			const uint32_t vlo = (uint32_t)v, vhi = (uint32_t)(v >> 32);
			asm volatile("v_mov_b32 v104, %2\n\tv_mov_b32 v105, %3\n\tv_lshrrev_b64 v[102:103], %1, v[104:105]\n\tv_mov_b32 v104, -1\n\tv_mov_b32 v105, -1\n\tv_and_b32 %0, 15, v102"
			             : "=&v"(rb) : "v"(sh), "v"(vlo), "v"(vhi) : "v102", "v103", "v104", "v105");
		}
		asm volatile("v_lshrrev_b64 v[102:103], %1, %2\n\tv_and_b32 %0, 15, v102"
		             : "=&v"(rc) : "v"(sh), "v"(v) : "v102", "v103");
		na += ra != want;
		nb += rb != want;
		nc += rc != want;
		st = (want + it) & 15u;
		v = v * 6364136223846793005ull + 1442695040888963407ull;
	}
	if (na) atomicAdd(&bad[0], na);
	if (nb) atomicAdd(&bad[1], nb);
	if (nc) atomicAdd(&bad[2], nc);
}

int main()
{
	uint64_t h[4096];
	for (int i = 0; i < 4096; i++) h[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
	uint64_t *d; unsigned long long *bad;
	hipMalloc(&d, sizeof h); hipMalloc(&bad, 24);
	hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
	for (int waves : {1, 4, 8, 16}) {
		hipMemset(bad, 0, 24);
		probe<<<1024, waves * 64>>>(d, 20000, bad);
		hipError_t e = hipDeviceSynchronize();
		unsigned long long r[3];
		hipMemcpy(r, bad, 24, hipMemcpyDeviceToHost);
		printf("waves/WG %2d: %s  shift amount rewritten: %llu wrong   data rewritten: %llu wrong   control: %llu wrong\n", waves, hipGetErrorString(e), r[0], r[1], r[2]);
	}
	return 0;
}
