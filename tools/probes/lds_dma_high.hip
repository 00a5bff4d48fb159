// tools/probes/lds_dma_high.hip -- is an LDS-DMA tile at a HIGH LDS address (beyond ~108 KiB of the
// workgroup's allocation) always complete once the issuing wave's vmcnt(0) has passed?
// 8 waves per workgroup; wave w owns a tile at  table_bytes + w * wave_stride; per iteration it fetches 8 KiB
// (8 x global_load_lds_dwordx4) from one of two source buffers (alternating, so a stale tile is visible),
// waits vmcnt(0), reads the tile back with ds_read_b128 and compares with the source.  Mismatches are
// counted per wave.
//   hipcc --offload-arch=gfx950 -O2 lds_dma_high.hip -o _build/lds_dma_high && _build/lds_dma_high
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) probe(const unsigned char *a, const unsigned char *b, uint32_t table_bytes, uint32_t wave_stride,
                                             int iters, int extra_wait, unsigned long long *bad)
{
	extern __shared__ __align__(16) unsigned char lds[];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned char *stg = lds + table_bytes + wave * wave_stride;
	// something to keep the low part of LDS busy, like a table
	for (uint32_t i = threadIdx.x * 16u; i < table_bytes; i += blockDim.x * 16u) *reinterpret_cast<u32x4 *>(lds + i) = u32x4{i, i, i, i};
	__syncthreads();
	unsigned long long nbad = 0;
	for (int it = 0; it < iters; it++) {
		const unsigned char *src = ((it & 1) ? b : a) + ((size_t)blockIdx.x * 8u + wave) * 8192u;
#pragma unroll
		for (uint32_t j = 0; j < 8; j++)
			__builtin_amdgcn_global_load_lds((glb_void_t *)(src + j * 1024u + lane * 16u), (lds_void_t *)(stg + j * 1024u), 16, 0, 0);
		__builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
		__asm__ volatile("" ::: "memory");
		if (extra_wait) __builtin_amdgcn_s_sleep(2);
		u32x4 w[8];
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) w[j] = *reinterpret_cast<const u32x4 *>(stg + j * 1024u + lane * 16u);
		__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
		__asm__ volatile("" ::: "memory");
#pragma unroll
		for (uint32_t j = 0; j < 8; j++) {
			const u32x4 e = *reinterpret_cast<const u32x4 *>(src + j * 1024u + lane * 16u);
			if (w[j].x != e.x || w[j].y != e.y || w[j].z != e.z || w[j].w != e.w) nbad++;
		}
	}
	if (nbad) atomicAdd(&bad[wave], nbad);
}

int main()
{
	const int nblk = 256, iters = 2000;
	const size_t bytes = (size_t)nblk * 8 * 8192;
	std::vector<uint32_t> ha(bytes / 4), hb(bytes / 4);
	for (size_t i = 0; i < ha.size(); i++) { ha[i] = (uint32_t)(i * 2654435761u) ^ 0x1234567u; hb[i] = ~ha[i] + (uint32_t)i; }
	unsigned char *a, *b; unsigned long long *bad;
	hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&bad, 8 * sizeof(*bad));
	hipMemcpy(a, ha.data(), bytes, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), bytes, hipMemcpyHostToDevice);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	const uint32_t cfg[][3] = { {65536, 11264, 0}, {65536, 11264, 2}, {65536, 8192, 0}, {1024, 11264, 0}, {65536, 10240, 0}, {73728, 11264, 0} };
	for (auto &c : cfg) {
		const uint32_t lds = c[0] + 8 * c[1];
		hipMemset(bad, 0, 8 * sizeof(*bad));
		probe<<<nblk, 512, lds>>>(a, b, c[0], c[1], iters, (int)c[2], bad);
		hipError_t e = hipDeviceSynchronize();
		unsigned long long h[8];
		hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
		printf("table %6u  wave stride %5u  lds %6u  sleep %u : %s  bad per wave:", c[0], c[1], lds, c[2], hipGetErrorString(e));
		for (int w = 0; w < 8; w++) printf(" %llu", h[w]);
		printf("\n");
	}
	return 0;
}
