// tools/probes/unaligned_dma.hip -- does global_load_lds_dwordx4 accept a source address that is not
// 16-byte (or even 4-byte) aligned on gfx950?  64 lanes each fetch 16 bytes from src + shift + 16 * lane
// straight into LDS; the LDS tile is copied out and compared with the bytes a plain memcpy gives.
//   hipcc --offload-arch=gfx950 -O2 unaligned_dma.hip -o _build/unaligned_dma && _build/unaligned_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__global__ void probe(const unsigned char *src, unsigned shift, unsigned char *out)
{
	__shared__ __align__(16) unsigned char tile[1024];
	const unsigned lane = threadIdx.x;
	__builtin_amdgcn_global_load_lds((glb_void_t *)(src + shift + 16u * lane), (lds_void_t *)tile, 16, 0, 0);
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__asm__ volatile("" ::: "memory");
	__syncthreads();
	for (unsigned k = 0; k < 16; k++) out[lane * 16u + k] = tile[lane * 16u + k];
}

int main()
{
	std::vector<unsigned char> h(4096);
	for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)((i * 131u + (i >> 8) * 17u + 7u) & 0xff);
	unsigned char *d_src, *d_out;
	hipMalloc(&d_src, h.size());
	hipMalloc(&d_out, 1024);
	hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
	int bad = 0;
	for (unsigned shift = 0; shift < 32; shift++) {
		std::vector<unsigned char> got(1024);
		hipMemset(d_out, 0, 1024);
		probe<<<1, 64>>>(d_src, shift, d_out);
		hipError_t e = hipDeviceSynchronize();
		if (e != hipSuccess) { printf("shift %2u: %s\n", shift, hipGetErrorString(e)); return 2; }
		hipMemcpy(got.data(), d_out, 1024, hipMemcpyDeviceToHost);
		const bool ok = memcmp(got.data(), h.data() + shift, 1024) == 0;
		printf("shift %2u: %s\n", shift, ok ? "ok" : "MISMATCH");
		bad += !ok;
	}
	printf("unaligned global_load_lds_dwordx4: %s\n", bad ? "NOT supported as is" : "supported");
	return bad != 0;
}
