// unaligned_probe.hip — what gfx950 does with buffer accesses whose byte address is not a multiple of 4:
//   (1) buffer_store_dwordx3 / dwordx4 at offsets 1, 2, 3 past a dword boundary
//   (2) buffer_load_dwordx4 ... lds (LDS-DMA) from such an address
//   (3) plain buffer_load_dwordx4 from such an address
// Prints, per case and byte shift, whether the bytes that arrived are the bytes a byte-granular copy would give.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/unaligned_probe tools/skeletons/unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

__global__ void k_store(uint8_t *dst, int shift, int width)   // every lane stores 12 (or 16) bytes of a known pattern at shift + width * lane
{
	const int lane = threadIdx.x;
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 1 << 20, 0x00020000);
	uint8_t b[16];
	for (int i = 0; i < 16; ++i) b[i] = (uint8_t)(lane * 16 + i + 1);
	u32x4 v;
	memcpy(&v, b, 16);
	if (width == 12) __builtin_amdgcn_raw_buffer_store_b96(u32x3{ v.x, v.y, v.z }, rd, (uint32_t)(shift + 12 * lane), 0, 0);
	else __builtin_amdgcn_raw_buffer_store_b128(v, rd, (uint32_t)(shift + 16 * lane), 0, 0);
}

__global__ void k_load(const uint8_t *src, uint8_t *out, int shift, int use_lds)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const int lane = threadIdx.x;
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, 1 << 20, 0x00020000);
	u32x4 v;
	if (use_lds) {
#if defined(__HIP_DEVICE_COMPILE__)
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, (uint32_t)(shift + 16 * lane), 0, 0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
		__syncthreads();
		v = *reinterpret_cast<const u32x4 *>(lds + 16 * lane);
	} else {
		v = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(shift + 16 * lane), 0, 0);
	}
	memcpy(out + 16 * lane, &v, 16);
}

int main()
{
	uint8_t *d, *o;
	hipMalloc(&d, 1 << 20);
	hipMalloc(&o, 4096);
	std::vector<uint8_t> h(4096), pat(4096);
	for (int width : { 12, 16 })
		for (int shift = 0; shift < 4; ++shift) {
			hipMemset(d, 0, 4096);
			hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, 0, d, shift, width);
			hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
			int bad = 0;
			for (int lane = 0; lane < 64; ++lane)
				for (int i = 0; i < width; ++i) bad += h[shift + width * lane + i] != (uint8_t)(lane * 16 + i + 1);
			for (int i = 0; i < shift; ++i) bad += h[i] != 0;
			bad += h[shift + width * 64] != 0;
			printf("store %2d bytes per lane at byte shift %d: %s (%d wrong bytes)\n", width, shift, bad ? "WRONG" : "ok", bad);
		}
	for (int i = 0; i < 4096; ++i) pat[i] = (uint8_t)(i * 7 + 3);
	hipMemcpy(d, pat.data(), 4096, hipMemcpyHostToDevice);
	for (int use_lds : { 0, 1 })
		for (int shift = 0; shift < 4; ++shift) {
			hipMemset(o, 0, 4096);
			hipLaunchKernelGGL(k_load, dim3(1), dim3(64), 2048, 0, d, o, shift, use_lds);
			hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
			int bad = 0, as_aligned = 0;
			for (int i = 0; i < 1024; ++i) { bad += h[i] != pat[shift + i]; as_aligned += h[i] != pat[i]; }
			printf("%s 16 bytes per lane from byte shift %d: %s (%d bytes differ from the shifted source, %d from the unshifted one)\n",
			       use_lds ? "LDS-DMA" : "load   ", shift, bad ? "NOT the shifted bytes" : "ok", bad, as_aligned);
		}
	return 0;
}
