// st96bench — does the write path care whether a lane stores 12 or 16 bytes?  Streams 1 GiB through three copy kernels:
// 16-byte loads + 16-byte nt stores; 12-byte loads + 12-byte nt stores (what k_roll_rgb's rows look like: 64 lanes x 12
// consecutive bytes); 16-byte loads + 12-byte stores of the first three dwords (store path alone).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/st96bench tools/skeletons/st96bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *src, uint8_t *dst, size_t chunk, int iters)
{
	// each block owns `chunk` bytes; per iteration a wave moves 64 x (MODE ? 12 : 16) consecutive bytes
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint8_t *s = src + (size_t)blockIdx.x * chunk;
	uint8_t *d = dst + (size_t)blockIdx.x * chunk;
	const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(s), 0, (uint32_t)chunk, 0x00020000);
	const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(d, 0, (uint32_t)chunk, 0x00020000);
	constexpr uint32_t LB = (MODE == 1) ? 12 : 16, SBY = (MODE == 0 || MODE >= 3) ? 16 : 12;
	const uint32_t row = 64 * ((MODE == 5) ? 15 : (MODE == 0 || MODE >= 3) ? 16 : 12);           // bytes a wave moves per iteration
	uint32_t off = wave * row;
	unsigned acc = 0;
	for (int it = 0; it < iters; ++it, off += 4 * row) {
		u32x4 v;
		if (MODE == 5) {      // round 4: 5 RGB8 pixels per lane: 16-byte loads and stores at a 15-byte lane stride (unaligned, one byte shared with the next lane)
			v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 15, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b128(v, rd, off + lane * 15, 0, 2);
			continue;
		}
		if (MODE == 4) { v = u32x4{ (unsigned)it, off, 1u, 2u }; __builtin_amdgcn_raw_buffer_store_b128(v, rd, off + lane * 16, 0, 2); continue; }     // stores only
		if (MODE == 3) { v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 16, 0, 0); acc ^= v.x ^ v.y ^ v.z ^ v.w; continue; }           // loads only
		if (LB == 16) v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * (MODE == 0 ? 16 : 12), 0, 0);   // MODE 2: 16-byte loads at 12-byte stride (unaligned, overlapping)
		else { u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, off + lane * 12, 0, 0); v = u32x4{ t.x, t.y, t.z, 0 }; }
		if (SBY == 16) __builtin_amdgcn_raw_buffer_store_b128(v, rd, off + lane * 16, 0, 2);
		else __builtin_amdgcn_raw_buffer_store_b96(u32x3{ v.x, v.y, v.z }, rd, off + lane * 12, 0, 2);
	}
	if (MODE == 3 && acc == 0x12345678u) d[0] = 1;
}

template <int MODE> int run(const char *name, const uint8_t *s, uint8_t *d, size_t total)
{
	const int blocks = 2048;
	const size_t chunk = total / blocks;
	const uint32_t row = 64 * ((MODE == 5) ? 15 : (MODE == 0 || MODE >= 3) ? 16 : 12) * 4;
	const int iters = (int)(chunk / row) - 1;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, s, d, chunk, iters);
	CK(hipEventRecord(e0, 0));
	const int reps = 20;
	for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, s, d, chunk, iters);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double bytes = ((MODE == 3 || MODE == 4) ? 1.0 : 2.0) * (double)blocks * iters * row;
	printf("%-44s %8.1f GB/s  (%.1f %% of 8 TB/s)\n", name, bytes * reps / (ms * 1e-3) / 1e9, bytes * reps / (ms * 1e-3) / 1e9 / 80.0);
	return 0;
}
int main()
{
	const size_t total = (size_t)1 << 30;
	uint8_t *s, *d; CK(hipMalloc(&s, total)); CK(hipMalloc(&d, total));
	CK(hipMemset(s, 1, total)); CK(hipMemset(d, 0, total));
	for (int round = 0; round < 2; ++round) {
		if (run<0>("16-byte loads, 16-byte nt stores", s, d, total)) return 1;
		if (run<1>("12-byte loads, 12-byte nt stores", s, d, total)) return 1;
		if (run<2>("16-byte loads (12-byte stride), 12-byte stores", s, d, total)) return 1;
		if (run<3>("16-byte loads only (bytes read)", s, d, total)) return 1;
		if (run<4>("16-byte nt stores only (bytes written)", s, d, total)) return 1;
		if (run<5>("16-byte loads and nt stores at a 15-byte lane stride", s, d, total)) return 1;
	}
	return 0;
}
