// kbench5 — RGB8 store shapes: every wave moves rows of `ow` bytes (12 bytes per lane, ow/12 lanes) at
// pitch 18000 through an LDS-DMA ring (one 16-byte-per-lane DMA of 768 bytes per row, 48 lanes).
//   ow = 720: round 1/2's RGB strips (240 px; boundaries only 16-byte aligned)
//   ow = 768: 256-px strips (64-byte aligned boundaries, all 64 lanes store)
//   ow = 576: 192-px strips (64-byte aligned, 48 lanes store)
// and the RGBA shape (1024-byte DMA, 960-byte rows of 16 bytes per lane, pitch 24000) for reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef const __attribute__((address_space(4))) unsigned long long *cull_p;
typedef const __attribute__((address_space(4))) int *cint_p;
struct Tile { unsigned long long rbase, wbase; int rstride, wstride, n, pad; };
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
struct __attribute__((packed, aligned(4))) P3 { uint32_t a, b, c; };

// BPL: bytes stored per lane (12 or 16); dma_lanes x 16 bytes loaded per row; st_lanes lanes store
template <int BPL>
__global__ __launch_bounds__(64) void pipe(const Tile *tiles, int dma_lanes, int st_lanes)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int P = 3, K = 16;
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	const int nblk = n / 4;
	auto load_block = [&](int b) {
		const int bb = min(b, nblk - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (lane < dma_lanes)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
	};
#pragma unroll
	for (int b = 0; b < P; ++b) load_block(b);
	const uint32_t vo = lane < st_lanes ? lane * BPL : 0x7fffffffu;
	for (int b = 0; b < nblk; ++b) {
		load_block(b + P);
		if (b >= P) wait_vm<8 * P>(); else wait_vm<4 * P>();
		if (BPL == 16) {
			u32x4 o[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, vo + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
		} else {
			P3 o[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const P3 *>(ring + ((4 * b + r) % K) * 1024 + lane * 12);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b96(u32x3{ o[r].a, o[r].b, o[r].c }, rd, vo + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
		}
	}
}
// two pieces per row: the wave's row is 2 x (dma_lanes x 16) bytes in, 2 x (st_lanes x BPL) bytes out
template <int BPL, int K>
__global__ __launch_bounds__(64) void pipe2x(const Tile *tiles, int dma_lanes, int st_lanes, int in_piece, int out_piece)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int P = K / 4 - 1;
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 2048u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 2048u, 0x00020000);
	const int nblk = n / 4;
	auto load_block = [&](int b) {
		const int bb = min(b, nblk - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			if (lane < dma_lanes) {
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 2048), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 2048 + 1024), 16, lane * 16 + in_piece, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
			}
		}
	};
#pragma unroll
	for (int b = 0; b < P; ++b) load_block(b);
	const uint32_t vo = lane < st_lanes ? lane * BPL : 0x7fffffffu;
	for (int b = 0; b < nblk; ++b) {
		load_block(b + P);
		if (b >= P) wait_vm<16 * P>(); else wait_vm<8 * P>();
		u32x4 o[4][2];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			o[r][0] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 2048 + lane * 16);
			o[r][1] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 2048 + 1024 + lane * 16);
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			if (BPL == 16) {
				__builtin_amdgcn_raw_buffer_store_b128(o[r][0], rd, vo + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
				__builtin_amdgcn_raw_buffer_store_b128(o[r][1], rd, vo + out_piece + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
			} else {
				__builtin_amdgcn_raw_buffer_store_b96(u32x3{ o[r][0].x, o[r][0].y, o[r][0].z }, rd, vo + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
				__builtin_amdgcn_raw_buffer_store_b96(u32x3{ o[r][1].x, o[r][1].y, o[r][1].z }, rd, vo + out_piece + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
			}
		}
	}
}
__global__ void fill_rand(uint32_t *p, size_t n, uint32_t seed)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; p[i] = x; }
}
template <typename F> static double time_it(int launches, F f)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 2; ++i) f();
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < launches; ++i) f();
	CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
	return ms * 1e3 / launches;
}
int main()
{
	const int NI = 32, H = 4000;
	const size_t cap = (size_t)24000 * H + 4096;
	std::vector<char *> srcs(NI), dsts(NI);
	for (int i = 0; i < NI; ++i) { CK(hipMalloc((void **)&srcs[i], cap)); CK(hipMalloc((void **)&dsts[i], cap));
		hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)srcs[i], cap / 4, 0x9E3779B9u * (i + 1)); }
	CK(hipDeviceSynchronize());
	Tile *d_tiles; CK(hipMalloc(&d_tiles, sizeof(Tile) * 400000));
	struct { const char *name; int bpl, pitch, ow, dma_lanes, rd_off; } cfg[] = {
		{ "RGBA 240 px: 1024 B in, 960 B out (60 lanes x 16), pitch 24000", 16, 24000, 960, 64, 0 },
		{ "RGB  240 px:  768 B in, 720 B out (60 lanes x 12), pitch 18000", 12, 18000, 720, 48, 0 },
		{ "RGB  240 px:  736 B in (46 lanes), 720 B out",                   12, 18000, 720, 46, 0 },
		{ "RGB  256 px:  816 B in (51 lanes), 768 B out (64 lanes x 12)",   12, 18000, 768, 51, 0 },
		{ "RGB  192 px:  624 B in (39 lanes), 576 B out (48 lanes x 12)",   12, 18000, 576, 39, 0 },
		{ "RGB  320 px: 1008 B in (63 lanes), 960 B out (needs 80 lanes: 2 stores; here 64 x 12 = 768 stored)", 12, 18000, 960, 63, 0 },
	};
	for (int th : {32}) for (auto &c : cfg) {
		const int nstrips = c.pitch / c.ow;
		std::vector<Tile> t;
		for (int i = 0; i < NI; ++i) for (int y = 0; y + th <= H; y += th) for (int s = 0; s < nstrips; ++s) {
			Tile q; const size_t off = (size_t)y * c.pitch + (size_t)s * c.ow;
			q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
			q.rstride = c.pitch; q.wstride = c.pitch; q.n = th; q.pad = 0; t.push_back(q);
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		const int st_lanes = std::min(64, c.ow / c.bpl);
		const double bytes = 2.0 * t.size() * th * (double)(st_lanes * c.bpl);
		double us = (c.bpl == 16) ? time_it(10, [&] { hipLaunchKernelGGL((pipe<16>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles, c.dma_lanes, st_lanes); })
		                          : time_it(10, [&] { hipLaunchKernelGGL((pipe<12>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles, c.dma_lanes, st_lanes); });
		printf("th=%d %-100s tiles=%6zu : %6.2f us per image-equivalent, %5.1f%% of 8 TB/s (bytes stored x 2)\n", th, c.name, t.size(), us / NI, bytes / us / 8e4);
		fflush(stdout);
	}
	// two pieces per row
	struct { const char *name; int bpl, pitch, ow, dma_lanes, st_lanes, in_piece, out_piece; } c2[] = {
		{ "RGBA 480 px: 2 x 1024 B in, 2 x 960 B out", 16, 24000, 1920, 64, 60, 1024, 960 },
		{ "RGBA 496 px: 2 x 1024 B in, 2 x 992 B out (62 lanes)", 16, 24000, 1984, 64, 62, 1024, 992 },
		{ "RGB  480 px: 2 x 752 B in (47 lanes), 2 x 720 B out", 12, 18000, 1440, 47, 60, 752, 720 },
		{ "RGB  496 px: 2 x 768 B in (48 lanes), 2 x 744 B out (62 lanes)", 12, 18000, 1488, 48, 62, 768, 744 },
	};
	for (int K : {8, 16}) for (auto &c : c2) {
		const int th = 32;
		const int nstrips = c.pitch / c.ow;
		std::vector<Tile> t;
		for (int i = 0; i < NI; ++i) for (int y = 0; y + th <= H; y += th) for (int s = 0; s < nstrips; ++s) {
			Tile q; const size_t off = (size_t)y * c.pitch + (size_t)s * c.ow;
			q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
			q.rstride = c.pitch; q.wstride = c.pitch; q.n = th; q.pad = 0; t.push_back(q);
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		const double bytes = 2.0 * t.size() * th * (double)(2 * c.st_lanes * c.bpl);
		double us;
		if (K == 8) us = (c.bpl == 16) ? time_it(10, [&] { hipLaunchKernelGGL((pipe2x<16, 8>), dim3(t.size()), dim3(64), 8 * 2048, 0, d_tiles, c.dma_lanes, c.st_lanes, c.in_piece, c.out_piece); })
		                               : time_it(10, [&] { hipLaunchKernelGGL((pipe2x<12, 8>), dim3(t.size()), dim3(64), 8 * 2048, 0, d_tiles, c.dma_lanes, c.st_lanes, c.in_piece, c.out_piece); });
		else        us = (c.bpl == 16) ? time_it(10, [&] { hipLaunchKernelGGL((pipe2x<16, 16>), dim3(t.size()), dim3(64), 16 * 2048, 0, d_tiles, c.dma_lanes, c.st_lanes, c.in_piece, c.out_piece); })
		                               : time_it(10, [&] { hipLaunchKernelGGL((pipe2x<12, 16>), dim3(t.size()), dim3(64), 16 * 2048, 0, d_tiles, c.dma_lanes, c.st_lanes, c.in_piece, c.out_piece); });
		printf("K=%2d %-70s tiles=%6zu : %6.2f us per image-equivalent, %5.1f%% of 8 TB/s\n", K, c.name, t.size(), us / NI, bytes / us / 8e4);
		fflush(stdout);
	}
	return 0;
}
