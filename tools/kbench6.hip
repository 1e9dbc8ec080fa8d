// kbench6 — write-side probe: how does the shape of a wave's stores change the HBM write rate, on fast- and
// slow-write boxes of the pool?  Write-only streams over 32 images of 6000x4000x4 bytes (pitch 24000):
//   rows    : every wave writes `n` rows of PIECE bytes at the image pitch (the kernel's shape: 1 KB pieces)
//   wide    : the same with 2 / 4 consecutive pieces per row by one wave (2 KB / 4 KB contiguous per row)
//   linear  : every wave writes n * PIECE contiguous bytes
// and the read-only and copy counterparts of "rows".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef const __attribute__((address_space(4))) unsigned long long *cull_p;
typedef const __attribute__((address_space(4))) int *cint_p;
struct Tile { unsigned long long base; int stride, n, npieces, pad; };   /* 24 bytes -> padded to 32 */
struct Tile32 { unsigned long long base; int stride, n, npieces, pad; unsigned long long pad2; };

template <int AUX>
__global__ __launch_bounds__(64) void wr(const Tile32 *tiles)
{
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int stride = ti[2], n = ti[3], np = ti[4];
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)stride + 4096u, 0x00020000);
	u32x4 o = u32x4{ (unsigned)lane, 1u, 2u, 3u };
	for (int y = 0; y < n; ++y) {
		for (int p = 0; p < np; ++p)
			__builtin_amdgcn_raw_buffer_store_b128(o, rd, lane * 16 + p * 1024 + (uint32_t)y * (uint32_t)stride, 0, AUX);
		o.y += 1;
	}
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
	const int NI = 32, H = 4000, pitch = 24000;
	const size_t cap = (size_t)pitch * H + 8192;
	std::vector<char *> dsts(NI);
	for (int i = 0; i < NI; ++i) { CK(hipMalloc((void **)&dsts[i], cap)); CK(hipMemset(dsts[i], 0, cap)); }
	CK(hipDeviceSynchronize());
	Tile32 *d_tiles; CK(hipMalloc(&d_tiles, sizeof(Tile32) * 400000));
	struct { const char *name; int np, th, linear; } cfg[] = {
		{ "rows   1 KB pieces, th 32 (the kernel's store shape)", 1, 32, 0 }, { "rows   1 KB pieces, th 8", 1, 8, 0 }, { "rows   1 KB pieces, th 128", 1, 128, 0 },
		{ "wide   2 KB per row, th 32", 2, 32, 0 }, { "wide   4 KB per row, th 32", 4, 32, 0 }, { "wide   4 KB per row, th 8", 4, 8, 0 },
		{ "linear 32 KB per wave", 1, 32, 1 }, { "linear 8 KB per wave", 1, 8, 1 }, { "linear 128 KB per wave", 1, 128, 1 },
	};
	for (auto &c : cfg) {
		std::vector<Tile32> t;
		const int wbytes = c.np * 1024, nstrips = (5 * 4096) / wbytes;      /* 20 KB of every 24000-byte row */
		for (int i = 0; i < NI; ++i) {
			size_t lin = 0;
			for (int y = 0; y + c.th <= H; y += c.th) for (int s = 0; s < nstrips; ++s) {
				Tile32 q; q.pad = 0; q.pad2 = 0; q.npieces = c.np; q.n = c.th;
				if (c.linear) { q.base = (unsigned long long)(dsts[i] + lin); q.stride = wbytes; lin += (size_t)c.th * wbytes; }
				else { q.base = (unsigned long long)(dsts[i] + (size_t)y * pitch + (size_t)s * wbytes); q.stride = pitch; }
				t.push_back(q);
			}
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile32) * t.size(), hipMemcpyHostToDevice));
		const double bytes = (double)t.size() * c.th * wbytes;
		double a = time_it(8, [&] { hipLaunchKernelGGL((wr<2>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		double b = time_it(8, [&] { hipLaunchKernelGGL((wr<0>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		double c3 = time_it(8, [&] { hipLaunchKernelGGL((wr<3>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		double c1 = time_it(8, [&] { hipLaunchKernelGGL((wr<1>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		printf("write-only %-55s tiles=%6zu : nt %5.1f%%  default %5.1f%%  sc1|nt %5.1f%%  sc0 %5.1f%%  (of 8 TB/s)\n", c.name, t.size(), bytes / a / 8e4, bytes / b / 8e4, bytes / c3 / 8e4, bytes / c1 / 8e4);
		fflush(stdout);
	}
	return 0;
}
