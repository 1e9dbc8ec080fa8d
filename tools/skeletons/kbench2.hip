// kbench2 — on-box skeleton study for the round-2 kernel structure (no filter arithmetic, only the
// data movement): which tile shape / schedule / cache policy gets a batch of 6000x4000 RGBA8 images
// through the chip fastest.  Every skeleton output is checkable (R byte from 3 rows up / 3 px left,
// B byte from 3 rows down / 1 px left, G and A from the pixel itself), so a race on the ring shows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/kbench2 tools/kbench2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MAXI 32
struct Imgs { const uint8_t *src[MAXI]; uint8_t *dst[MAXI]; };
struct Tile { int img, col, y0, n; };
typedef const __attribute__((address_space(4))) int *cint_p;

#define HB 3          /* source rows above an output row */
#define HA 3          /* source rows below */
#define RDEPTH 6

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_younger(int y)
{
	if (y >= 8) wait_vm<8>();
	else if (y == 7) wait_vm<7>();
	else if (y == 6) wait_vm<6>();
	else if (y == 5) wait_vm<5>();
	else if (y == 4) wait_vm<4>();
	else if (y == 3) wait_vm<3>();
	else if (y == 2) wait_vm<2>();
	else if (y == 1) wait_vm<1>();
	else wait_vm<0>();
}

// ---- linear copy over the whole batch in one launch (the ceiling) ---------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void copy_lin(Imgs im, size_t n16, int bpi)
{
	const int img = blockIdx.x / bpi, b = blockIdx.x - img * bpi;
	const u32x4 *s = (const u32x4 *)im.src[img];
	u32x4 *d = (u32x4 *)im.dst[img];
	size_t base = (size_t)b * 1024 + threadIdx.x;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; ++u) { size_t i = base + u * 256; if (i < n16) v[u] = NT ? __builtin_nontemporal_load(&s[i]) : s[i]; }
#pragma unroll
	for (int u = 0; u < 4; ++u) { size_t i = base + u * 256; if (i < n16) { if (NT) __builtin_nontemporal_store(v[u], &d[i]); else d[i] = v[u]; } }
}

struct __attribute__((packed, aligned(4))) Px4 { uint32_t a, b, c, d; };

// ---- one wave per tile, private 256-column ring (the round-1 structure) ----------------------------
template <int AUX>
__global__ __launch_bounds__(64) void strip1(Imgs im, const Tile *tiles, int W, int H, int oc, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *ring = lds + 16;
	const int lane = threadIdx.x;
	cint_p tp = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 4;
	const int img = tp[0], col = tp[1], y0 = tp[2], n = tp[3];
	const int X0 = col * oc;
	const int U0 = max(0, min(X0 - 8, W - 256));
	const uint32_t pitch = (uint32_t)W * 4;
	const int first = max(y0 - HB, 0), last = min(y0 + n - 1 + HA, H - 1);
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(im.src[img]) + (size_t)first * pitch, 0, (uint32_t)(last - first + 1) * pitch, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(im.dst[img] + (size_t)y0 * pitch, 0, (uint32_t)n * pitch, 0x00020000);
	const uint32_t vo_ld = (uint32_t)(U0 + 4 * lane) * 4;
	const int nact = min(oc, W - X0) >> 2;
	const uint32_t vo_st = lane < nact ? (uint32_t)(X0 + 4 * lane) * 4 : (uint32_t)n * pitch;
	int ld = first - 1;
	auto issue = [&]() {
		++ld;
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + (ld % K) * 1024);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)row, 16, vo_ld, (uint32_t)(ld - first) * pitch, 0, AUX);
	};
	const int px = X0 - U0 + 4 * lane;     /* my first output pixel inside the window */
	for (int yb = y0; yb < y0 + n; yb += 4) {
		const int nb = min(4, y0 + n - yb);
		const int need = min(yb + nb - 1 + HA, last), keep = max(yb - HB, first);
		{ const int target = min(min(need + RDEPTH, keep + K - 1), last); while (ld < target) issue(); }
		wait_younger(ld - need);
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			if (r < nb) {
				const int y = yb + r;
				const u32x4 idt = *reinterpret_cast<const u32x4 *>(ring + (y % K) * 1024 + px * 4);
				const Px4 pr = *reinterpret_cast<const Px4 *>(ring + (max(y - HB, 0) % K) * 1024 + (px - 3) * 4);
				const Px4 pb = *reinterpret_cast<const Px4 *>(ring + (min(y + HA, H - 1) % K) * 1024 + (px - 1) * 4);
				o[r] = u32x4{ (idt.x & 0xFF00FF00u) | (pr.a & 0xFFu) | (pb.a & 0xFF0000u), (idt.y & 0xFF00FF00u) | (pr.b & 0xFFu) | (pb.b & 0xFF0000u),
				              (idt.z & 0xFF00FF00u) | (pr.c & 0xFFu) | (pb.c & 0xFF0000u), (idt.w & 0xFF00FF00u) | (pr.d & 0xFFu) | (pb.d & 0xFF0000u) };
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (yb + 4 < y0 + n) {      /* refill the slots the next block no longer needs while this block's stores go out */
			const int nneed = min(yb + 4 + 3 + HA, last), nkeep = max(yb + 4 - HB, first);
			const int target = min(min(nneed + RDEPTH, nkeep + K - 1), last);
			while (ld < target) issue();
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (r < nb) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, vo_st + (uint32_t)(yb + r - y0) * pitch, 0, 2);
	}
}

// ---- NW waves per tile, one shared ring whose rows span the whole group: no window overlap ----------
// Each wave DMAs only its own oc columns of a row (the last wave 16 more), reads its 256-column window
// across the neighbour's part, and a workgroup barrier per block of RB rows keeps the ring consistent.
template <int AUX, int RB>
__global__ __launch_bounds__(512) void group_k(Imgs im, const Tile *tiles, int W, int H, int oc, int K, int rs_bytes, int lmargin)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *ring = lds + 16;
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int NW = blockDim.x >> 6;
	cint_p tp = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 4;
	const int img = tp[0], col = tp[1], y0 = tp[2], n = tp[3];
	const int X0g = col * NW * oc;
	const int winpx = NW * oc + 16;
	const int U0 = max(0, min(X0g - lmargin, W - winpx));
	const int X0 = X0g + w * oc;
	const uint32_t pitch = (uint32_t)W * 4;
	const int first = max(y0 - HB, 0), last = min(y0 + n - 1 + HA, H - 1);
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(im.src[img]) + (size_t)first * pitch, 0, (uint32_t)(last - first + 1) * pitch, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(im.dst[img] + (size_t)y0 * pitch, 0, (uint32_t)n * pitch, 0x00020000);
	// my share of a ring row: columns [U0 + w*oc, +oc), the last wave up to the end of the window
	int nl = (w == NW - 1) ? (winpx - w * oc) >> 2 : oc >> 2;
	if (U0 + w * oc >= W) nl = 0;
	nl = min(nl, (W - (U0 + w * oc)) >> 2);
	const bool ldl = lane < nl;
	const uint32_t vo_ld = (uint32_t)(U0 + w * oc + 4 * lane) * 4;
	const int nact = max(0, min(oc, W - X0)) >> 2;
	const uint32_t vo_st = lane < nact ? (uint32_t)(X0 + 4 * lane) * 4 : (uint32_t)n * pitch;
	int ld = first - 1;
	auto issue = [&]() {
		++ld;
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + (ld % K) * rs_bytes + w * oc * 4);
		if (ldl) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)row, 16, vo_ld, (uint32_t)(ld - first) * pitch, 0, AUX);
	};
	const int px = X0 - U0 + 4 * lane;     /* my first output pixel inside the group window */
	{
		const int need = min(y0 + min(RB, n) - 1 + HA, last);
		const int target = min(min(need + RDEPTH, first + K - 1), last);
		while (ld < target) issue();
	}
	for (int yb = y0; yb < y0 + n; yb += RB) {
		const int nb = min(RB, y0 + n - yb);
		const int need = min(yb + nb - 1 + HA, last), keep = max(yb - HB, first);
		wait_younger(ld - need);
		__builtin_amdgcn_s_barrier();
		{ const int target = min(min(need + RDEPTH, keep + K - 1), last); while (ld < target) issue(); }
		u32x4 o[RB];
#pragma unroll
		for (int r = 0; r < RB; ++r) {
			if (r < nb) {
				const int y = yb + r;
				const u32x4 idt = *reinterpret_cast<const u32x4 *>(ring + (y % K) * rs_bytes + px * 4);
				const Px4 pr = *reinterpret_cast<const Px4 *>(ring + (max(y - HB, 0) % K) * rs_bytes + (px - 3) * 4);
				const Px4 pb = *reinterpret_cast<const Px4 *>(ring + (min(y + HA, H - 1) % K) * rs_bytes + (px - 1) * 4);
				o[r] = u32x4{ (idt.x & 0xFF00FF00u) | (pr.a & 0xFFu) | (pb.a & 0xFF0000u), (idt.y & 0xFF00FF00u) | (pr.b & 0xFFu) | (pb.b & 0xFF0000u),
				              (idt.z & 0xFF00FF00u) | (pr.c & 0xFFu) | (pb.c & 0xFF0000u), (idt.w & 0xFF00FF00u) | (pr.d & 0xFFu) | (pb.d & 0xFF0000u) };
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < RB; ++r)
			if (r < nb) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, vo_st + (uint32_t)(yb + r - y0) * pitch, 0, 2);
	}
}

__global__ void fill_rand(uint32_t *p, size_t n, uint32_t seed)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 2654435761u + seed;
		x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
		p[i] = x;
	}
}

static int W = 6000, H = 4000, NI = 16;
static std::vector<void *> srcs, dsts;
static Imgs imgs;

static long check_image(int k, bool pattern)
{
	const size_t n = (size_t)W * H * 4;
	std::vector<uint8_t> s(n), d(n);
	CK(hipMemcpy(s.data(), srcs[k], n, hipMemcpyDeviceToHost));
	CK(hipMemcpy(d.data(), dsts[k], n, hipMemcpyDeviceToHost));
	long bad = 0;
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x) {
			const uint8_t *o = &d[((size_t)y * W + x) * 4];
			uint8_t e[4];
			const uint8_t *c = &s[((size_t)y * W + x) * 4];
			if (pattern) {
				const int yr = std::max(y - HB, 0), yb = std::min(y + HA, H - 1);
				const int xr = x - 3, xb = x - 1;
				e[0] = xr >= 0 ? s[((size_t)yr * W + xr) * 4] : o[0];      /* columns left of the image: whatever the pad held */
				e[2] = xb >= 0 ? s[((size_t)yb * W + xb) * 4 + 2] : o[2];
				e[1] = c[1]; e[3] = c[3];
			} else { e[0] = c[0]; e[1] = c[1]; e[2] = c[2]; e[3] = c[3]; }
			if (memcmp(e, o, 4)) { if (bad < 5) printf("   mismatch img %d y %d x %d: got %d %d %d %d want %d %d %d %d\n", k, y, x, o[0], o[1], o[2], o[3], e[0], e[1], e[2], e[3]); ++bad; }
		}
	return bad;
}

template <typename F>
static double time_it(int launches, F f)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 3; ++i) f();
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < launches; ++i) f();
	CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	CK(hipGetLastError());
	return ms * 1e3 / launches;
}

static void report(const char *name, double us_launch, long bad)
{
	const double us = us_launch / NI;
	const double gbs = 2.0 * W * H * 4 / us / 1e3;
	printf("%-78s %8.2f us/img %7.1f GB/s %5.1f%%  %s\n", name, us, gbs, gbs / 80.0, bad < 0 ? "" : (bad == 0 ? "ok" : "WRONG"));
	fflush(stdout);
}

// tile lists ------------------------------------------------------------------------------------
// ncols columns per image; a column's rows are cut into `nbig` equal big tiles covering a fraction
// `fbig` of the rows (from the top), the rest into pool tiles of `th` rows.  Big tiles come first in
// the grid (the dispatcher hands out workgroups in order), pool tiles fill the slots as they free.
static std::vector<Tile> make_tiles(int ncols, int nbig, double fbig, int th, int order)
{
	std::vector<Tile> big, pool;
	for (int i = 0; i < NI; ++i)
		for (int c = 0; c < ncols; ++c) {
			int y = 0;
			if (nbig > 0) {
				int rows = ((int)(H * fbig) / (4 * nbig)) * 4 * nbig;
				for (int b = 0; b < nbig; ++b) { big.push_back(Tile{ i, c, y, rows / nbig }); y += rows / nbig; }
			}
			for (; y < H; y += th) pool.push_back(Tile{ i, c, y, std::min(th, H - y) });
		}
	if (order == 1)       /* pool tiles row-major over the whole batch: all columns advance together */
		std::stable_sort(pool.begin(), pool.end(), [](const Tile &a, const Tile &b) { return a.y0 < b.y0; });
	if (order == 2)       /* image by image, rows from the top, columns inside a row band (round 1's order) */
		std::stable_sort(pool.begin(), pool.end(), [](const Tile &a, const Tile &b) { return a.img != b.img ? a.img < b.img : a.y0 < b.y0; });
	big.insert(big.end(), pool.begin(), pool.end());
	return big;
}

int main(int argc, char **argv)
{
	const int L = argc > 1 ? atoi(argv[1]) : 12;
	const size_t n4 = (size_t)W * H * 4;
	srcs.resize(NI); dsts.resize(NI);
	for (int i = 0; i < NI; ++i) {
		CK(hipMalloc(&srcs[i], n4)); CK(hipMalloc(&dsts[i], n4));
		hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)srcs[i], n4 / 4, 0x9E3779B9u * (i + 1));
		CK(hipMemset(dsts[i], 0, n4));
		imgs.src[i] = (const uint8_t *)srcs[i]; imgs.dst[i] = (uint8_t *)dsts[i];
	}
	for (int i = NI; i < MAXI; ++i) { imgs.src[i] = imgs.src[0]; imgs.dst[i] = imgs.dst[0]; }
	CK(hipDeviceSynchronize());
	Tile *d_tiles; CK(hipMalloc(&d_tiles, sizeof(Tile) * 200000));
	const bool verify = !getenv("KB_NOVERIFY");

	// ---- ceiling: linear copies ----
	{
		const size_t n16 = n4 / 16; const int bpi = (int)((n16 + 1023) / 1024);
		report("copy_lin default policy (16 images per launch)", time_it(L, [&] { hipLaunchKernelGGL((copy_lin<false>), dim3(bpi * NI), dim3(256), 0, 0, imgs, n16, bpi); }), -1);
		double us = time_it(L, [&] { hipLaunchKernelGGL((copy_lin<true>), dim3(bpi * NI), dim3(256), 0, 0, imgs, n16, bpi); });
		report("copy_lin nt loads + nt stores", us, verify ? check_image(NI - 1, false) : -1);
	}

	auto clear = [&] { for (int i = 0; i < NI; ++i) CK(hipMemsetAsync(dsts[i], 0, n4, 0)); };
	auto upload = [&](const std::vector<Tile> &t) { CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice)); };

	// ---- round-1 structure: one wave per tile ----
	if (!getenv("KB_SKIP_STRIP")) {
		const int oc = 240, K = 16, ncols = (W + oc - 1) / oc;
		const size_t lds = 16 + K * 1024 + 64;
		struct { int nbig; double f; int th; int order; } cfg[] = { {0, 0, 33, 2}, {5, 0.5, 32, 2}, {5, 0.6, 32, 2}, {5, 0.65, 32, 2}, {5, 0.7, 32, 2}, {5, 0.75, 32, 2}, {5, 0.8, 32, 2}, {5, 0.7, 32, 1},
		                                                        {5, 0.7, 16, 2}, {4, 0.6, 32, 2}, {4, 0.5, 32, 2}, {0, 0, 33, 2} };
		for (auto &c : cfg) for (int aux : {0}) {
			std::vector<Tile> t = make_tiles(ncols, c.nbig, c.f, c.th, c.order);
			upload(t); clear();
			char nm[160]; snprintf(nm, 160, "strip1 oc=240 K=16 big=%d f=%.2f th=%d order=%d ld_aux=%d tiles=%zu", c.nbig, c.f, c.th, c.order, aux, t.size());
			auto go = [&] { if (aux) hipLaunchKernelGGL((strip1<2>), dim3(t.size()), dim3(64), lds, 0, imgs, d_tiles, W, H, oc, K);
			                else hipLaunchKernelGGL((strip1<0>), dim3(t.size()), dim3(64), lds, 0, imgs, d_tiles, W, H, oc, K); };
			double us = time_it(L, go);
			report(nm, us, verify && aux == 0 ? check_image(NI / 2, true) : -1);
		}
	}

	// ---- shared-ring groups ----
	if (!getenv("KB_SKIP_GROUP")) {
		const int oc = 240;
		struct { int NW, K, RB, lm, nbig; double f; int th; int order; } cfg[] = {
			{5, 16, 4, 4, 0, 0, 32, 2},  {5, 16, 4, 4, 0, 0, 64, 2},   {5, 16, 4, 4, 0, 0, 128, 2},
			{5, 16, 4, 4, 6, 1.0, 32, 2}, {5, 16, 4, 4, 6, 0.9, 32, 2}, {5, 16, 4, 4, 6, 0.8, 32, 2}, {5, 16, 4, 4, 6, 0.8, 32, 1},
			{5, 16, 4, 4, 6, 0.8, 64, 1}, {5, 16, 2, 4, 6, 0.8, 32, 1}, {5, 16, 2, 4, 0, 0, 64, 2},
			{5, 12, 2, 4, 6, 0.8, 32, 1}, {5, 20, 4, 4, 3, 0.8, 32, 1}, {5, 16, 4, 16, 6, 0.8, 32, 1},
			{4, 16, 4, 4, 0, 0, 64, 2},  {4, 20, 4, 4, 0, 0, 64, 2},   {3, 16, 4, 4, 0, 0, 64, 2},   {2, 16, 4, 4, 0, 0, 64, 2},
		};
		hipFuncSetAttribute((const void *)group_k<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		hipFuncSetAttribute((const void *)group_k<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		hipFuncSetAttribute((const void *)group_k<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		hipFuncSetAttribute((const void *)group_k<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		for (auto &c : cfg) for (int aux : {0, 2}) {
			const int gw = c.NW * oc, ncols = (W + gw - 1) / gw;
			const int rs = c.NW * oc * 4 + 64;
			const size_t lds = 16 + (size_t)c.K * rs + 64;
			std::vector<Tile> t = make_tiles(ncols, c.nbig, c.f, c.th, c.order);
			upload(t); clear();
			char nm[160]; snprintf(nm, 160, "group NW=%d K=%d RB=%d lm=%d big=%d f=%.2f th=%d order=%d ld_aux=%d wgs=%zu lds=%zu", c.NW, c.K, c.RB, c.lm, c.nbig, c.f, c.th, c.order, aux, t.size(), lds);
			auto go = [&] {
				if (c.RB == 4) { if (aux) hipLaunchKernelGGL((group_k<2, 4>), dim3(t.size()), dim3(64 * c.NW), lds, 0, imgs, d_tiles, W, H, oc, c.K, rs, c.lm);
				                 else hipLaunchKernelGGL((group_k<0, 4>), dim3(t.size()), dim3(64 * c.NW), lds, 0, imgs, d_tiles, W, H, oc, c.K, rs, c.lm); }
				else           { if (aux) hipLaunchKernelGGL((group_k<2, 2>), dim3(t.size()), dim3(64 * c.NW), lds, 0, imgs, d_tiles, W, H, oc, c.K, rs, c.lm);
				                 else hipLaunchKernelGGL((group_k<0, 2>), dim3(t.size()), dim3(64 * c.NW), lds, 0, imgs, d_tiles, W, H, oc, c.K, rs, c.lm); }
			};
			double us = time_it(L, go);
			report(nm, us, verify && aux == 2 ? check_image(NI / 2 + 1, true) : -1);
		}
	}
	return 0;
}
