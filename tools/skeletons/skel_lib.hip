// skel_lib — kbench2's one-wave-per-tile skeleton as a shared library, so tools/r2_exp22.py can time it on the
// SAME buffers as the real kernel (allocation modes, profiles/NOTES.md).  Original header:
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


// ---- C entry points ---------------------------------------------------------------------------------
#include <map>
#include <tuple>
static std::map<std::tuple<int,int,int,int,int,int,int,int>, std::pair<Tile *, int>> g_lists;

static std::pair<Tile *, int> tile_list(int ni, int W, int H, int oc, int nbig, int fbig_pm, int th, int order)
{
	auto key = std::make_tuple(ni, W, H, oc, nbig, fbig_pm, th, order);
	auto it = g_lists.find(key);
	if (it != g_lists.end()) return it->second;
	const int ncols = (W + oc - 1) / oc;
	std::vector<Tile> big, pool;
	for (int i = 0; i < ni; ++i)
		for (int c = 0; c < ncols; ++c) {
			int y = 0;
			if (nbig > 0) {
				int rows = ((int)((long)H * fbig_pm / 1000) / (4 * nbig)) * 4 * nbig;
				for (int b = 0; b < nbig; ++b) { big.push_back(Tile{ i, c, y, rows / nbig }); y += rows / nbig; }
			}
			for (; y < H; y += th) pool.push_back(Tile{ i, c, y, std::min(th, H - y) });
		}
	if (order == 1) std::stable_sort(pool.begin(), pool.end(), [](const Tile &a, const Tile &b) { return a.y0 < b.y0; });
	if (order == 2) std::stable_sort(pool.begin(), pool.end(), [](const Tile &a, const Tile &b) { return a.img != b.img ? a.img < b.img : a.y0 < b.y0; });
	if (order == 3) {      /* order 2 plus the library's XCD remap inside each image: XCD x gets a contiguous run of the image's tiles */
		std::stable_sort(pool.begin(), pool.end(), [](const Tile &a, const Tile &b) { return a.img != b.img ? a.img < b.img : a.y0 < b.y0; });
		const size_t per = pool.size() / ni;
		if (per % 8 == 0 && !big.size()) {
			std::vector<Tile> r(pool.size());
			for (int i = 0; i < ni; ++i)
				for (size_t b = 0; b < per; ++b) r[i * per + b] = pool[i * per + (b & 7) * (per >> 3) + (b >> 3)];
			pool.swap(r);
		}
	}
	big.insert(big.end(), pool.begin(), pool.end());
	Tile *d; CK(hipMalloc(&d, sizeof(Tile) * big.size()));
	CK(hipMemcpy(d, big.data(), sizeof(Tile) * big.size(), hipMemcpyHostToDevice));
	return g_lists[key] = std::make_pair(d, (int)big.size());
}

extern "C" int skel_strip(const void *const *src, void *const *dst, int ni, int W, int H, int oc, int K, int nbig, int fbig_pm, int th, int order, void *stream)
{
	if (ni > MAXI) return -1;
	Imgs im;
	for (int i = 0; i < MAXI; ++i) { im.src[i] = (const uint8_t *)src[i < ni ? i : 0]; im.dst[i] = (uint8_t *)dst[i < ni ? i : 0]; }
	auto tl = tile_list(ni, W, H, oc, nbig, fbig_pm, th, order);
	const size_t lds = 16 + (size_t)K * 1024 + 64;
	hipLaunchKernelGGL((strip1<0>), dim3(tl.second), dim3(64), lds, (hipStream_t)stream, im, tl.first, W, H, oc, K);
	return tl.second;
}

extern "C" int skel_copy(const void *const *src, void *const *dst, int ni, int W, int H, int nt, void *stream)
{
	if (ni > MAXI) return -1;
	Imgs im;
	for (int i = 0; i < MAXI; ++i) { im.src[i] = (const uint8_t *)src[i < ni ? i : 0]; im.dst[i] = (uint8_t *)dst[i < ni ? i : 0]; }
	const size_t n16 = (size_t)W * H * 4 / 16; const int bpi = (int)((n16 + 1023) / 1024);
	if (nt) hipLaunchKernelGGL((copy_lin<true>), dim3(bpi * ni), dim3(256), 0, (hipStream_t)stream, im, n16, bpi);
	else hipLaunchKernelGGL((copy_lin<false>), dim3(bpi * ni), dim3(256), 0, (hipStream_t)stream, im, n16, bpi);
	return 0;
}
