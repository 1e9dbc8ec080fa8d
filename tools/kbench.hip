// kbench — on-box micro-benchmark: copy-kernel variants (the bandwidth ceiling) and the
// Fix-CA kernels through the C-ABI with FIXCA_TH / FIXCA_OC sweeps.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/kbench tools/kbench.hip -Lgimp-fix-ca_amd -lfixca_hip -Wl,-rpath,$PWD/gimp-fix-ca_amd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../include/fixca_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_u(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n)
{
	size_t base = ((size_t)blockIdx.x * 256 * UNROLL) + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
	for (; base < n; base += stride) {
		u32x4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			size_t i = base + (size_t)u * 256;
			if (i < n) v[u] = NT ? __builtin_nontemporal_load(&s[i]) : s[i];
		}
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			size_t i = base + (size_t)u * 256;
			if (i < n) { if (NT) __builtin_nontemporal_store(v[u], &d[i]); else d[i] = v[u]; }
		}
	}
}


// ---- skeleton of the LDS-DMA row ring: global -> LDS (async, D rows ahead) -> registers -> global -------
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

#ifndef SKEL_LD_AUX
#define SKEL_LD_AUX 0
#endif
template <int D, bool NTS>
__global__ __launch_bounds__(256) void strip_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int th, int nstrips, int nwaves, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x & 63;
	const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int wid = blockIdx.x * 4 + wib;
	if (wid >= nwaves) return;
	const int strip = wid % nstrips, chunk = wid / nstrips;
	int X0 = strip * 256; if (X0 > W - 256) X0 = W - 256;
	const int Y0 = chunk * th, Y1 = min(Y0 + th, H);
	const size_t pitch = (size_t)W * 4;
	uint8_t *myring = ring + (size_t)wib * K * 1024;
	const uint32_t loff = (uint32_t)(X0 + 4 * lane) * 4;
	auto issue = [&](int row) {
		const uint8_t *g = src + (size_t)row * pitch + loff;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
			(__attribute__((address_space(3))) void *)(myring + (row % K) * 1024), 16, 0, SKEL_LD_AUX);
	};
#pragma unroll
	for (int d = 0; d < D; ++d) if (Y0 + d < Y1) issue(Y0 + d);
	for (int y = Y0; y < Y1; ++y) {
		if (y + D < Y1) issue(y + D);
		wait_vm<2 * D>();
		u32x4 v = *reinterpret_cast<const u32x4 *>(myring + (y % K) * 1024 + lane * 16);
		u32x4 *o = reinterpret_cast<u32x4 *>(dst + (size_t)y * pitch + loff);
		if (NTS) __builtin_nontemporal_store(v, o); else *o = v;
	}
}

template <int D, bool NTS>
__global__ __launch_bounds__(256) void strip_copy_sync(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int th, int nstrips, int nwaves, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x & 63;
	const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int wid = blockIdx.x * 4 + wib;
	const bool dead = wid >= nwaves;
	const int widc = dead ? nwaves - 1 : wid;
	const int strip = widc % nstrips, chunk = widc / nstrips;
	int X0 = strip * 256; if (X0 > W - 256) X0 = W - 256;
	const int Y0 = chunk * th, Y1 = min(Y0 + th, H);
	const size_t pitch = (size_t)W * 4;
	uint8_t *myring = ring + (size_t)wib * K * 1024;
	const uint32_t loff = (uint32_t)(X0 + 4 * lane) * 4;
	auto issue = [&](int row) {
		const uint8_t *g = src + (size_t)row * pitch + loff;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
			(__attribute__((address_space(3))) void *)(myring + (row % K) * 1024), 16, 0, 0);
	};
#pragma unroll
	for (int d = 0; d < D; ++d) if (Y0 + d < Y1) issue(Y0 + d);
	for (int y = Y0; y < Y1; ++y) {
		if (y + D < Y1) issue(y + D);
		wait_vm<2 * D>();
		__builtin_amdgcn_s_barrier();            /* the 4 waves of the workgroup (4 adjacent strips) stay on the same row */
		u32x4 v = *reinterpret_cast<const u32x4 *>(myring + (y % K) * 1024 + lane * 16);
		u32x4 *o = reinterpret_cast<u32x4 *>(dst + (size_t)y * pitch + loff);
		if (NTS) __builtin_nontemporal_store(v, o); else *o = v;
	}
}

// ---- LDS-ring skeleton, one strip per 4-wave workgroup: wave w copies the rows y = w (mod 4) of the chunk,
// so a quarter as many row streams are in flight, each four times as fast ----------------------------------
template <int D>
__global__ __launch_bounds__(256) void strip_copy_wg(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int th, int nstrips, int ntiles, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x & 63;
	const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int tile = blockIdx.x;
	if (tile >= ntiles) return;
	const int strip = tile % nstrips, chunk = tile / nstrips;
	int X0 = strip * 256; if (X0 > W - 256) X0 = W - 256;
	const int Y0 = chunk * th, Y1 = min(Y0 + th, H);
	const size_t pitch = (size_t)W * 4;
	uint8_t *myring = ring + (size_t)wib * K * 1024;
	const uint32_t loff = (uint32_t)(X0 + 4 * lane) * 4;
	auto issue = [&](int row) {
		const uint8_t *g = src + (size_t)row * pitch + loff;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
			(__attribute__((address_space(3))) void *)(myring + ((row >> 2) % K) * 1024), 16, 0, 0);
	};
#pragma unroll
	for (int d = 0; d < D; ++d) if (Y0 + wib + 4 * d < Y1) issue(Y0 + wib + 4 * d);
	for (int y = Y0 + wib; y < Y1; y += 4) {
		if (y + 4 * D < Y1) issue(y + 4 * D);
		wait_vm<2 * D>();
		u32x4 v = *reinterpret_cast<const u32x4 *>(myring + ((y >> 2) % K) * 1024 + lane * 16);
		u32x4 *o = reinterpret_cast<u32x4 *>(dst + (size_t)y * pitch + loff);
		__builtin_nontemporal_store(v, o);
	}
}

// ---- LDS-ring skeleton with 512-column strips: two 1 KB DMAs and two 16-byte stores per lane and row ----
template <int D>
__global__ __launch_bounds__(256) void strip_copy_wide(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int th, int nstrips, int nwaves, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x & 63;
	const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int wid = blockIdx.x * 4 + wib;
	if (wid >= nwaves) return;
	const int strip = wid % nstrips, chunk = wid / nstrips;
	int X0 = strip * 512; if (X0 > W - 512) X0 = W - 512;
	const int Y0 = chunk * th, Y1 = min(Y0 + th, H);
	const size_t pitch = (size_t)W * 4;
	uint8_t *myring = ring + (size_t)wib * K * 2048;
	const uint32_t loff = (uint32_t)(X0 + 4 * lane) * 4;
	auto issue = [&](int row) {
		const uint8_t *g = src + (size_t)row * pitch + loff;
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
			(__attribute__((address_space(3))) void *)(myring + (row % K) * 2048), 16, 0, 0);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 1024),
			(__attribute__((address_space(3))) void *)(myring + (row % K) * 2048 + 1024), 16, 0, 0);
	};
#pragma unroll
	for (int d = 0; d < D; ++d) if (Y0 + d < Y1) issue(Y0 + d);
	for (int y = Y0; y < Y1; ++y) {
		if (y + D < Y1) issue(y + D);
		wait_vm<4 * D>();
		u32x4 v0 = *reinterpret_cast<const u32x4 *>(myring + (y % K) * 2048 + lane * 16);
		u32x4 v1 = *reinterpret_cast<const u32x4 *>(myring + (y % K) * 2048 + 1024 + lane * 16);
		u32x4 *o = reinterpret_cast<u32x4 *>(dst + (size_t)y * pitch + loff);
		__builtin_nontemporal_store(v0, o);
		__builtin_nontemporal_store(v1, o + 64);
	}
}

// ---- the same tile pattern without LDS: global_load -> registers (D rows in flight) -> global_store -------
template <int D, bool NT>
__global__ __launch_bounds__(256) void strip_direct(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H, int th, int nstrips, int nwaves)
{
	const int lane = threadIdx.x & 63;
	const int wib = threadIdx.x >> 6;
	const int wid = blockIdx.x * 4 + wib;
	if (wid >= nwaves) return;
	const int strip = wid % nstrips, chunk = wid / nstrips;
	int X0 = strip * 256; if (X0 > W - 256) X0 = W - 256;
	const int Y0 = chunk * th, Y1 = min(Y0 + th, H);
	const size_t pitch = (size_t)W * 4;
	const uint8_t *g = src + (size_t)Y0 * pitch + (size_t)(X0 + 4 * lane) * 4;
	uint8_t *o = dst + (size_t)Y0 * pitch + (size_t)(X0 + 4 * lane) * 4;
	u32x4 buf[D];
#pragma unroll
	for (int d = 0; d < D; ++d) if (Y0 + d < Y1) buf[d] = NT ? __builtin_nontemporal_load((const u32x4 *)(g + d * pitch)) : *(const u32x4 *)(g + d * pitch);
	for (int y = Y0; y < Y1; y += D) {
#pragma unroll
		for (int d = 0; d < D; ++d) {
			if (y + d < Y1) {
				u32x4 v = buf[d];
				if (y + d + D < Y1) buf[d] = NT ? __builtin_nontemporal_load((const u32x4 *)(g + (size_t)(y - Y0 + d + D) * pitch)) : *(const u32x4 *)(g + (size_t)(y - Y0 + d + D) * pitch);
				__builtin_nontemporal_store(v, (u32x4 *)(o + (size_t)(y - Y0 + d) * pitch));
			}
		}
	}
}

template <typename F>
static double time_it(int launches, F f)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 8; ++i) f(i);
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < launches; ++i) f(i);
	CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms * 1e3 / launches;
}

int main(int argc, char **argv)
{
	const int W = argc > 1 ? atoi(argv[1]) : 6000, H = argc > 2 ? atoi(argv[2]) : 4000;
	const int NP = 6, L = 120;
	std::vector<void *> src4(NP), dst4(NP);
	const size_t n4 = (size_t)W * H * 4;
	for (int i = 0; i < NP; ++i) {
		CK(hipMalloc(&src4[i], n4)); CK(hipMalloc(&dst4[i], n4));
		std::vector<unsigned char> h(n4);
		unsigned long long z = 0x9E3779B97F4A7C15ull * (i + 1);
		for (size_t k = 0; k < n4; k += 8) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; memcpy(&h[k], &z, 8); }
		CK(hipMemcpy(src4[i], h.data(), n4, hipMemcpyHostToDevice));
	}
	const double bytes = 2.0 * n4;
	auto report = [&](const char *name, double us) { printf("%-44s %8.2f us  %7.1f GB/s  %9.0f Mpix/s\n", name, us, bytes / us / 1e3, (double)W * H / us); fflush(stdout); };

	if (!getenv("KB_SKIP_COPY")) {
		size_t n16 = n4 / 16;
		for (int grid : {1024, 2048, 4096, 8192, 16384}) {
			char nm[64];
			snprintf(nm, 64, "copy u1 grid=%d", grid);  report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((copy_u<1, false>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)src4[i % NP], (u32x4 *)dst4[i % NP], n16); }));
			snprintf(nm, 64, "copy u2 grid=%d", grid);  report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((copy_u<2, false>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)src4[i % NP], (u32x4 *)dst4[i % NP], n16); }));
			snprintf(nm, 64, "copy u4 grid=%d", grid);  report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((copy_u<4, false>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)src4[i % NP], (u32x4 *)dst4[i % NP], n16); }));
			snprintf(nm, 64, "copy u4 nt grid=%d", grid); report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((copy_u<4, true>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)src4[i % NP], (u32x4 *)dst4[i % NP], n16); }));
		}
		report("hipMemcpyDtoD", time_it(L, [&](int i) { CK(hipMemcpyAsync(dst4[i % NP], src4[i % NP], n4, hipMemcpyDeviceToDevice, 0)); }));
	}


	if (!getenv("KB_SKIP_SKEL")) {
		for (int th : {16, 32, 64}) for (int K : {8, 16, 32}) {
			int nstrips = (W + 255) / 256, nchunks = (H + th - 1) / th, nwaves = nstrips * nchunks, nb = (nwaves + 3) / 4;
			size_t lds = (size_t)4 * K * 1024;
			char nm[96];
#define SK(D, NTS) if (D < K) { hipFuncSetAttribute((const void *)strip_copy<D, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
			snprintf(nm, 96, "strip_copy th=%d K=%d D=%d nt=%d waves=%d", th, K, D, (int)NTS, nwaves); \
			report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((strip_copy<D, NTS>), dim3(nb), dim3(256), lds, 0, (const uint8_t *)src4[i % NP], (uint8_t *)dst4[i % NP], W, H, th, nstrips, nwaves, K); })); }
			SK(2, false) SK(4, false) SK(7, false) SK(4, true) SK(7, true) SK(14, true) SK(24, true)
		}
	}

	if (!getenv("KB_SKIP_SKEL")) {
		for (int th : {16, 32, 64}) {
			int nstrips = (W + 255) / 256, nchunks = (H + th - 1) / th, nwaves = nstrips * nchunks, nb = (nwaves + 3) / 4;
			char nm[96];
#define SD(D, NT) { snprintf(nm, 96, "strip_direct th=%d D=%d ntload=%d waves=%d", th, D, (int)NT, nwaves); \
			report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((strip_direct<D, NT>), dim3(nb), dim3(256), 0, 0, (const uint8_t *)src4[i % NP], (uint8_t *)dst4[i % NP], W, H, th, nstrips, nwaves); })); }
			SD(4, false) SD(8, false) SD(8, true)
		}
	}

	if (!getenv("KB_SKIP_SKEL")) {
		for (int th : {64, 128, 256}) {
			const int K = 16;
			int nstrips = (W + 255) / 256, nchunks = (H + th - 1) / th, ntiles = nstrips * nchunks;
			size_t lds = (size_t)4 * K * 1024;
			char nm[96];
			hipFuncSetAttribute((const void *)strip_copy_wg<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			snprintf(nm, 96, "strip_copy_wg th=%d K=16 D=7 workgroups=%d", th, ntiles);
			report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((strip_copy_wg<7>), dim3(ntiles), dim3(256), lds, 0, (const uint8_t *)src4[i % NP], (uint8_t *)dst4[i % NP], W, H, th, nstrips, ntiles, K); }));
		}
	}
	if (!getenv("KB_SKIP_SKEL")) {
		for (int th : {16, 32}) {
			const int K = 16;
			int nstrips = (W + 255) / 256, nchunks = (H + th - 1) / th;
			// strips padded to a multiple of 4 so that a workgroup's 4 waves share a chunk
			int nstrips4 = (nstrips + 3) & ~3, nwaves = nstrips4 * nchunks, nb = nwaves / 4;
			size_t lds = (size_t)4 * K * 1024;
			char nm[96];
			hipFuncSetAttribute((const void *)strip_copy_sync<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			snprintf(nm, 96, "strip_copy_sync th=%d K=16 D=7 nt=1 waves=%d", th, nwaves);
			report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((strip_copy_sync<7, true>), dim3(nb), dim3(256), lds, 0, (const uint8_t *)src4[i % NP], (uint8_t *)dst4[i % NP], W, H, th, nstrips4, nwaves, K); }));
		}
	}
	if (!getenv("KB_SKIP_SKEL")) {
		for (int th : {16, 32, 64}) for (int K : {8}) {
			int nstrips = (W + 511) / 512, nchunks = (H + th - 1) / th, nwaves = nstrips * nchunks, nb = (nwaves + 3) / 4;
			size_t lds = (size_t)4 * K * 2048;
			char nm[96];
			hipFuncSetAttribute((const void *)strip_copy_wide<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			snprintf(nm, 96, "strip_copy_wide(512) th=%d K=%d D=4 waves=%d", th, K, nwaves);
			report(nm, time_it(L, [&](int i) { hipLaunchKernelGGL((strip_copy_wide<4>), dim3(nb), dim3(256), lds, 0, (const uint8_t *)src4[i % NP], (uint8_t *)dst4[i % NP], W, H, th, nstrips, nwaves, K); }));
		}
	}

	FixCaParams p; memset(&p, 0, sizeof p);
	p.blue = 2.0; p.red = -2.0; p.lens_x = 658; p.lens_y = 1280;
	const char *abl = getenv("KB_ABLATE") ? getenv("KB_ABLATE") : "0";
	const char *ths = getenv("KB_TH") ? getenv("KB_TH") : "8 16 32 64";
	const char *ocs = getenv("KB_OC") ? getenv("KB_OC") : "252";
	for (int interp : {2, 1, 0}) {
		p.interpolation = interp;
		char *al = strdup(abl), *asave;
		for (char *ab = strtok_r(al, " ", &asave); ab; ab = strtok_r(NULL, " ", &asave)) {
		setenv("FIXCA_ABLATE", ab, 1);
		char *tl = strdup(ths), *tsave;
		for (char *t = strtok_r(tl, " ", &tsave); t; t = strtok_r(NULL, " ", &tsave)) {
			char *ol = strdup(ocs), *save;
			for (char *o = strtok_r(ol, " ", &save); o; o = strtok_r(NULL, " ", &save)) {
				setenv("FIXCA_TH", t, 1); setenv("FIXCA_OC", o, 1);
				char desc[256];
				fixca_plan_describe(W, H, 4, 1, &p, 0, W, 0, H, 1, FIXCA_KERNEL_FAST, desc, sizeof desc);
				double us = time_it(L, [&](int i) {
					if (fixca_region_device(src4[i % NP], dst4[i % NP], W, H, 4, 1, &p, 0, W, 0, H, 1, FIXCA_KERNEL_FAST, NULL)) { printf("err %s\n", fixca_last_error()); exit(1); }
				});
				char d2[300]; snprintf(d2, 300, "[abl %s] %s", ab, desc);
				report(d2, us);
				if (interp == 0) break;
			}
			free(ol);
		}
		free(tl);
		}
		free(al);
	}
	return 0;
}
