// kbench3 — which side of the strip pattern costs: every wave moves `n` rows of 1 KB through an
// LDS-DMA ring, reading row r at rbase + r*rstride and writing it at wbase + r*wstride.  The host
// builds the tile list, so any traversal can be tried without touching the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/kbench3 tools/kbench3.hip
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
typedef const __attribute__((address_space(4))) unsigned long long *cull_p;
typedef const __attribute__((address_space(4))) int *cint_p;

struct Tile { unsigned long long rbase, wbase; int rstride, wstride, n, pad; };   /* 32 bytes */

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_younger(int y)
{
	if (y >= 8) wait_vm<8>(); else if (y == 7) wait_vm<7>(); else if (y == 6) wait_vm<6>(); else if (y == 5) wait_vm<5>();
	else if (y == 4) wait_vm<4>(); else if (y == 3) wait_vm<3>(); else if (y == 2) wait_vm<2>(); else if (y == 1) wait_vm<1>(); else wait_vm<0>();
}

template <int AUX, int D, int STAUX>
__global__ __launch_bounds__(64) void mover(const Tile *tiles, int K)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const unsigned long long rbase = tq[0], wbase = tq[1];
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)rbase, 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)wbase, 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	int ld = -1;
	auto issue = [&]() {
		++ld;
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + (ld % K) * 1024);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)row, 16, lane * 16, (uint32_t)ld * (uint32_t)rstride, 0, AUX);
	};
	for (int yb = 0; yb < n; yb += 4) {
		const int nb = min(4, n - yb);
		const int need = yb + nb - 1;
		{ const int target = min(min(need + D, yb + K - 1), n - 1); while (ld < target) issue(); }
		wait_younger(ld - need);
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (r < nb) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((yb + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (yb + 4 < n) { const int target = min(min(yb + 7 + D, yb + 4 + K - 1), n - 1); while (ld < target) issue(); }
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (r < nb) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(yb + r) * (uint32_t)wstride, 0, STAUX);
	}
}

// same loop as mover<AUX,6,2>, plus a per-tile log record {start, end (100 MHz ticks), HW_ID, XCC_ID}
template <int AUX>
__global__ __launch_bounds__(64) void mover_log(const Tile *tiles, int K, unsigned long long *log)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const unsigned long long rbase = tq[0], wbase = tq[1];
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)rbase, 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)wbase, 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	int ld = -1;
	auto issue = [&]() {
		++ld;
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + (ld % K) * 1024);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)row, 16, lane * 16, (uint32_t)ld * (uint32_t)rstride, 0, AUX);
	};
	for (int yb = 0; yb < n; yb += 4) {
		const int nb = min(4, n - yb);
		const int need = yb + nb - 1;
		{ const int target = min(min(need + 6, yb + K - 1), n - 1); while (ld < target) issue(); }
		wait_younger(ld - need);
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (r < nb) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((yb + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (yb + 4 < n) { const int target = min(min(yb + 7 + 6, yb + 4 + K - 1), n - 1); while (ld < target) issue(); }
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (r < nb) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(yb + r) * (uint32_t)wstride, 0, 2);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();     /* stores issued, not yet acknowledged */
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
	if (lane == 0) {
		unsigned long long *q = log + (size_t)blockIdx.x * 4;
		q[0] = t0; q[1] = t1; q[2] = t2;
		q[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
	}
}

// long tiles moved in bursts: B rows loaded at once, waited for together, stored together; the next
// burst's loads are issued before this burst's stores (ring of 2B rows)
template <int AUX, int B, int WAITN>
__global__ __launch_bounds__(64) void mover_burst(const Tile *tiles)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const unsigned long long rbase = tq[0], wbase = tq[1];
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)rbase, 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)wbase, 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	auto burst = [&](int y0) {
#pragma unroll
		for (int r = 0; r < B; ++r) {
			const int y = min(y0 + r, n - 1);
			__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + ((y0 / B & 1) * B + r) * 1024);
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)row, 16, lane * 16, (uint32_t)y * (uint32_t)rstride, 0, AUX);
		}
	};
	burst(0);
	for (int yb = 0; yb < n; yb += B) {
		if (yb + B < n) { burst(yb + B); if (yb > 0) wait_vm<WAITN>(); else wait_vm<B>(); }       /* the B youngest are the next burst; older stores must be done too */
		else wait_vm<0>();
		u32x4 o[B];
#pragma unroll
		for (int r = 0; r < B; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((yb / B & 1) * B + r) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < B; ++r)
			if (yb + r < n) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(yb + r) * (uint32_t)wstride, 0, 2);
	}
}

// plain-register version of the same traversal: 4 rows loaded, 4 rows stored, no LDS
template <bool NT>
__global__ __launch_bounds__(64) void mover_reg(const Tile *tiles)
{
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const uint8_t *rb = (const uint8_t *)tq[0] + lane * 16;
	uint8_t *wb = (uint8_t *)tq[1] + lane * 16;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	for (int yb = 0; yb < n; yb += 8) {
		u32x4 v[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) if (yb + r < n) v[r] = NT ? __builtin_nontemporal_load((const u32x4 *)(rb + (size_t)(yb + r) * rstride)) : *(const u32x4 *)(rb + (size_t)(yb + r) * rstride);
#pragma unroll
		for (int r = 0; r < 8; ++r) if (yb + r < n) __builtin_nontemporal_store(v[r], (u32x4 *)(wb + (size_t)(yb + r) * wstride));
	}
}

// ---- linear copy variants: what makes the plain nt copy reach 80 %? -------------------------------
// every workgroup copies one contiguous chunk of T/64 waves x U x 1 KB; chunk = perm(blockIdx)
struct Bufs { const u32x4 *src[16]; u32x4 *dst[16]; };
template <int T, int U, bool NT, bool LDSDMA>
__global__ __launch_bounds__(T) void copy_var(Bufs b, int chunks_per_img, unsigned mult, unsigned total)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	// a multiplicative permutation of the chunk index (mult = 1: identity order)
	const unsigned idx = (unsigned)(((unsigned long long)blockIdx.x * mult) % total);
	const int img = idx / chunks_per_img, c = idx - img * chunks_per_img;
	const size_t base = (size_t)c * (T * U) + threadIdx.x;         /* in 16-byte units */
	const u32x4 *s = b.src[img] + base;
	u32x4 *d = b.dst[img] + base;
	u32x4 v[U];
	if (LDSDMA) {
		const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
		uint8_t *my = ring + w * U * 1024;
#pragma unroll
		for (int u = 0; u < U; ++u)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s + u * T),
				(__attribute__((address_space(3))) void *)(my + u * 1024), 16, 0, NT ? 2 : 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
		for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4 *>(my + u * 1024 + lane * 16);
	} else {
#pragma unroll
		for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + u * T) : s[u * T];
	}
#pragma unroll
	for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], d + u * T);
}

__global__ void count_diff(const u32x4 *a, const u32x4 *b, size_t n, unsigned long long *out)
{
	unsigned long long bad = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		u32x4 x = a[i], y = b[i];
		bad += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
	}
	if (bad) atomicAdd(out, bad);
}

__global__ void fill_rand(uint32_t *p, size_t n, uint32_t seed)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 2654435761u + seed;
		x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
		p[i] = x;
	}
}

static const int NI = 16;
static std::vector<char *> srcs, dsts;

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

// mode bits: 1 = strip-shaped reads, 2 = strip-shaped writes (else linear: the tile's rows are contiguous)
static std::vector<Tile> make_tiles(int W, int H, int th, int mode, int order, size_t *bytes)
{
	const int pitch = W * 4, nstrips = W / 256;
	std::vector<Tile> t;
	size_t total = 0;
	for (int i = 0; i < NI; ++i) {
		size_t lin = 0;        /* running linear offset inside image i */
		auto add = [&](int s, int y0, int n) {
			Tile q;
			const size_t strip_off = (size_t)y0 * pitch + (size_t)s * 1024;
			q.rbase = (unsigned long long)(srcs[i] + ((mode & 1) ? strip_off : lin));
			q.wbase = (unsigned long long)(dsts[i] + ((mode & 2) ? strip_off : lin));
			q.rstride = (mode & 1) ? pitch : 1024;
			q.wstride = (mode & 2) ? pitch : 1024;
			q.n = n; q.pad = 0;
			t.push_back(q);
			lin += (size_t)n * 1024;
			total += (size_t)n * 1024;
		};
		if (order == 0) { for (int y = 0; y < H; y += th) for (int s = 0; s < nstrips; ++s) add(s, y, std::min(th, H - y)); }
		else            { for (int s = 0; s < nstrips; ++s) for (int y = 0; y < H; y += th) add(s, y, std::min(th, H - y)); }
	}
	*bytes = 2 * total;
	return t;
}

int main(int argc, char **argv)
{
	const int L = argc > 1 ? atoi(argv[1]) : 12;
	const size_t cap = (size_t)6144 * 4000 * 4 + 4096;
	srcs.resize(NI); dsts.resize(NI);
	for (int i = 0; i < NI; ++i) {
		CK(hipMalloc((void **)&srcs[i], cap)); CK(hipMalloc((void **)&dsts[i], cap));
		hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)srcs[i], cap / 4, 0x9E3779B9u * (i + 1));
		CK(hipMemset(dsts[i], 0, cap));
	}
	CK(hipDeviceSynchronize());
	Tile *d_tiles; CK(hipMalloc(&d_tiles, sizeof(Tile) * 400000));
	const int K = 16;
	const size_t lds = K * 1024;
	static const char *mname[] = { "lin->lin", "strip->lin", "lin->strip", "strip->strip" };
	struct { int W, th, mode, order; } cfg[] = {
		{6000, 33, 0, 0}, {6000, 33, 1, 0}, {6000, 33, 2, 0}, {6000, 33, 3, 0}, {6000, 33, 3, 1},
		{6000, 800, 0, 0}, {6000, 800, 3, 0}, {6000, 800, 3, 1}, {6000, 800, 1, 0}, {6000, 800, 2, 0},
		{6000, 8, 3, 0}, {6000, 8, 0, 0}, {6000, 128, 3, 0}, {6000, 128, 0, 0},
		{6144, 33, 3, 0}, {5888, 33, 3, 0}, {5632, 33, 3, 0}, {6080, 33, 3, 0}, {6016, 33, 3, 0},
		{2048, 33, 3, 0}, {1024, 33, 3, 0}, {512, 33, 3, 0}, {256, 33, 3, 0},
	};
	if (getenv("KB_COPYVAR")) {
		Bufs bf; for (int i = 0; i < NI; ++i) { bf.src[i] = (const u32x4 *)srcs[i]; bf.dst[i] = (u32x4 *)dsts[i]; }
		const size_t img16 = (size_t)96000000 / 16;
#define CV(T, U, NT, DMA, MULT) { const int cpi = (int)(img16 / (T * U)); const unsigned total = (unsigned)cpi * NI; \
			double us = time_it(L, [&] { hipLaunchKernelGGL((copy_var<T, U, NT, DMA>), dim3(total), dim3(T), DMA ? (T / 64) * U * 1024 : 0, 0, bf, cpi, (unsigned)(MULT), total); }); \
			printf("copy_var threads=%3d U=%2d nt_ld=%d lds_dma=%d perm_mult=%-8u : %5.1f%% of 8 TB/s\n", T, U, (int)NT, (int)DMA, (unsigned)(MULT), 2.0 * cpi * NI * T * U * 16 / us / 8e4); fflush(stdout); }
		CV(256, 4, false, false, 1) CV(256, 4, true, false, 1)
		CV(64, 4, false, false, 1)  CV(64, 4, true, false, 1)
		CV(64, 16, false, false, 1) CV(64, 16, true, false, 1)
		CV(256, 16, true, false, 1) CV(256, 1, true, false, 1) CV(512, 4, true, false, 1) CV(1024, 4, true, false, 1)
		CV(256, 4, true, false, 7919) CV(256, 4, true, false, 1000003) CV(64, 16, true, false, 1000003) CV(64, 16, false, false, 1000003)
		CV(256, 4, false, true, 1) CV(256, 4, true, true, 1) CV(64, 16, true, true, 1) CV(64, 16, false, true, 1) CV(64, 8, true, true, 1) CV(64, 32, true, true, 1)
		CV(64, 16, true, true, 1000003)
	}
	if (getenv("KB_LOG")) {
		// per-tile timing of the th=33 strip->strip traversal: who is slow?
		size_t bytes;
		for (int th : {33, 800}) {
			std::vector<Tile> t = make_tiles(6000, 4000, th, 3, 0, &bytes);
			CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
			unsigned long long *d_log; CK(hipMalloc(&d_log, t.size() * 32));
			for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mover_log<0>), dim3(t.size()), dim3(64), lds, 0, d_tiles, K, d_log);
			CK(hipDeviceSynchronize());
			std::vector<unsigned long long> h(t.size() * 4);
			CK(hipMemcpy(h.data(), d_log, t.size() * 32, hipMemcpyDeviceToHost));
			char fn[64]; snprintf(fn, 64, "gpurun_out/tilelog_th%d.bin", th);
			FILE *f = fopen(fn, "wb"); if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
			CK(hipFree(d_log));
		}
	}
	if (getenv("KB_BURST")) {
		size_t bytes;
		for (int th : {32, 128, 800}) for (int mode : {0, 3}) {
			std::vector<Tile> t = make_tiles(6000, 4000, th, mode, 0, &bytes);
			CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
			double a = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<0, 4, 4>), dim3(t.size()), dim3(64), 8 * 1024, 0, d_tiles); });
			double b = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<0, 8, 8>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles); });
			double c = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<2, 8, 8>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles); });
			double d = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<0, 16, 16>), dim3(t.size()), dim3(64), 32 * 1024, 0, d_tiles); });
			// exact counts: the previous burst's stores are younger than the loads waited for
			for (int i = 0; i < NI; ++i) CK(hipMemsetAsync(dsts[i], 0, cap, 0));
			double a2 = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<0, 4, 8>), dim3(t.size()), dim3(64), 8 * 1024, 0, d_tiles); });
			double b2 = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<0, 8, 16>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles); });
			double c2 = time_it(L, [&] { hipLaunchKernelGGL((mover_burst<2, 8, 16>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles); });
			unsigned long long *d_bad, h_bad = 0; CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
			// strip->strip and lin->lin both copy a region onto itself: compare it (W=6000: columns < 5888 of every row / the first 23*4000 KB)
			if (mode == 0) hipLaunchKernelGGL(count_diff, dim3(2048), dim3(256), 0, 0, (const u32x4 *)srcs[3], (const u32x4 *)dsts[3], (size_t)23 * 4000 * 64, d_bad);
			CK(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipFree(d_bad));
			printf("burst th=%3d %-12s tiles=%6zu | B=4: %5.1f%%  B=8: %5.1f%%  B=8 nt-ld: %5.1f%%  B=16: %5.1f%% | exact vmcnt: B=4: %5.1f%%  B=8: %5.1f%%  B=8 nt-ld: %5.1f%%  diff words %llu\n", th, mname[mode], t.size(),
			       bytes / a / 8e4, bytes / b / 8e4, bytes / c / 8e4, bytes / d / 8e4, bytes / a2 / 8e4, bytes / b2 / 8e4, bytes / c2 / 8e4, h_bad);
			fflush(stdout);
		}
	}
	for (auto &c : cfg) {
		if (getenv("KB_SKIP_MAIN")) break;
		const int H = (c.W >= 5000) ? 4000 : (int)((size_t)6000 * 4000 / c.W);
		size_t bytes;
		std::vector<Tile> t = make_tiles(c.W, H, c.th, c.mode, c.order, &bytes);
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		double us[5];
		us[0] = time_it(L, [&] { hipLaunchKernelGGL((mover<0, 6, 2>), dim3(t.size()), dim3(64), lds, 0, d_tiles, K); });
		us[1] = time_it(L, [&] { hipLaunchKernelGGL((mover<2, 6, 2>), dim3(t.size()), dim3(64), lds, 0, d_tiles, K); });
		us[2] = time_it(L, [&] { hipLaunchKernelGGL((mover<0, 6, 0>), dim3(t.size()), dim3(64), lds, 0, d_tiles, K); });
		us[3] = time_it(L, [&] { hipLaunchKernelGGL((mover_reg<false>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		us[4] = time_it(L, [&] { hipLaunchKernelGGL((mover_reg<true>), dim3(t.size()), dim3(64), 0, 0, d_tiles); });
		printf("W=%4d th=%3d %-12s order=%d tiles=%6zu | dma: %6.1f%%  dma-nt-ld: %6.1f%%  dma,plain-st: %6.1f%%  reg: %6.1f%%  reg-nt-ld: %6.1f%%   (%% of 8 TB/s; dma %.2f us per 192 MB)\n",
		       c.W, c.th, mname[c.mode], c.order, t.size(), bytes / us[0] / 8e4, bytes / us[1] / 8e4, bytes / us[2] / 8e4, bytes / us[3] / 8e4, bytes / us[4] / 8e4,
		       us[0] * 192e6 / bytes);
		fflush(stdout);
	}
	return 0;
}
