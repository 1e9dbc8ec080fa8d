// kbench4 — steady state of long-lived waves: 2048 equal tiles (8 per CU), every wave streams 500
// rows of 1 KB at the image pitch.  Compares (a) one wave doing LDS-DMA loads and stores with exact
// vmcnt accounting and P blocks of 4 rows in flight, (b) a loader wave and a storer wave per tile
// (loads and stores in different waves' queues), (c) the burst copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/kbench4 tools/kbench4.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef const __attribute__((address_space(4))) unsigned long long *cull_p;
typedef const __attribute__((address_space(4))) int *cint_p;
struct Tile { unsigned long long rbase, wbase; int rstride, wstride, n, pad; };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// (a) one wave: block b = rows 4b..4b+3; loads run P blocks ahead; ring of 4(P+1) rows.
// STRICT: wait with the loads-only count (round 1's rule: also waits for the stores in between).
template <int AUX, int P, bool STRICT, int STAUX>
__global__ __launch_bounds__(64) void pipe1(const Tile *tiles)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int K = 4 * (P + 1);
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	const int nblk = n / 4;                      /* n is a multiple of 4 here */
	auto load_block = [&](int b) {
		const int bb = min(b, nblk - 1);             /* past the end: reload the last block (keeps the counts regular) */
#pragma unroll
		for (int r = 0; r < 4; ++r)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, AUX);
	};
#pragma unroll
	for (int b = 0; b < P; ++b) load_block(b);
	for (int b = 0; b < nblk; ++b) {
		load_block(b + P);
		// younger than block b's loads: 4P loads, and the stores of the min(b, P) blocks stored since they were issued
		if (STRICT) wait_vm<4 * P>();
		else if (b >= P) wait_vm<8 * P>();
		else wait_vm<4 * P>();
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, STAUX);
	}
}

// (a') pipe1 with a per-wave start delay (hash of the block index, up to `maxdelay` us): breaks the
// lockstep in which equal static tiles march through memory
template <int P>
__global__ __launch_bounds__(64) void pipe1_skew(const Tile *tiles, int maxdelay_us)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int K = 4 * (P + 1);
	const int lane = threadIdx.x;
	{
		unsigned h = blockIdx.x * 2654435761u; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 13;
		const unsigned long long ticks = (unsigned long long)(h % (unsigned)(maxdelay_us * 100 + 1));
		const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
		while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	}
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
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
	};
#pragma unroll
	for (int b = 0; b < P; ++b) load_block(b);
	for (int b = 0; b < nblk; ++b) {
		load_block(b + P);
		if (b >= P) wait_vm<8 * P>(); else wait_vm<4 * P>();
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
	}
}

// (a'') NW waves per tile: each wave moves its own 1 KB column of a (NW KB wide) row band through its own
// ring, and one barrier per block keeps the NW waves on the same rows: NW KB contiguous per request burst
template <int P>
__global__ __launch_bounds__(512) void pipe_group(const Tile *tiles, int use_barrier)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring_all[];
	constexpr int K = 4 * (P + 1);
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint8_t *ring = ring_all + w * K * 1024;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0] + w * 1024, 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1] + w * 1024, 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	const int nblk = n / 4;
	auto load_block = [&](int b) {
		const int bb = min(b, nblk - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
	};
#pragma unroll
	for (int b = 0; b < P; ++b) load_block(b);
	for (int b = 0; b < nblk; ++b) {
		if (use_barrier) __builtin_amdgcn_s_barrier();
		load_block(b + P);
		if (b >= P) wait_vm<8 * P>(); else wait_vm<4 * P>();
		u32x4 o[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
		for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
	}
}

// (a3) persistent: a resident wave works through tiles w, w + G, w + 2G, ... of a short-tile list, one
// after the other (pipeline restarted per tile): the memory pattern of dynamic dispatch without launches
template <int P>
__global__ __launch_bounds__(64) void pipe1_multi(const Tile *tiles, int ntiles)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int K = 4 * (P + 1);
	const int lane = threadIdx.x;
	for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
		cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)t * 4;
		cint_p ti = (cint_p)(const int *)tiles + (size_t)t * 8;
		const int rstride = ti[4], wstride = ti[5], n = ti[6];
		rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
		rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
		const int nblk = n / 4;
		auto load_block = [&](int b) {
			const int bb = min(b, nblk - 1);
#pragma unroll
			for (int r = 0; r < 4; ++r)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, 0);
		};
#pragma unroll
		for (int b = 0; b < P; ++b) load_block(b);
		for (int b = 0; b < nblk; ++b) {
			load_block(b + P);
			wait_vm<4 * P>();
			u32x4 o[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
		}
	}
}

// (b) two waves per tile: wave 0 only loads, wave 1 only stores; one barrier per block
template <int AUX, int K>
__global__ __launch_bounds__(128) void pipe2(const Tile *tiles)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int P = K / 4 - 2;
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], wstride = ti[5], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	const int nblk = n / 4;
	if (w == 0) {
		auto load_block = [&](int b) {
			const int bb = min(b, nblk - 1);
#pragma unroll
			for (int r = 0; r < 4; ++r)
				__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, AUX);
		};
#pragma unroll
		for (int b = 0; b < P; ++b) load_block(b);
		for (int b = 0; b < nblk; ++b) {
			load_block(b + P);
			wait_vm<4 * P>();
			__builtin_amdgcn_s_barrier();
		}
	} else {
		for (int b = 0; b < nblk; ++b) {
			__builtin_amdgcn_s_barrier();
			u32x4 o[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) o[r] = *reinterpret_cast<const u32x4 *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b128(o[r], rd, lane * 16 + (uint32_t)(4 * b + r) * (uint32_t)wstride, 0, 2);
		}
	}
}

// (c) read-only and write-only steady state of the same traversal
template <int AUX, int P>
__global__ __launch_bounds__(64) void rd_only(const Tile *tiles, unsigned *sink)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int K = 4 * (P + 1);
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int rstride = ti[4], n = ti[6];
	rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[0], 0, (uint32_t)(n - 1) * (uint32_t)rstride + 1024u, 0x00020000);
	const int nblk = n / 4;
	auto load_block = [&](int b) {
		const int bb = min(b, nblk - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(ring + ((4 * b + r) % K) * 1024), 16, lane * 16, (uint32_t)(4 * bb + r) * (uint32_t)rstride, 0, AUX);
	};
	for (int b = 0; b < P; ++b) load_block(b);
	unsigned acc = 0;
	for (int b = 0; b < nblk; ++b) {
		load_block(b + P);
		wait_vm<4 * P>();
#pragma unroll
		for (int r = 0; r < 4; ++r) acc ^= *reinterpret_cast<const unsigned *>(ring + ((4 * b + r) % K) * 1024 + lane * 16);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	if (acc == 0x12345678u) sink[0] = acc;
}
template <int STAUX>
__global__ __launch_bounds__(64) void wr_only(const Tile *tiles)
{
	const int lane = threadIdx.x;
	cull_p tq = (cull_p)(const unsigned long long *)tiles + (size_t)blockIdx.x * 4;
	cint_p ti = (cint_p)(const int *)tiles + (size_t)blockIdx.x * 8;
	const int wstride = ti[5], n = ti[6];
	rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((uint8_t *)tq[1], 0, (uint32_t)(n - 1) * (uint32_t)wstride + 1024u, 0x00020000);
	u32x4 o = u32x4{ (unsigned)lane, 1u, 2u, 3u };
	for (int y = 0; y < n; ++y) {
		__builtin_amdgcn_raw_buffer_store_b128(o, rd, lane * 16 + (uint32_t)y * (uint32_t)wstride, 0, STAUX);
		o.y += 1;
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
__global__ void count_diff_strips(const uint8_t *a, const uint8_t *b, int pitch, int H, int nbytes_row, unsigned long long *out)
{
	unsigned long long bad = 0;
	const size_t per = nbytes_row / 16;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per * H; i += (size_t)gridDim.x * blockDim.x) {
		const size_t y = i / per, x = i - y * per;
		const u32x4 p = *(const u32x4 *)(a + y * pitch + x * 16), q = *(const u32x4 *)(b + y * pitch + x * 16);
		bad += (p.x != q.x) + (p.y != q.y) + (p.z != q.z) + (p.w != q.w);
	}
	if (bad) atomicAdd(out, bad);
}

static const int NI = 16;
template <typename F>
static double time_it(int launches, F f)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 2; ++i) f();
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < launches; ++i) f();
	CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	CK(hipGetLastError());
	return ms * 1e3 / launches;
}

int main(int argc, char **argv)
{
	const int L = argc > 1 ? atoi(argv[1]) : 10;
	const int W = 6000, H = 4000, pitch = W * 4;
	const size_t cap = (size_t)pitch * H;
	std::vector<char *> srcs(NI), dsts(NI);
	for (int i = 0; i < NI; ++i) {
		CK(hipMalloc((void **)&srcs[i], cap)); CK(hipMalloc((void **)&dsts[i], cap));
		hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, (uint32_t *)srcs[i], cap / 4, 0x9E3779B9u * (i + 1));
		CK(hipMemset(dsts[i], 0, cap));
	}
	CK(hipDeviceSynchronize());
	Tile *d_tiles; CK(hipMalloc(&d_tiles, sizeof(Tile) * 200000));
	unsigned *d_sink; CK(hipMalloc(&d_sink, 64));
	unsigned long long *d_bad; CK(hipMalloc(&d_bad, 8));

	// ---- the same mover on dynamically dispatched short tiles (same box, same binary) ----
	for (int th : {8, 12, 16, 24, 32, 64, 128}) {
		std::vector<Tile> t;
		const int nstrips = 16;
		for (int i = 0; i < NI; ++i) for (int y = 0; y + th <= H; y += th) for (int s = 0; s < nstrips; ++s) {
			Tile q; const size_t off = (size_t)y * pitch + (size_t)s * 1024;
			q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
			q.rstride = pitch; q.wstride = pitch; q.n = th; q.pad = 0; t.push_back(q);
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		const double bytes = 2.0 * t.size() * th * 1024;
		double a = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 3, false, 2>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles); });
		double b = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 1, false, 2>), dim3(t.size()), dim3(64), 8 * 1024, 0, d_tiles); });
		double c = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 3, false, 2>), dim3(t.size()), dim3(64), 20 * 1024, 0, d_tiles); });
		double d = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 3, false, 2>), dim3(t.size()), dim3(64), 32 * 1024, 0, d_tiles); });
		double d4 = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 3, false, 2>), dim3(t.size()), dim3(64), 40 * 1024, 0, d_tiles); });
		double d6 = time_it(L, [&] { hipLaunchKernelGGL((pipe1<0, 3, false, 2>), dim3(t.size()), dim3(64), 26 * 1024, 0, d_tiles); });
		double e = time_it(L, [&] { hipLaunchKernelGGL((pipe1_multi<3>), dim3(2048), dim3(64), 16 * 1024, 0, d_tiles, (int)t.size()); });
		double f = time_it(L, [&] { hipLaunchKernelGGL((pipe1_multi<3>), dim3(1280), dim3(64), 16 * 1024, 0, d_tiles, (int)t.size()); });
		double g = time_it(L, [&] { hipLaunchKernelGGL((pipe1_multi<1>), dim3(2048), dim3(64), 8 * 1024, 0, d_tiles, (int)t.size()); });
		printf("dynamic th=%3d tiles=%6zu : 9/CU %5.1f%%  P=1 %5.1f%%  8/CU %5.1f%%  6/CU %5.1f%%  5/CU %5.1f%%  4/CU %5.1f%% | persistent 2048 waves %5.1f%%  1280 waves %5.1f%%  2048 waves P=1 %5.1f%%\n", th, t.size(), bytes / a / 8e4, bytes / b / 8e4, bytes / c / 8e4, bytes / d6 / 8e4, bytes / d / 8e4, bytes / d4 / 8e4, bytes / e / 8e4, bytes / f / 8e4, bytes / g / 8e4);
		fflush(stdout);
	}
	// ---- groups: NW waves on NW adjacent strips (tile = NW KB wide), static equal tiles and dynamic short tiles ----
	for (int NW : {2, 4, 8}) for (int dyn : {0, 1}) for (int bar : {0, 1}) {
		std::vector<Tile> t;
		const int ngroups = 16 / NW;              /* 16 strips of 1 KB */
		int th;
		if (!dyn) {
			// 16 images x ngroups x nch = 256 * wgs_per_cu ; 8 waves per CU -> wgs_per_cu = 8 / NW
			const int nch = (256 * (8 / NW)) / (NI * ngroups);
			th = (H / nch) & ~3;
			for (int c = 0; c < nch; ++c) for (int i = 0; i < NI; ++i) for (int g = 0; g < ngroups; ++g) {
				Tile q; const size_t off = (size_t)c * th * pitch + (size_t)g * NW * 1024;
				q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
				q.rstride = pitch; q.wstride = pitch; q.n = th; q.pad = 0; t.push_back(q);
			}
		} else {
			th = 32;
			for (int i = 0; i < NI; ++i) for (int y = 0; y + th <= H; y += th) for (int g = 0; g < ngroups; ++g) {
				Tile q; const size_t off = (size_t)y * pitch + (size_t)g * NW * 1024;
				q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
				q.rstride = pitch; q.wstride = pitch; q.n = th; q.pad = 0; t.push_back(q);
			}
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		const double bytes = 2.0 * t.size() * th * 1024 * NW;
		CK(hipFuncSetAttribute((const void *)pipe_group<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		CK(hipFuncSetAttribute((const void *)pipe_group<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		double a = time_it(L, [&] { hipLaunchKernelGGL((pipe_group<3>), dim3(t.size()), dim3(64 * NW), NW * 16 * 1024, 0, d_tiles, bar); });
		double b = time_it(L, [&] { hipLaunchKernelGGL((pipe_group<1>), dim3(t.size()), dim3(64 * NW), NW * 8 * 1024, 0, d_tiles, bar); });
		printf("group NW=%d %s th=%4d wgs=%6zu barrier=%d : P=3 %5.1f%%   P=1 %5.1f%%\n", NW, dyn ? "dynamic" : "static ", th, t.size(), bar, bytes / a / 8e4, bytes / b / 8e4);
		fflush(stdout);
	}
	for (int per_cu : {8}) {
		// NI images x nstrips strips x nch chunks = 256 * per_cu tiles of equal height
		int nstrips = 16, nch = per_cu;           /* 16 * 16 * per_cu = 256 * per_cu */
		const int th = (H / nch) & ~3;
		std::vector<Tile> t;
		for (int c = 0; c < nch; ++c) for (int i = 0; i < NI; ++i) for (int s = 0; s < nstrips; ++s) {
			Tile q; const size_t off = (size_t)c * th * pitch + (size_t)s * 1024;
			q.rbase = (unsigned long long)(srcs[i] + off); q.wbase = (unsigned long long)(dsts[i] + off);
			q.rstride = pitch; q.wstride = pitch; q.n = th; q.pad = 0; t.push_back(q);
		}
		CK(hipMemcpy(d_tiles, t.data(), sizeof(Tile) * t.size(), hipMemcpyHostToDevice));
		const double bytes = 2.0 * t.size() * th * 1024;
		auto pct = [&](double us) { return bytes / us / 8e4; };
		auto check = [&]() -> unsigned long long {
			unsigned long long h = 0; CK(hipMemset(d_bad, 0, 8));
			hipLaunchKernelGGL(count_diff_strips, dim3(2048), dim3(256), 0, 0, (const uint8_t *)srcs[5], (const uint8_t *)dsts[5], pitch, th * nch, nstrips * 1024, d_bad);
			CK(hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost));
			for (int i = 0; i < NI; ++i) CK(hipMemsetAsync(dsts[i], 0, cap, 0));
			return h;
		};
		printf("== %d tiles per CU (%zu tiles of %d rows)\n", per_cu, t.size(), th);
#define RUN1(AUX, P, STRICT, ST) if (4 * (P + 1) * per_cu <= 160) { double us = time_it(L, [&] { hipLaunchKernelGGL((pipe1<AUX, P, STRICT, ST>), dim3(t.size()), dim3(64), 4 * (P + 1) * 1024, 0, d_tiles); }); \
			printf("  pipe1 ld_aux=%d P=%d (%2d KB ring) %s st_aux=%d : %5.1f%%  diff %llu\n", AUX, P, 4 * (P + 1), STRICT ? "loads-only count" : "exact count     ", ST, pct(us), check()); fflush(stdout); }
		RUN1(0, 3, true, 2) RUN1(0, 3, false, 2) RUN1(2, 3, false, 2) RUN1(0, 2, false, 2) RUN1(0, 1, false, 2) RUN1(0, 5, false, 2) RUN1(2, 5, false, 2) RUN1(0, 7, false, 2) RUN1(0, 3, false, 0) RUN1(2, 3, false, 0)
#define RUN2(AUX, K) if (K * per_cu <= 160) { double us = time_it(L, [&] { hipLaunchKernelGGL((pipe2<AUX, K>), dim3(t.size()), dim3(128), K * 1024, 0, d_tiles); }); \
			printf("  pipe2 loader+storer ld_aux=%d K=%d : %5.1f%%  diff %llu\n", AUX, K, pct(us), check()); fflush(stdout); }
		RUN2(0, 16) RUN2(2, 16) RUN2(0, 24) RUN2(2, 24)
		for (int d : {0, 2, 10, 40, 150}) {
			double us = time_it(L, [&] { hipLaunchKernelGGL((pipe1_skew<3>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles, d); });
			printf("  pipe1 P=3 static, random start delay up to %3d us : %5.1f%% (of which the delay itself costs up to %.1f%%)\n", d, pct(us), 100.0 * d / us);
		}
		if (per_cu == 8) {
			{ double us = time_it(L, [&] { hipLaunchKernelGGL((rd_only<0, 3>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles, d_sink); }); printf("  read-only  default : %5.1f%% (of 8 TB/s, bytes read only)\n", pct(us) / 2); }
			{ double us = time_it(L, [&] { hipLaunchKernelGGL((rd_only<2, 3>), dim3(t.size()), dim3(64), 16 * 1024, 0, d_tiles, d_sink); }); printf("  read-only  nt      : %5.1f%%\n", pct(us) / 2); }
			{ double us = time_it(L, [&] { hipLaunchKernelGGL((wr_only<2>), dim3(t.size()), dim3(64), 0, 0, d_tiles); }); printf("  write-only nt      : %5.1f%%\n", pct(us) / 2); }
			{ double us = time_it(L, [&] { hipLaunchKernelGGL((wr_only<0>), dim3(t.size()), dim3(64), 0, 0, d_tiles); }); printf("  write-only default : %5.1f%%\n", pct(us) / 2); }
		}
	}
	return 0;
}
