/*
 * fixca_abi.hip — the extern "C" entry points declared in include/fixca_hip.h.
 * Thin: validate, plan, launch.  Every failure is reported (code + text); there
 * is no CPU fallback.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "../../include/fixca_hip.h"
#include "fixca_geom.h"
#include "fixca_internal.h"
#include "fixca_roll.h"

extern "C" hipError_t fixca_launch(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_copy(const void *, void *, size_t, hipStream_t);
extern "C" hipError_t fixca_launch_noise(void *, size_t, unsigned long long, hipStream_t);
extern "C" hipError_t fixca_launch_preview(const void *, void *, int, int, int, int, int, int, int, hipStream_t);
extern "C" hipError_t fixca_launch_centerline(void *, const FixcaGeom *, hipStream_t);
extern "C" hipError_t fixca_launch_roll_batch(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_roll_batch_rgb(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_move_batch(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_lin_batch(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_wide_batch(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_cub_batch(const void *const *, void *const *, int, const FixcaGeom *, const FixcaPlan *, hipStream_t);
#include "fixca_move.h"

/* a batch of images of one geometry: the row-rolling kernel takes up to ROLL_BATCH_MAX (128) per launch, the
 * kernels of the wide sample types (k_wcol, k_move, k_lin, k_cub) up to MOVE_BATCH_MAX (128), the other kernels one */
static int launch_many(const void *const *d_src, void *const *d_dst, int n, const FixcaGeom *g, const FixcaPlan *pl,
                       hipStream_t stream, int *launches);

static thread_local char g_err[512] = "";

int fixca_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

extern "C" const char *fixca_last_error(void) { return g_err; }

/* FIXCA_PHASES=1: time stamps of a call's phases on stderr ("fixca-phase <name> <ms>"), in milliseconds of
 * CLOCK_MONOTONIC since FIXCA_T0_NS (the spawning process puts its own clock reading there: the clock is system
 * wide) or since the first stamp.  bench.py's cold_call leg reads them; off by default, one getenv per process. */
#include <time.h>
extern "C" void fixca_phase(const char *name)
{
	/* the prewarm thread and the calling thread may both make the first call: the settings are read once, behind the
	 * guard of a function-local static, and the clock origin (when the parent gave none) is taken there too */
	struct Settings {
		bool      on;
		long long t0;
		Settings()
		{
			const char *e = getenv("FIXCA_PHASES");
			on = e && *e && strcmp(e, "0") != 0;
			const char *z = getenv("FIXCA_T0_NS");
			t0 = (z && *z) ? atoll(z) : 0;
			if (on && !t0) {
				struct timespec ts;
				clock_gettime(CLOCK_MONOTONIC, &ts);
				t0 = (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
			}
		}
	};
	static const Settings cfg;
	if (!cfg.on) return;
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	const long long now = (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
	fprintf(stderr, "fixca-phase %s %.3f\n", name, (double)(now - cfg.t0) / 1e6);
}
extern "C" int fixca_abi_version(void) { return 1; }

#define HIP_TRY(call)                                                                       \
	do {                                                                                    \
		hipError_t e_ = (call);                                                             \
		if (e_ != hipSuccess)                                                               \
			return fixca_fail(FIXCA_E_HIP, "%s: %s", #call, hipGetErrorString(e_));         \
	} while (0)

/* ---- device plumbing ---------------------------------------------------- */

int fixca_dyn_lds(const void *fn, int bytes)
{
	constexpr int MAXDEV = 16, MAXFN = 96;
	static std::mutex mu;
	static const void *done[MAXDEV][MAXFN];
	static int ndone[MAXDEV];
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return (int)e;
	const bool tracked = dev >= 0 && dev < MAXDEV;
	std::lock_guard<std::mutex> lock(mu);
	if (tracked)
		for (int i = 0; i < ndone[dev]; ++i)
			if (done[dev][i] == fn) return (int)hipSuccess;
	e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);      /* untracked devices: every launch, it is cheap */
	if (e != hipSuccess) return (int)e;
	if (tracked && ndone[dev] < MAXFN) done[dev][ndone[dev]++] = fn;
	return (int)hipSuccess;
}

extern "C" int fixca_device_count(void)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e == hipErrorNoDevice) return 0;
	if (e != hipSuccess) return fixca_fail(FIXCA_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
	return n;
}

extern "C" int fixca_set_device(int ordinal)
{
	HIP_TRY(hipSetDevice(ordinal));
	return FIXCA_OK;
}

extern "C" void *fixca_dev_alloc(size_t nbytes)
{
	void *p = NULL;
	hipError_t e = hipMalloc(&p, nbytes ? nbytes : 1);
	if (e != hipSuccess) {
		fixca_fail(FIXCA_E_HIP, "hipMalloc(%zu): %s", nbytes, hipGetErrorString(e));
		return NULL;
	}
	return p;
}

extern "C" int fixca_dev_free(void *p)
{
	HIP_TRY(hipFree(p));
	return FIXCA_OK;
}

extern "C" void *fixca_host_alloc(size_t nbytes)
{
	void *p = NULL;
	hipError_t e = hipHostMalloc(&p, nbytes ? nbytes : 1, hipHostMallocDefault);
	if (e != hipSuccess) {
		fixca_fail(FIXCA_E_HIP, "hipHostMalloc(%zu): %s", nbytes, hipGetErrorString(e));
		return NULL;
	}
	return p;
}

extern "C" int fixca_host_free(void *p)
{
	HIP_TRY(hipHostFree(p));
	return FIXCA_OK;
}

extern "C" int fixca_copy_h2d(void *d, const void *h, size_t n, void *stream)
{
	HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)stream));
	return FIXCA_OK;
}

extern "C" int fixca_copy_d2h(void *h, const void *d, size_t n, void *stream)
{
	HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return FIXCA_OK;
}

extern "C" int fixca_stream_sync(void *stream)
{
	HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	return FIXCA_OK;
}

/* ---- does this device move dwords at any byte address? --------------------------------------------------------------
 * Images whose rows are not whole dwords (RGB8 of a width that is not a multiple of 4, RGB u16 of odd width) run on the
 * main kernels since round 4 because gfx950's buffer loads, LDS-DMA and 12 / 16-byte stores take byte addresses as they
 * come (tools/skeletons/unaligned_probe.hip).  That is a property of the device's memory configuration, not of the ISA,
 * so the planner asks: the first such geometry on a device runs this one-wave probe (a 4-KiB scratch, one
 * synchronisation; the answer is cached per device) and falls back to the register-window kernels if the bytes that
 * arrive are not the shifted ones.  No device (planning only): assumed, as fixca_plan_describe documents. */
namespace {
typedef __amdgpu_buffer_rsrc_t probe_rsrc_t;
typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int probe_u32x3 __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(64) void k_unaligned_probe(unsigned char *buf, int *verdict)
{
	/* buf: 4096 bytes; [0, 1024): a byte pattern written by the host; [2048, 3072): the store target, zeroed */
	__shared__ __attribute__((aligned(16))) unsigned char lds[1024 + 16];
	const int lane = threadIdx.x;
	probe_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
	int bad = 0;
#pragma unroll
	for (int shift = 1; shift < 4; ++shift) {
#if defined(__HIP_DEVICE_COMPILE__)
		if (lane < 48)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, (unsigned)(shift + 16 * lane), 0, 0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
		__syncthreads();
		const probe_u32x4 a = *reinterpret_cast<const probe_u32x4 *>(lds + 16 * lane);                 /* LDS-DMA from a byte address */
		const probe_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(shift + 16 * lane), 0, 0);   /* plain load from one */
		unsigned char pa[16], pb[16];
		__builtin_memcpy(pa, &a, 16);
		__builtin_memcpy(pb, &b, 16);
		if (lane < 48)
			for (int i = 0; i < 16; ++i) bad |= (pa[i] != buf[shift + 16 * lane + i]) | (pb[i] != buf[shift + 16 * lane + i]);
		/* 12-byte store to a byte address, read back bytewise */
		__builtin_amdgcn_raw_buffer_store_b96(probe_u32x3{ b.x, b.y, b.z }, rs, (unsigned)(2048 + shift + 12 * lane), 0, 0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (lane < 48)
			for (int i = 0; i < 12; ++i) bad |= buf[2048 + shift + 12 * lane + i] != pb[i];
		__syncthreads();
	}
	bad = __any(bad);
	if (lane == 0) *verdict = bad ? 2 : 1;
}
}

extern "C" int fixca_unaligned_ok(void)
{
	constexpr int MAXDEV = 16;
	static std::mutex mu;
	static int verdict[MAXDEV];                          /* 0 not asked, 1 faithful, 2 not */
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		return 1;                                        /* no device: the planner describes what a gfx950 would run */
	}
	if (dev < 0 || dev >= MAXDEV) return 0;
	std::lock_guard<std::mutex> lock(mu);
	if (verdict[dev] == 0) {
		unsigned char host[4096];
		for (int i = 0; i < 4096; ++i) host[i] = i < 2048 ? (unsigned char)(i * 7 + 3) : 0;
		unsigned char *d = nullptr;
		int *dv = nullptr, v = 2;
		if (hipMalloc(&d, 4096 + sizeof(int)) == hipSuccess) {
			dv = reinterpret_cast<int *>(d + 4096);
			if (hipMemcpy(d, host, 4096, hipMemcpyHostToDevice) == hipSuccess && hipMemset(dv, 0, sizeof(int)) == hipSuccess) {
				hipLaunchKernelGGL(k_unaligned_probe, dim3(1), dim3(64), 0, 0, d, dv);
				if (hipMemcpy(&v, dv, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) v = 2;
			}
			(void)hipFree(d);
		}
		(void)hipGetLastError();
		verdict[dev] = (v == 1) ? 1 : 2;
	}
	return verdict[dev] == 1;
}

/* ---- start-up overlap ---------------------------------------------------------------------------------------------
 * A PDB call is a fresh process, and the first HIP call in it costs 56-227 ms (runtime and device initialisation,
 * profiles/NOTES.md, "cold-start").  fixca_prewarm() starts that work on a background thread and returns at once, so a
 * host can call it first thing in run() and spend the time in what it has to do anyway before the pixels are ready
 * (gegl_buffer_get of the drawable; fixca_cli: reading the file).  The next entry that needs the device joins the
 * thread.  Calling it is optional, calling it twice is harmless. */
#include <thread>
namespace {
struct Prewarm {
	std::thread th;
	bool        started = false;
	std::mutex  mu;
	void join() { std::lock_guard<std::mutex> lock(mu); if (th.joinable()) th.join(); }
	~Prewarm() { if (th.joinable()) th.join(); }        /* a process that exits early must not leave the thread inside the runtime */
} g_prewarm;
}

static void prewarm_join(void) { g_prewarm.join(); }

extern "C" void fixca_prewarm(void)
{
	std::lock_guard<std::mutex> lock(g_prewarm.mu);
	if (g_prewarm.started) return;
	g_prewarm.started = true;
	g_prewarm.th = std::thread([] {
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); return; }
		(void)hipFree(nullptr);                      /* creates the primary context of device 0 */
		fixca_phase("prewarm:device_ready");
	});
}

static int require_device(void)
{
	prewarm_join();
	int n = fixca_device_count();
	if (n < 0) return n;
	if (n == 0) return fixca_fail(FIXCA_E_NODEVICE, "no HIP device visible: the Fix-CA GPU path cannot run");
	return FIXCA_OK;
}

/* ---- the hot path ------------------------------------------------------- */

extern "C" int fixca_region_device(const void *d_src, void *d_dst, int W, int H, int bytes, int bpc,
                                   const FixCaParams *params, int x1, int x2, int y1, int y2,
                                   int show_progress, unsigned flags, void *stream)
{
	if (!d_src || !d_dst) return fixca_fail(FIXCA_E_INVALID, "null image pointer");
	if (d_src == d_dst) return fixca_fail(FIXCA_E_INVALID, "src and dst must not alias");
	FixcaGeom g;
	FixcaPlan pl;
	int rc = fixca_make_geom(W, H, bytes, bpc, params, x1, x2, y1, y2, show_progress, &g);
	if (rc) return rc;
	rc = fixca_make_plan(&g, flags, &pl);
	if (rc) return rc;
	HIP_TRY(fixca_launch(d_src, d_dst, &g, &pl, (hipStream_t)stream));
	if (g.preview)                               /* fix-ca.c:1322-1327, after the row's pixels are final */
		HIP_TRY(fixca_launch_centerline(d_dst, &g, (hipStream_t)stream));
	return FIXCA_OK;
}

extern "C" int fixca_batch_device(const void *const *d_src, void *const *d_dst, int n_images,
                                  int W, int H, int bytes, int bpc, const FixCaParams *params,
                                  unsigned flags, void *stream)
{
	if (!d_src || !d_dst || n_images < 0) return fixca_fail(FIXCA_E_INVALID, "bad batch arguments");
	FixcaGeom g;
	FixcaPlan pl;
	int rc = fixca_make_geom(W, H, bytes, bpc, params, 0, W, 0, H, 1, &g);
	if (rc) return rc;
	rc = fixca_make_plan(&g, n_images > 1 ? (flags | FIXCA_PLAN_BATCH) : flags, &pl);
	if (rc) return rc;
	for (int i = 0; i < n_images; ++i)
		if (!d_src[i] || !d_dst[i] || d_src[i] == d_dst[i])
			return fixca_fail(FIXCA_E_INVALID, "image %d: null or aliased pointers", i);
	return launch_many(d_src, d_dst, n_images, &g, &pl, (hipStream_t)stream, NULL);
}

static int launch_many(const void *const *d_src, void *const *d_dst, int n, const FixcaGeom *g, const FixcaPlan *pl,
                       hipStream_t stream, int *launches)
{
	int count = 0;
	if (pl->kernel == FIXCA_K_ROLL && g->y2 > g->y1) {
		for (int i = 0; i < n; i += ROLL_BATCH_MAX) {
			const int m = n - i < ROLL_BATCH_MAX ? n - i : ROLL_BATCH_MAX;
			if (g->bpp == 3) HIP_TRY(fixca_launch_roll_batch_rgb(d_src + i, d_dst + i, m, g, pl, stream));
			else             HIP_TRY(fixca_launch_roll_batch(d_src + i, d_dst + i, m, g, pl, stream));
			++count;
		}
	} else if ((pl->kernel == FIXCA_K_MOVE || pl->kernel == FIXCA_K_LIN || pl->kernel == FIXCA_K_CUB || pl->kernel == FIXCA_K_WCOL) && g->y2 > g->y1) {
		for (int i = 0; i < n; i += MOVE_BATCH_MAX) {
			const int m = n - i < MOVE_BATCH_MAX ? n - i : MOVE_BATCH_MAX;
			if (pl->kernel == FIXCA_K_MOVE)     HIP_TRY(fixca_launch_move_batch(d_src + i, d_dst + i, m, g, pl, stream));
			else if (pl->kernel == FIXCA_K_LIN) HIP_TRY(fixca_launch_lin_batch(d_src + i, d_dst + i, m, g, pl, stream));
			else if (pl->kernel == FIXCA_K_WCOL) HIP_TRY(fixca_launch_wide_batch(d_src + i, d_dst + i, m, g, pl, stream));
			else                                HIP_TRY(fixca_launch_cub_batch(d_src + i, d_dst + i, m, g, pl, stream));
			++count;
		}
	} else {
		for (int i = 0; i < n; ++i) {
			HIP_TRY(fixca_launch(d_src[i], d_dst[i], g, pl, stream));
			++count;
		}
	}
	if (launches) *launches = count;
	return FIXCA_OK;
}

/* fixca_region_host keeps its two device images between calls (grow only).  Per host thread and
 * keyed by the device they were allocated on, like the pipeline streams and the table cache: two
 * threads never share (or free) each other's images, and after fixca_set_device(n) the images of
 * the previous device are dropped instead of being handed to kernels of another GPU.  A thread
 * that exits without fixca_release_cache() leaves its images allocated until the process ends. */
namespace {
struct HostCache {
	void  *buf[2] = { nullptr, nullptr };
	size_t bytes = 0;
	int    device = -1;
};
thread_local HostCache g_cache;
}

extern "C" void fixca_roll_release_tables(void);
extern "C" void fixca_roll_release_tables_rgb(void);

extern "C" void fixca_release_cache(void)
{
	fixca_roll_release_tables();
	fixca_roll_release_tables_rgb();
	for (int i = 0; i < 2; ++i) {
		if (g_cache.buf[i]) (void)hipFree(g_cache.buf[i]);
		g_cache.buf[i] = nullptr;
	}
	g_cache.bytes = 0;
	g_cache.device = -1;
}

/* test hook: the calling thread's cached device images (NULL before the first host call) */
extern "C" __attribute__((visibility("default"))) void fixca_debug_host_cache(void **src_image, void **dst_image, size_t *nbytes)
{
	if (src_image) *src_image = g_cache.buf[0];
	if (dst_image) *dst_image = g_cache.buf[1];
	if (nbytes) *nbytes = g_cache.bytes;
}

/* the calling thread's two device images, at least `total` bytes each, on the current device */
static int host_cache_reserve(size_t total)
{
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	if (total > g_cache.bytes || dev != g_cache.device) {
		for (int i = 0; i < 2; ++i) {
			if (g_cache.buf[i]) (void)hipFree(g_cache.buf[i]);
			g_cache.buf[i] = nullptr;
		}
		g_cache.bytes = 0;
		g_cache.device = dev;
		HIP_TRY(hipMalloc(&g_cache.buf[0], total));
		HIP_TRY(hipMalloc(&g_cache.buf[1], total));
		g_cache.bytes = total;
	}
	return FIXCA_OK;
}

/* rows above / below an output row that its taps can reach: |y' - y| <= |s - 1| * (distance of the
 * farthest row from the centre) + |shift|, plus the cubic window */
static int vertical_reach(const FixcaGeom *g)
{
	double r = 0.0;
	for (int c = 0; c < 2; ++c) {
		const double far = fmax(fabs(0.0 - (double)g->yc), fabs((double)(g->H - 1) - (double)g->yc));
		const double d = fabs(g->s[c] - 1.0) * far + fabs(g->ys[c]);
		if (d > r) r = d;
	}
	if (!(r < (double)g->H)) return g->H;
	return (int)ceil(r) + 3;
}

static bool is_pinned(const void *p)
{
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError();                 /* an ordinary malloc'ed pointer is "invalid value" here */
		return false;
	}
	return a.type == hipMemoryTypeHost;
}

/* streams and events of the host entry's upload / compute / download pipeline */
namespace {
constexpr int PIPE_BANDS = 8;
struct Pipe {
	bool        ready = false;
	hipStream_t up = nullptr, run = nullptr, down = nullptr;
	hipEvent_t  uploaded[PIPE_BANDS], done[PIPE_BANDS];
};
thread_local Pipe g_pipe;
}

static int pipe_init(void)
{
	if (g_pipe.ready) return FIXCA_OK;
	HIP_TRY(hipStreamCreateWithFlags(&g_pipe.up, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&g_pipe.run, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&g_pipe.down, hipStreamNonBlocking));
	for (int i = 0; i < PIPE_BANDS; ++i) {
		HIP_TRY(hipEventCreateWithFlags(&g_pipe.uploaded[i], hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&g_pipe.done[i], hipEventDisableTiming));
	}
	g_pipe.ready = true;
	return FIXCA_OK;
}

/* the body of fixca_region_host; on a failure it may return with copies and kernels still queued */
static int region_host_queue(const unsigned char *src, unsigned char *dst, int W, int H, int bytes,
                             int bpc, const FixCaParams *params, int x1, int x2, int y1, int y2,
                             int show_progress, const FixcaGeom &g)
{
	int rc;
	const size_t pitch = (size_t)W * bytes, total = pitch * H;
	fixca_phase("region_host:entered");
	rc = host_cache_reserve(total);
	if (rc) return rc;
	fixca_phase("region_host:device_images_allocated");
	const char *ex = getenv("FIXCA_EXACT");
	const unsigned flags = (ex && *ex && strcmp(ex, "0") != 0) ? FIXCA_KERNEL_EXACT : FIXCA_KERNEL_AUTO;
	char *const d_src = static_cast<char *>(g_cache.buf[0]);
	char *const d_dst = static_cast<char *>(g_cache.buf[1]);

	/* only the source rows the requested rows can reach are uploaded (the preview asks for a
	 * window of rows, fix-ca.c:656) */
	const int reach = vertical_reach(&g);
	const int lo = y1 - reach > 0 ? y1 - reach : 0;
	const int hi = y2 + reach < H ? y2 + reach : H;
	const int rows = y2 - y1;

	if (rows < 256 || fixca_debug_flag("FIXCA_NO_PIPELINE") || !is_pinned(src) || !is_pinned(dst)) {
		HIP_TRY(hipMemcpy(d_src + (size_t)lo * pitch, src + (size_t)lo * pitch, (size_t)(hi - lo) * pitch, hipMemcpyHostToDevice));
		fixca_phase("region_host:uploaded");
		rc = fixca_region_device(d_src, d_dst, W, H, bytes, bpc, params, x1, x2, y1, y2, show_progress, flags, NULL);
		if (rc) return rc;
		fixca_phase("region_host:kernel_launched");
		HIP_TRY(hipMemcpy(dst + (size_t)y1 * pitch, d_dst + (size_t)y1 * pitch, (size_t)rows * pitch, hipMemcpyDeviceToHost));
		HIP_TRY(hipDeviceSynchronize());
		fixca_phase("region_host:downloaded");
		return FIXCA_OK;
	}

	/* Page-locked buffers (fixca_host_alloc): upload, kernel and download of consecutive row
	 * bands overlap on three streams, so the call takes about one transfer instead of two plus
	 * the kernel (PCIe is full duplex). */
	rc = pipe_init();
	if (rc) return rc;
	fixca_phase("region_host:streams_ready");
	int bh = (rows + PIPE_BANDS - 1) / PIPE_BANDS;
	bh = (bh + 3) & ~3;
	int up_to = lo, band = 0;
	for (int yb = y1; yb < y2; yb += bh, ++band) {
		const int ye = yb + bh < y2 ? yb + bh : y2;
		const int need_hi = ye + reach < H ? ye + reach : H;
		if (need_hi > up_to) {
			HIP_TRY(hipMemcpyAsync(d_src + (size_t)up_to * pitch, src + (size_t)up_to * pitch,
			                       (size_t)(need_hi - up_to) * pitch, hipMemcpyHostToDevice, g_pipe.up));
			up_to = need_hi;
		}
		HIP_TRY(hipEventRecord(g_pipe.uploaded[band], g_pipe.up));
		HIP_TRY(hipStreamWaitEvent(g_pipe.run, g_pipe.uploaded[band], 0));
		rc = fixca_region_device(d_src, d_dst, W, H, bytes, bpc, params, x1, x2, yb, ye, show_progress, flags, g_pipe.run);
		if (rc) return rc;
		HIP_TRY(hipEventRecord(g_pipe.done[band], g_pipe.run));
		HIP_TRY(hipStreamWaitEvent(g_pipe.down, g_pipe.done[band], 0));
		HIP_TRY(hipMemcpyAsync(dst + (size_t)yb * pitch, d_dst + (size_t)yb * pitch, (size_t)(ye - yb) * pitch,
		                       hipMemcpyDeviceToHost, g_pipe.down));
	}
	fixca_phase("region_host:bands_queued");
	HIP_TRY(hipStreamSynchronize(g_pipe.down));
	HIP_TRY(hipStreamSynchronize(g_pipe.run));
	HIP_TRY(hipStreamSynchronize(g_pipe.up));
	fixca_phase("region_host:downloaded");
	return FIXCA_OK;
}

extern "C" int fixca_region_host(const unsigned char *src, unsigned char *dst, int W, int H, int bytes,
                                 int bpc, const FixCaParams *params, int x1, int x2, int y1, int y2,
                                 int show_progress)
{
	if (!src || !dst) return fixca_fail(FIXCA_E_INVALID, "null image pointer");
	FixcaGeom g;
	int rc = fixca_make_geom(W, H, bytes, bpc, params, x1, x2, y1, y2, show_progress, &g);
	if (rc) return rc;                       /* argument errors are reported without touching the GPU */
	rc = require_device();
	if (rc) return rc;
	if (y2 <= y1) return FIXCA_OK;
	rc = region_host_queue(src, dst, W, H, bytes, bpc, params, x1, x2, y1, y2, show_progress, g);
	if (rc != FIXCA_OK) {
		/* One exit for every failure: nothing of this call may still be running when the caller
		 * gets its buffers back (on a non-zero return the plug-in's own loop
		 * writes the same dst).  Queued downloads would land in dst and queued uploads would read
		 * src behind the caller's back, so the device is drained first; the error text of the
		 * first failure is kept. */
		(void)hipDeviceSynchronize();
		(void)hipGetLastError();
	}
	return rc;
}

/* ---- the preview caller (fix-ca.c:617-679) --------------------------------- */

static int preview_args_check(int W, int H, int bytes, int bpc, int x, int y, int width, int height)
{
	const int sb = bpc < 0 ? -bpc : bpc;
	if (!(bpc == 1 || bpc == 2 || bpc == 4 || bpc == -4 || bpc == -8) || (bytes != 3 * sb && bytes != 4 * sb))
		return fixca_fail(FIXCA_E_UNSUPPORTED, "bpc=%d bytes=%d: implemented are RGB/RGBA with u8, u16, u32, float or double samples", bpc, bytes);
	if (W <= 0 || H <= 0 || x < 0 || y < 0 || width < 0 || height < 0 || x > W - width || y > H - height)
		return fixca_fail(FIXCA_E_INVALID, "preview window %dx%d at (%d,%d) outside the %dx%d image", width, height, x, y, W, H);
	return FIXCA_OK;
}

extern "C" int fixca_preview_device(const void *d_img, void *d_prev, int W, int H, int bytes, int bpc,
                                    int x, int y, int width, int height, void *stream)
{
	if (!d_img || !d_prev) return fixca_fail(FIXCA_E_INVALID, "null image pointer");
	int rc = preview_args_check(W, H, bytes, bpc, x, y, width, height);
	if (rc) return rc;
	HIP_TRY(fixca_launch_preview(d_img, d_prev, W, bytes, bpc, x, y, width, height, (hipStream_t)stream));
	return FIXCA_OK;
}

static int preview_host_queue(const unsigned char *src, unsigned char *prev, int W, int H, int bytes, int bpc,
                              const FixCaParams *params, int x, int y, int width, int height, const FixcaGeom &g)
{
	const size_t pitch = (size_t)W * bytes, total = pitch * H;
	const int sb = bpc < 0 ? -bpc : bpc;
	const size_t prev_bytes = (size_t)width * height * (bytes / sb);
	int rc = host_cache_reserve(total);
	if (rc) return rc;
	char *const d_src = static_cast<char *>(g_cache.buf[0]);
	char *const d_dst = static_cast<char *>(g_cache.buf[1]);
	const int reach = vertical_reach(&g);
	const int lo = y - reach > 0 ? y - reach : 0;
	const int hi = y + height + reach < H ? y + height + reach : H;
	HIP_TRY(hipMemcpy(d_src + (size_t)lo * pitch, src + (size_t)lo * pitch, (size_t)(hi - lo) * pitch, hipMemcpyHostToDevice));
	rc = fixca_region_device(d_src, d_dst, W, H, bytes, bpc, params, 0, W, y, y + height, 0, FIXCA_KERNEL_AUTO, NULL);
	if (rc) return rc;
	/* the window is converted into the (already consumed) upload area: no third buffer */
	void *d_prev = d_src;
	HIP_TRY(fixca_launch_preview(d_dst, d_prev, W, bytes, bpc, x, y, width, height, NULL));
	HIP_TRY(hipMemcpy(prev, d_prev, prev_bytes, hipMemcpyDeviceToHost));
	return FIXCA_OK;
}

extern "C" int fixca_preview_host(const unsigned char *src, unsigned char *prev, int W, int H, int bytes, int bpc,
                                  const FixCaParams *params, int x, int y, int width, int height)
{
	if (!src || !prev) return fixca_fail(FIXCA_E_INVALID, "null image pointer");
	int rc = preview_args_check(W, H, bytes, bpc, x, y, width, height);
	if (rc) return rc;
	FixcaGeom g;
	rc = fixca_make_geom(W, H, bytes, bpc, params, 0, W, y, y + height, 0, &g);
	if (rc) return rc;
	rc = require_device();
	if (rc) return rc;
	if (width == 0 || height == 0) return FIXCA_OK;
	rc = preview_host_queue(src, prev, W, H, bytes, bpc, params, x, y, width, height, g);
	if (rc != FIXCA_OK) {
		(void)hipDeviceSynchronize();
		(void)hipGetLastError();
	}
	return rc;
}

/* ---- synthetic input ----------------------------------------- */

extern "C" int fixca_noise_device(void *d_dst, size_t nbytes, unsigned long long seed, void *stream)
{
	if (!d_dst) return fixca_fail(FIXCA_E_INVALID, "null pointer");
	HIP_TRY(fixca_launch_noise(d_dst, nbytes, seed, (hipStream_t)stream));
	return FIXCA_OK;
}

/* ---- timing on the launch stream ---------------------------------------- */

namespace {
/* two timing events that are destroyed on every way out */
struct EventPair {
	hipEvent_t e0 = nullptr, e1 = nullptr;
	hipError_t create()
	{
		hipError_t e = hipEventCreate(&e0);
		return e != hipSuccess ? e : hipEventCreate(&e1);
	}
	~EventPair()
	{
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
	}
};
/* launches not timed in front of the timed ones: the clocks come down during any pause of the device (the bench's
 * host-side oracle check is one of half a second) and take a few milliseconds of work to come back; round 3 timed
 * 200 launches of 45 us with no warm-up and read them 4 us slow */
constexpr int TIMING_WARMUP = 32;
}

extern "C" int fixca_time_launches(const void *const *d_src, void *const *d_dst, int n_pairs, int launches,
                                   int W, int H, int bytes, int bpc, const FixCaParams *params,
                                   unsigned flags, void *stream, double *ms_per_launch)
{
	if (!d_src || !d_dst || n_pairs < 1 || launches < 1 || !ms_per_launch)
		return fixca_fail(FIXCA_E_INVALID, "bad timing arguments");
	FixcaGeom g;
	FixcaPlan pl;
	int rc = fixca_make_geom(W, H, bytes, bpc, params, 0, W, 0, H, 1, &g);
	if (rc) return rc;
	rc = fixca_make_plan(&g, flags, &pl);
	if (rc) return rc;
	hipStream_t st = (hipStream_t)stream;
	EventPair ev;
	HIP_TRY(ev.create());
	const int warm = launches < TIMING_WARMUP ? launches : TIMING_WARMUP;
	for (int i = 0; i < warm; ++i)
		HIP_TRY(fixca_launch(d_src[i % n_pairs], d_dst[i % n_pairs], &g, &pl, st));
	HIP_TRY(hipEventRecord(ev.e0, st));
	for (int i = 0; i < launches; ++i)
		HIP_TRY(fixca_launch(d_src[(warm + i) % n_pairs], d_dst[(warm + i) % n_pairs], &g, &pl, st));
	HIP_TRY(hipEventRecord(ev.e1, st));
	HIP_TRY(hipEventSynchronize(ev.e1));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
	*ms_per_launch = (double)ms / launches;
	return FIXCA_OK;
}

extern "C" int fixca_time_copy(const void *const *d_src, void *const *d_dst, int n_pairs, int launches,
                               size_t nbytes, void *stream, double *ms_per_launch)
{
	if (!d_src || !d_dst || n_pairs < 1 || launches < 1 || !ms_per_launch)
		return fixca_fail(FIXCA_E_INVALID, "bad timing arguments");
	hipStream_t st = (hipStream_t)stream;
	EventPair ev;
	HIP_TRY(ev.create());
	const int warm = launches < TIMING_WARMUP ? launches : TIMING_WARMUP;
	for (int i = 0; i < warm; ++i)
		HIP_TRY(fixca_launch_copy(d_src[i % n_pairs], d_dst[i % n_pairs], nbytes, st));
	HIP_TRY(hipEventRecord(ev.e0, st));
	for (int i = 0; i < launches; ++i)
		HIP_TRY(fixca_launch_copy(d_src[(warm + i) % n_pairs], d_dst[(warm + i) % n_pairs], nbytes, st));
	HIP_TRY(hipEventRecord(ev.e1, st));
	HIP_TRY(hipEventSynchronize(ev.e1));
	float ms = 0.f;
	HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
	*ms_per_launch = (double)ms / launches;
	return FIXCA_OK;
}
