/*
 * fixca_hip.h — C-ABI of libfixca_hip.so, the MI355X (gfx950) implementation of
 * the Fix-CA plug-in's pixel loop.
 *
 * The boundary is the reference's own function
 *
 *     static void fix_ca_region (guchar *srcPTR, guchar *dstPTR,
 *                                gint orig_width, gint orig_height,
 *                                gint bytes, gint bpc, FixCaParams *params,
 *                                gint x1, gint x2, gint y1, gint y2,
 *                                gboolean show_progress);     fix-ca.c:105-109, :998-1001
 *
 * called by fix_ca() (fix-ca.c:373, whole drawable) and preview_update()
 * (fix-ca.c:656, visible rows).  Every entry below takes plain pointers and
 * sizes; no GIMP, glib, torch or HIP types appear in a signature (a stream is
 * passed as void*).  All entries return FIXCA_OK (0) or a negative FIXCA_E_*
 * code and never abort the process; fixca_last_error() gives the text.  There is
 * no CPU fallback inside the library: if the GPU path cannot run, the call fails.
 */
#ifndef FIXCA_HIP_H
#define FIXCA_HIP_H

#include <stddef.h>
#include "fixca_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FIXCA_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------- */
#define FIXCA_OK              0
#define FIXCA_E_INVALID      (-1)  /* null pointer, non-positive size, bad row range   */
#define FIXCA_E_UNSUPPORTED  (-2)  /* x1 != 0 or x2 != orig_width (reference indexes
                                      wrongly there); bpc/bytes pair not
                                      implemented; preview saturation boost           */
#define FIXCA_E_DOMAIN       (-3)  /* max_dim + amount <= 0, |amount| > 30, interp ∉ 0..2 */
#define FIXCA_E_HIP          (-4)  /* a HIP runtime call failed                        */
#define FIXCA_E_NODEVICE     (-5)  /* no gfx950 device visible                         */

/* ---- kernel selection flags (device entries) ---------------------------- */
#define FIXCA_KERNEL_AUTO   0u   /* fast row-scan kernel when its plan fits, else exact */
#define FIXCA_KERNEL_EXACT  1u   /* fp64 kernel, operation for operation the reference's
                                    arithmetic: bit-exact in all three modes            */
#define FIXCA_KERNEL_FAST   2u   /* fast kernel or FIXCA_E_UNSUPPORTED (tests, bench)   */
#define FIXCA_KERNEL_MASK   3u
#define FIXCA_PLAN_BATCH    4u   /* planning hint: the call is one of many of the same geometry launched back to
                                    back (fixca_batch_device sets it): rows per wave are chosen for a full GPU */

/*
 * Replaces the body of fix_ca_region (fix-ca.c:998-1348) for host buffers.
 * Same argument order and meaning.  Copies src to the device, runs the kernel on
 * rows [y1,y2), copies exactly those rows back into dst; rows of dst outside
 * [y1,y2) are not written (set_data, fix-ca.c:864-871, writes only row y).
 * Buffers are borrowed for the duration of the call.  Synchronous.
 * Kernel choice: FIXCA_KERNEL_AUTO, or FIXCA_KERNEL_EXACT when the environment
 * variable FIXCA_EXACT is set to a non-zero value.
 */
FIXCA_API int fixca_region_host(const unsigned char *src, unsigned char *dst,
                                int orig_width, int orig_height, int bytes, int bpc,
                                const FIXCA_PARAMS_T *params,
                                int x1, int x2, int y1, int y2, int show_progress);

/*
 * The same call on buffers already resident in device memory (hipMalloc'd or a
 * torch tensor's data_ptr).  Asynchronous on `stream` (a hipStream_t, NULL = the
 * default stream); the caller synchronises.  src and dst must not overlap.
 */
FIXCA_API int fixca_region_device(const void *d_src, void *d_dst,
                                  int orig_width, int orig_height, int bytes, int bpc,
                                  const FIXCA_PARAMS_T *params,
                                  int x1, int x2, int y1, int y2, int show_progress,
                                  unsigned flags, void *stream);

/*
 * A batch of independent images of one geometry and one parameter block (the
 * "512 images" configuration): image i is d_src[i] -> d_dst[i], all rows.  The
 * pointer arrays are host arrays of device pointers (read during the call, not kept).
 * Asynchronous on `stream`.  Launches: the row-rolling kernel (6) and the kernels of the
 * wider sample types (7 - 10) take up to 128 images per kernel launch, every other kernel
 * one image per launch, all back to back on `stream`.
 *
 * Stream capture: once a geometry's tables exist (one earlier call with the same size,
 * parameters and sample format on any stream), the call makes no allocation and no
 * synchronisation and may be captured into a graph.  The tables a captured launch reads
 * are then kept until fixca_release_cache(); a thread holds at most 8 such geometries
 * (the 9th call under capture fails with FIXCA_E_HIP).  A capture must not be the FIRST
 * call of a geometry: that call allocates the tables.
 */
FIXCA_API int fixca_batch_device(const void *const *d_src, void *const *d_dst, int n_images,
                                 int orig_width, int orig_height, int bytes, int bpc,
                                 const FIXCA_PARAMS_T *params, unsigned flags, void *stream);

/*
 * Which kernel the planner picks for a call, without launching anything.  Writes a short
 * description ("roll cubic bpp=4 oc=240 th=57 ring=16 ...") into buf and returns the
 * kernel id, or a negative FIXCA_E_* if the call itself would be refused:
 *   1 exact-fp64      one thread per pixel, the reference's arithmetic operation for operation
 *                     (every sample type; FIXCA_KERNEL_EXACT, and the fallback of the wide types)
 *   2 fast-none       8-bit register-window gather (images narrower than 256 pixels, shifts the ring kernels refuse)
 *   3 fast-interp     8-bit register-window LINEAR / CUBIC (likewise)
 *   4 scan            8-bit LDS-ring row scan (scales far from 1)
 *   6 roll            8-bit LDS-ring row-rolling kernel, RGBA8 and RGB8 of any width from 256 pixels: the main kernel
 *   7 wide-roll-fp64  u16 / u32 / float / double, LINEAR and CUBIC, fp64 rolling window
 *   8 move-none       u16 / u32 / float / double, NONE: LDS-ring byte mover
 *   9 ring-linear-fp64 u32 / float / double, RGBA and RGB, LINEAR: LDS ring + fp64 bilinear
 *  10 ring-cubic-fp64  RGBA double (and float in batches), CUBIC: LDS ring + a rolling window per lane
 * (5 is retired.)  Usable with no GPU: no pixel kernel is launched.  Two questions go to the
 * device when there is one: for LINEAR / CUBIC on the wide sample types the current device's
 * occupancy (this may initialise the runtime; without a device the planner assumes a 256-CU
 * part), and, the first time a geometry's rows are not whole dwords (RGB8 of a width that is
 * not a multiple of 4, RGB u16 of odd width), whether the device moves dwords at byte
 * addresses faithfully: a one-wave probe on a 4-KiB scratch, once per device and process
 * (without a device: assumed).  Like the first call of a geometry, such a first call must
 * not happen inside a stream capture.
 */
FIXCA_API int fixca_plan_describe(int orig_width, int orig_height, int bytes, int bpc,
                                  const FIXCA_PARAMS_T *params, int x1, int x2, int y1, int y2,
                                  int show_progress, unsigned flags, char *buf, size_t buflen);

/*
 * Validates a parameter block the way run() does for non-interactive calls
 * (fix-ca.c:279-295): every amount within ±FIXCA_INPUT_MAX, interpolation 0..2.
 */
FIXCA_API int fixca_params_check(const FIXCA_PARAMS_T *params);

/* ---- device plumbing for C hosts (the plug-in has no torch) ------------- */
FIXCA_API int   fixca_device_count(void);             /* >= 0, or FIXCA_E_HIP         */
FIXCA_API int   fixca_set_device(int ordinal);
FIXCA_API void *fixca_dev_alloc(size_t nbytes);        /* NULL on failure              */
FIXCA_API int   fixca_dev_free(void *d_ptr);
FIXCA_API int   fixca_copy_h2d(void *d_dst, const void *h_src, size_t nbytes, void *stream);
FIXCA_API int   fixca_copy_d2h(void *h_dst, const void *d_src, size_t nbytes, void *stream);
FIXCA_API int   fixca_stream_sync(void *stream);
/*
 * Page-locked host memory for the two images fix_ca() allocates (g_new at fix-ca.c:366-367):
 * gegl_buffer_get / gegl_buffer_set then work directly on memory the GPU's DMA engines can reach,
 * and fixca_region_host's copies run at PCIe speed instead of through a staging buffer.
 * Ordinary malloc'ed buffers keep working; these are an optimisation.
 */
FIXCA_API void *fixca_host_alloc(size_t nbytes);       /* NULL on failure              */
FIXCA_API int   fixca_host_free(void *h_ptr);
FIXCA_API void  fixca_release_cache(void);             /* frees fixca_region_host's buffers and the kernel tables */
/*
 * Optional: starts the HIP runtime's initialisation (56-227 ms in a fresh process) on a background thread and
 * returns at once; the next fixca_region_host / fixca_preview_host joins it.  A plug-in calls it first thing in run()
 * (fix-ca.c:197), so that the initialisation overlaps with gimp_drawable_get and gegl_buffer_get (fix-ca.c:215, :369).
 */
FIXCA_API void  fixca_prewarm(void);

/* ---- the preview caller --------------------------------------------------- */
/*
 * preview_update() (fix-ca.c:617-679) between gegl_buffer_get and gimp_preview_draw_buffer:
 * corrects rows [y, y + height) of the whole image with show_progress = FALSE (the lens-centre
 * lines are drawn, fix-ca.c:1326; saturation != 0 is refused), copies the visible window
 * (x, y, width, height) into prev and, when the samples are wider than 8 bits, converts them with
 * get_pixel -> set_pixel(.., 1) (fix-ca.c:665-669).  prev receives width * height * bytes bytes for
 * bpc = 1 and width * height * (bytes / |bpc|) bytes otherwise: what gimp_preview_draw_buffer is
 * given (rowstride width * bytes / |bpc|, fix-ca.c:673).
 *
 * fixca_preview_device is the crop + conversion alone, on an image already on the device (d_prev
 * must not overlap the window it is cut from).
 */
FIXCA_API int fixca_preview_host(const unsigned char *src, unsigned char *prev,
                                 int orig_width, int orig_height, int bytes, int bpc,
                                 const FIXCA_PARAMS_T *params, int x, int y, int width, int height);
FIXCA_API int fixca_preview_device(const void *d_img, void *d_prev,
                                   int orig_width, int orig_height, int bytes, int bpc,
                                   int x, int y, int width, int height, void *stream);

/* ---- synthetic input ------------------------------------------------------ */
/*
 * Fills d_dst (8-byte aligned device memory) with the first nbytes of SURVEY.md §8(d)'s noise(seed):
 * the SplitMix64 stream, word k >= 1 = mix(seed + k * 0x9E3779B97F4A7C15), little-endian.  Used by
 * bench.py to make config 5's 512 images (image i = noise(0xF1CA1000 + i)) without a host copy.
 */
FIXCA_API int fixca_noise_device(void *d_dst, size_t nbytes, unsigned long long seed, void *stream);

/* ---- timing on the launch stream (bench.py's roofline leg) -------------- */
/*
 * Runs `launches` back-to-back kernel launches over the n_pairs src/dst pairs in
 * rotation on `stream`, bracketed by HIP events recorded on that same stream, and
 * returns the mean milliseconds per launch in *ms_per_launch.  Up to 32 launches of the
 * same rotation run untimed in front (the device clocks drop during any pause).
 */
FIXCA_API int fixca_time_launches(const void *const *d_src, void *const *d_dst, int n_pairs,
                                  int launches, int orig_width, int orig_height, int bytes,
                                  int bpc, const FIXCA_PARAMS_T *params, unsigned flags,
                                  void *stream, double *ms_per_launch);

/*
 * A plain 16-byte-per-lane device copy of nbytes (the on-box bandwidth ceiling the
 * kernel is compared with), timed the same way.
 */
FIXCA_API int fixca_time_copy(const void *const *d_src, void *const *d_dst, int n_pairs,
                              int launches, size_t nbytes, void *stream, double *ms_per_launch);

/*
 * With FIXCA_PHASES=1 in the environment: writes "fixca-phase <name> <ms>" to stderr, ms of CLOCK_MONOTONIC since
 * FIXCA_T0_NS (a clock reading of the spawning process) or since the first stamp; a no-op otherwise.  The library
 * stamps the phases of fixca_region_host; a host may add its own (fixca_cli does).
 */
FIXCA_API void fixca_phase(const char *name);

FIXCA_API const char *fixca_last_error(void);          /* thread-local, never NULL     */
FIXCA_API int         fixca_abi_version(void);         /* 1                            */

/* ---- test hooks (used by tests/ only) ----------------------------------------------------
 * fixca_debug_sched: the tile schedule of a batch launch of the row-rolling kernel as 6 ints
 * {ncols, rows, pool_chunks, pool_waves, pool_blocks, th} (no GPU needed).  fixca_debug_host_cache: the
 * calling thread's cached device images of fixca_region_host. */
FIXCA_API int  fixca_debug_sched(int orig_width, int orig_height, int bytes, const FIXCA_PARAMS_T *params,
                                 int n_images, int *out6);
FIXCA_API void fixca_debug_host_cache(void **src_image, void **dst_image, size_t *nbytes);

#ifdef __cplusplus
}
#endif
#endif /* FIXCA_HIP_H */
