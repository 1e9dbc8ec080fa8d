/*
 * fixca_host.h — plain-C host layer of the Fix-CA plug-in on top of libfixca_hip.so.
 *
 * It keeps the reference's entry names and argument lists for the path
 * (fix-ca.c:101-109):
 *     fix_ca_region()  same 12 arguments; now a thin call into the HIP shim
 *     fix_ca()         the drawable shell; here on a caller-supplied packed buffer
 *                      (FixcaDrawable) because libgimp/GEGL are not in this image
 *     run()            the PDB entry; its non-interactive argument handling is
 *                      fixca_unmarshal_noninteractive() below
 *     preview_update() the second caller of fix_ca_region (fix-ca.c:617-679): its pixel
 *                      work between the GEGL fetch and gimp_preview_draw_buffer is
 *                      fixca_preview_pixels() below
 * glib's typedefs are spelled out (gint = int, guchar = unsigned char,
 * gboolean = int, gdouble = double) so the file builds with or without glib.
 */
#ifndef FIXCA_HOST_H
#define FIXCA_HOST_H

#include <stdint.h>
#include "../../include/fixca_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status of the last fix_ca_region() call: the reference's function is void, so
 * the shell reads this to turn a GPU failure into fix_ca()'s -1. */
extern int fixca_host_status;

void fix_ca_region(unsigned char *srcPTR, unsigned char *dstPTR,
                   int orig_width, int orig_height, int bytes, int bpc,
                   FixCaParams *params, int x1, int x2, int y1, int y2,
                   int show_progress);

/* What fix_ca() needs to know about a drawable (fix-ca.c:340-367), GIMP-free. */
typedef struct FixcaDrawable {
	unsigned char *pixels;   /* packed, interleaved, width*height*bpp bytes, updated in place */
	int width, height;       /* gimp_drawable_width/height                                   */
	int bpp;                 /* babl_format_get_bytes_per_pixel                               */
	int bpc;                 /* color_size(): 1,2,4,8,-4,-8, or -99 = unsupported             */
	int sel_x, sel_y, sel_w, sel_h;   /* gimp_drawable_mask_intersect rectangle               */
} FixcaDrawable;

/* fix_ca() (fix-ca.c:332-394): 0 on success, -1 on empty selection, unsupported
 * colour type or GPU failure.  Processes the rows of the selection rectangle. */
int fix_ca(FixcaDrawable *drawable, FixCaParams *params);

/*
 * preview_update()'s pixels (fix-ca.c:647-671): the drawable's rows [y, y+height) corrected with
 * show_progress FALSE, the visible window (x, y, width, height) copied to prevImg, samples wider
 * than 8 bits converted to 8 bits (get_pixel -> set_pixel(.., 1)).  prevImg must hold
 * width * height * bpp / |bpc| bytes (what gimp_preview_draw_buffer is then given with rowstride
 * width * bpp / |bpc|, fix-ca.c:673).  The drawable is not modified.  0 on success, -1 otherwise.
 */
int fixca_preview_pixels(const FixcaDrawable *drawable, FixCaParams *params,
                         int x, int y, int width, int height, unsigned char *prevImg);

/* 1: fix_ca() page-locks its result image with fixca_host_alloc (long-lived hosts); 0 (default): malloc, as the
 * reference's g_new (fix-ca.c:366-367): a plug-in process lives for one call. */
extern int fixca_host_use_pinned;

/* color_size() (fix-ca.c:681-711) from a babl format name and bytes per pixel. */
int fixca_color_size(const char *babl_format_name, int bytes_per_pixel);

/*
 * One PDB argument as run() sees it.  Layout-compatible with libgimp 2.10's
 * GimpParam for the members run() reads (type, then a 32-byte union whose first
 * bytes are d_int32 / d_int8 / d_float).
 */
typedef struct FixcaPdbArg {
	int32_t type;
	union {
		int32_t d_int32;
		int8_t  d_int8;
		double  d_float;
		double  pad_[4];
	} data;
} FixcaPdbArg;

enum { FIXCA_PDB_SUCCESS = 3, FIXCA_PDB_CALLING_ERROR = 1, FIXCA_PDB_CANCEL = 4 };   /* GimpPDBStatusType */

/*
 * run()'s GIMP_RUN_NONINTERACTIVE branch (fix-ca.c:231-296): nparams in 5..12,
 * missing arguments default to lens -1, interpolation NONE, directional 0; every
 * amount must lie in +-30 and interpolation in 0..2.  lens_x / lens_y are read
 * through d_int32 exactly as the reference does (fix-ca.c:254,258), which is what
 * makes the shipped test run with centre (0,0).
 * Returns FIXCA_PDB_SUCCESS or FIXCA_PDB_CALLING_ERROR.
 */
int fixca_unmarshal_noninteractive(int nparams, const FixcaPdbArg *param, FixCaParams *out);

#ifdef __cplusplus
}
#endif
#endif
