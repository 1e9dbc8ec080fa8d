/*
 * fixca_host.c — C host layer: the reference's entry points for the pixel path,
 * calling the HIP shim.  See fixca_host.h.  No CPU fallback: a GPU failure is
 * reported (stderr + status) and surfaces as fix_ca() == -1, which run() maps to
 * GIMP_PDB_CALLING_ERROR (fix-ca.c:315-316).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fixca_host.h"

int fixca_host_status = FIXCA_OK;
/* 1: fix_ca() page-locks its result image (fixca_host_alloc).  Worth it for a long-lived host that calls again and
 * again (the copies then overlap with the kernel in row bands); a plug-in process lives for ONE call, and page-locking
 * 96 MB costs more than the pipelining saves (bench.py's cold_call_ms.phases), so the default is off. */
int fixca_host_use_pinned = 0;

void fix_ca_region(unsigned char *srcPTR, unsigned char *dstPTR,
                   int orig_width, int orig_height, int bytes, int bpc,
                   FixCaParams *params, int x1, int x2, int y1, int y2,
                   int show_progress)
{
	fixca_host_status = fixca_region_host(srcPTR, dstPTR, orig_width, orig_height, bytes, bpc,
	                                      params, x1, x2, y1, y2, show_progress);
	if (fixca_host_status != FIXCA_OK)
		fprintf(stderr, "fix-ca (HIP): %s (code %d)\n", fixca_last_error(), fixca_host_status);
}

int fixca_color_size(const char *name, int bpp)
{
	/* fix-ca.c:681-711 */
	if (!name) return -99;
	if (strstr(name, "double")) return -8;
	if (strstr(name, "float")) return -4;
	if (strstr(name, "u15")) return -99;
	if (!strstr(name, " u")) return -99;
	if (bpp > 32) return -99;
	if (bpp >= 24) return 8;
	if (bpp >= 12) return 4;
	if (bpp >= 6) return 2;
	if (bpp >= 3) return 1;
	return -99;
}

int fix_ca(FixcaDrawable *d, FixCaParams *params)
{
	unsigned char *dest;
	size_t total;
	int pinned;

	if (!d || !d->pixels || !params) return -1;
	if (d->sel_w <= 0 || d->sel_h <= 0) return -1;          /* mask_intersect failed (fix-ca.c:340) */
	if (d->bpc <= -99) {                                    /* fix-ca.c:355-358 */
		fprintf(stderr, "fix-ca: Invalid color type!\n");
		return -1;
	}
	total = (size_t)d->width * d->height * d->bpp;
	/* page-locked like the reference's g_new'ed destImg would be with fixca_host_alloc:
	 * with a page-locked source too, fixca_region_host pipelines upload, kernel and download */
	dest = fixca_host_use_pinned ? fixca_host_alloc(total) : NULL;
	pinned = dest != NULL;
	if (!dest) dest = malloc(total);                        /* g_new at fix-ca.c:367 */
	if (!dest) return -1;
	fixca_phase("fix_ca:result_image_allocated");

	/* FIX, NOT REPLICATE.  The reference is broken for every selection that is not the whole
	 * drawable: it fetches the selection rectangle tightly packed at offset 0 of srcImg (fix-ca.c:369-370,
	 * GEGL_AUTO_ROWSTRIDE) and then indexes that buffer with IMAGE rows and the image width (fix-ca.c:373-374,
	 * :855), so a full-width selection with y != 0 reads rows that were never fetched, and x != 0 is wrong three
	 * more ways.  This mirror does the sane thing instead: the drawable is held whole, rows
	 * [sel_y, sel_y + sel_h) of it are corrected in place of the image (what a user selecting a band of rows
	 * expects), and a selection narrower than the image is refused by the shim (FIXCA_E_UNSUPPORTED).  For the
	 * whole drawable - the only case in which the reference's output is defined - the two agree. */
	fix_ca_region(d->pixels, dest, d->width, d->height, d->bpp, d->bpc, params,
	              d->sel_x, d->sel_x + d->sel_w, d->sel_y, d->sel_y + d->sel_h, 1);
	if (fixca_host_status != FIXCA_OK) {
		if (pinned) fixca_host_free(dest); else free(dest);
		return -1;
	}
	fixca_phase("fix_ca:region_done");
	/* gegl_buffer_set of the rectangle + merge_shadow (fix-ca.c:376-385) */
	memcpy(d->pixels + (size_t)d->sel_y * d->width * d->bpp,
	       dest + (size_t)d->sel_y * d->width * d->bpp,
	       (size_t)d->sel_h * d->width * d->bpp);
	if (pinned) fixca_host_free(dest); else free(dest);
	return 0;
}

int fixca_preview_pixels(const FixcaDrawable *d, FixCaParams *params,
                         int x, int y, int width, int height, unsigned char *prevImg)
{
	if (!d || !d->pixels || !params || !prevImg) return -1;
	if (d->bpc <= -99) return -1;                           /* fix-ca.c:642-643: the preview stays as it is */
	fixca_host_status = fixca_preview_host(d->pixels, prevImg, d->width, d->height, d->bpp, d->bpc,
	                                       params, x, y, width, height);
	if (fixca_host_status != FIXCA_OK) {
		fprintf(stderr, "fix-ca (HIP): %s (code %d)\n", fixca_last_error(), fixca_host_status);
		return -1;
	}
	return 0;
}

int fixca_unmarshal_noninteractive(int nparams, const FixcaPdbArg *param, FixCaParams *out)
{
	if (!param || !out || nparams < 5 || nparams > 12)     /* fix-ca.c:231-235 */
		return FIXCA_PDB_CALLING_ERROR;
	memset(out, 0, sizeof *out);
	out->update_preview = 1;
	out->blue = param[3].data.d_float;
	out->red = param[4].data.d_float;
	out->lens_x = (nparams < 6) ? -1.0 : param[5].data.d_int32;   /* sic: d_int32 of a FLOAT arg */
	out->lens_y = (nparams < 7) ? -1.0 : param[6].data.d_int32;
	out->interpolation = (nparams < 8) ? FIXCA_INTERP_NONE : param[7].data.d_int8;
	out->x_blue = (nparams < 9) ? 0.0 : param[8].data.d_float;
	out->x_red = (nparams < 10) ? 0.0 : param[9].data.d_float;
	out->y_blue = (nparams < 11) ? 0.0 : param[10].data.d_float;
	out->y_red = (nparams < 12) ? 0.0 : param[11].data.d_float;
	if (fixca_params_check(out) != FIXCA_OK) {
		fprintf(stderr, "fix-ca: Parameter out of range!\n");
		return FIXCA_PDB_CALLING_ERROR;
	}
	return FIXCA_PDB_SUCCESS;
}
