/*
 * fixca_cli — GIMP-less driver for the host layer: what tests/test-fix-ca.scm does
 * through GIMP (load, Test-Fix-CA non-interactive, save), on binary PPM (P6, RGB8)
 * or PAM (RGB_ALPHA, RGBA8) files.
 *
 *   fixca_cli in.ppm out.ppm blue red [lens_x lens_y [interp [x_blue x_red y_blue y_red]]]
 *
 * Arguments are marshalled into PDB-shaped values and go through
 * fixca_unmarshal_noninteractive(), so omitted arguments take run()'s defaults.
 * Pass --lens-as-float to give lens_x/lens_y directly instead of through the
 * reference's d_int32 read; --pinned to page-lock the two images (pays off only for a
 * host that lives for many calls: a PDB call is one process, one image).
 * --fast-exit: leave with _exit(0) once the result file is written, flushed and closed: the HIP
 * runtime's exit handlers (60-90 ms of teardown after the result is already delivered) do not run
 * and the kernel driver reclaims the process's GPU resources as it does for any process that ends.
 * Opt-in, and only here: a library never ends its host's process, and inside GIMP gimp_quit()
 * owns the plug-in's exit.
 * FIXCA_PHASES=1 prints the call's phases with time stamps on stderr (fixca_phase).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fixca_host.h"

static int read_token(FILE *f, char *buf, int n)
{
	int c, i = 0;
	for (;;) {
		c = fgetc(f);
		if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
		if (c == EOF) return 0;
		if (c != ' ' && c != '\n' && c != '\r' && c != '\t') break;
	}
	while (c != EOF && c != ' ' && c != '\n' && c != '\r' && c != '\t') {
		if (i < n - 1) buf[i++] = (char)c;
		c = fgetc(f);
	}
	buf[i] = 0;
	return 1;
}

static int pixels_pinned;

static void release_pixels(unsigned char *px)
{
	if (pixels_pinned) fixca_host_free(px); else free(px);
}

static unsigned char *read_image(const char *path, int *w, int *h, int *bpp)
{
	char tok[64];
	FILE *f = fopen(path, "rb");
	unsigned char *px = NULL;
	if (!f) return NULL;
	if (!read_token(f, tok, sizeof tok)) goto done;
	if (!strcmp(tok, "P6")) {
		int maxv;
		if (!read_token(f, tok, sizeof tok)) goto done;
		*w = atoi(tok);
		if (!read_token(f, tok, sizeof tok)) goto done;
		*h = atoi(tok);
		if (!read_token(f, tok, sizeof tok)) goto done;
		maxv = atoi(tok);
		if (maxv != 255) goto done;
		*bpp = 3;
	} else if (!strcmp(tok, "P7")) {
		int depth = 0, maxv = 0;
		*w = *h = 0;
		while (read_token(f, tok, sizeof tok) && strcmp(tok, "ENDHDR")) {
			char val[64];
			if (!read_token(f, val, sizeof val)) goto done;
			if (!strcmp(tok, "WIDTH")) *w = atoi(val);
			else if (!strcmp(tok, "HEIGHT")) *h = atoi(val);
			else if (!strcmp(tok, "DEPTH")) depth = atoi(val);
			else if (!strcmp(tok, "MAXVAL")) maxv = atoi(val);
		}
		if (depth != 4 || maxv != 255) goto done;
		*bpp = 4;
	} else goto done;
	if (*w <= 0 || *h <= 0) goto done;
	px = fixca_host_use_pinned ? fixca_host_alloc((size_t)*w * *h * *bpp) : NULL;      /* page-locked: see fixca_host.c */
	pixels_pinned = px != NULL;
	if (!px) px = malloc((size_t)*w * *h * *bpp);       /* g_new at fix-ca.c:366 */
	if (px && fread(px, (size_t)*w * *bpp, *h, f) != (size_t)*h) { release_pixels(px); px = NULL; }
done:
	fclose(f);
	return px;
}

static int write_image(const char *path, const unsigned char *px, int w, int h, int bpp)
{
	FILE *f = fopen(path, "wb");
	if (!f) return -1;
	if (bpp == 3) fprintf(f, "P6\n%d %d\n255\n", w, h);
	else fprintf(f, "P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n", w, h);
	int ok = fwrite(px, (size_t)w * bpp, h, f) == (size_t)h;
	fclose(f);
	return ok ? 0 : -1;
}

int main(int argc, char **argv)
{
	FixcaPdbArg a[12];
	FixCaParams p;
	FixcaDrawable d;
	int lens_float = 0, fast_exit = 0, n = 0, i;
	const char *pos[16];

	fixca_phase("cli:main_entered");           /* after the dynamic loader: libfixca_hip.so, the HIP runtime and their constructors */
	fixca_prewarm();                           /* what a plug-in does first thing in run(): the runtime initialises while the image is read */

	for (i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--lens-as-float")) lens_float = 1;
		else if (!strcmp(argv[i], "--pinned")) fixca_host_use_pinned = 1;
		else if (!strcmp(argv[i], "--fast-exit")) fast_exit = 1;
		else if (n < 16) pos[n++] = argv[i];
	}
	if (n < 4 || n > 11) {
		fprintf(stderr, "usage: %s [--lens-as-float] [--pinned] [--fast-exit] in.ppm out.ppm blue red [lens_x lens_y [interp [x_blue x_red y_blue y_red]]]\n", argv[0]);
		return 2;
	}
	memset(a, 0, sizeof a);
	for (i = 2; i < n; ++i) {                 /* PDB args 3.. : blue red lens_x lens_y interp xb xr yb yr */
		int k = i + 1;
		if (k == 7) a[k].data.d_int8 = (int8_t)atoi(pos[i]);
		else a[k].data.d_float = atof(pos[i]);
	}
	if (fixca_unmarshal_noninteractive(n + 1, a, &p) != FIXCA_PDB_SUCCESS) return 1;
	if (lens_float) {
		if (n > 4) p.lens_x = atof(pos[4]);
		if (n > 5) p.lens_y = atof(pos[5]);
	}
	memset(&d, 0, sizeof d);
	d.pixels = read_image(pos[0], &d.width, &d.height, &d.bpp);
	if (!d.pixels) { fprintf(stderr, "cannot read %s (binary P6 or P7/RGB_ALPHA, maxval 255)\n", pos[0]); return 1; }
	fixca_phase("cli:image_read");
	d.bpc = fixca_color_size(d.bpp == 4 ? "R'G'B'A u8" : "R'G'B' u8", d.bpp);
	d.sel_x = 0; d.sel_y = 0; d.sel_w = d.width; d.sel_h = d.height;
	if (fix_ca(&d, &p)) return 1;
	fixca_phase("cli:fix_ca_returned");
	if (write_image(pos[1], d.pixels, d.width, d.height, d.bpp)) { fprintf(stderr, "cannot write %s\n", pos[1]); return 1; }
	fixca_phase("cli:image_written");
	if (fast_exit) {
		fflush(NULL);
		_exit(0);
	}
	release_pixels(d.pixels);
	fixca_phase("cli:exiting");
	return 0;
}
