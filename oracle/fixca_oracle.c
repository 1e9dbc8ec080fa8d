/*
 * fixca_oracle.c — CPU restatement of the reference's lateral/directional
 * chromatic-aberration row loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (gimp-fix-ca_amd/) never links, loads or calls it.
 *
 * What it restates (all citations into /root/reference/fix-ca.c):
 *   fix_ca_region()            :998-1348   the per-row loop, three modes
 *   scale() / scale_d()        :799-820    dest -> source coordinate
 *   round_nearest()            :776-789    half-away-from-zero, saturating
 *   get_pixel() / set_pixel()  :713-774    sample <-> double in [0,1]
 *   clip_d()                   :873-880
 *   bilinear()                 :882-894
 *   cubicY() / cubicX()        :896-920    Catmull-Rom, columns then rows
 *   centerline()               :945-996    preview cross-hair (show_progress=0)
 *   preview_update()           :659-671    crop of the visible window + 8-bit down-conversion
 *   load_data()/set_data()     :822-871    a 120-row cache that is
 *                                          semantically row(y) = src[y]; not
 *                                          reproduced, only its meaning.
 * It is written from the arithmetic description in SURVEY.md Appendix A; the
 * expression ORDER of every floating-point formula is kept because bit-exact
 * agreement depends on it.  Compile with -ffp-contract=off and no -march that
 * would license FMA contraction.
 *
 * PARITY UNPINNED (see oracle/README.md): the reference cannot be compiled in
 * this image (it needs <libgimp/gimp.h>, GEGL, babl, GTK; writing stand-ins for
 * them is not allowed), so oracle/_ref is not built, and the reference's single
 * own fixture, tests/test1.md5, needs GIMP's JPEG loader and BMP writer.  What
 * the oracle is checked against is the table of md5 vectors SURVEY.md Appendix
 * C / §8(d) records (60 + 7 cases, tests/test_oracle_golden.py): a consistency
 * check, not a reference-held pin.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fixca_types.h"

#define ORACLE_API __attribute__((visibility("default")))

/* ---- sample codecs (fix-ca.c:713-774) ---------------------------------- */

static inline double sample_get(const unsigned char *p, int bpc)
{
	double v = 0.0;
	if (bpc == 1) {                 /* u8 first: the common case stays a */
		v += *p;                    /* predictable branch, not a jump table */
		v /= 255;
	} else if (bpc == 2) {
		uint16_t s; memcpy(&s, p, 2); v += s; v /= 65535;
	} else if (bpc == 4) {
		uint32_t s; memcpy(&s, p, 4); v += s; v /= 4294967295;
	} else if (bpc == 8) {
		uint64_t s; long double l = 0.0; memcpy(&s, p, 8);
		l += s; l /= 18446744073709551615UL; v = l;
	} else if (bpc == -8) {
		double s; memcpy(&s, p, 8); v += s;
	} else if (bpc == -4) {
		float s; memcpy(&s, p, 4); v += s;
	}
	return v;
}

static inline void sample_put(unsigned char *p, double d, int bpc)
{
	if (bpc == 1) {
		*p = round(d * 255);
	} else if (bpc == 2) {
		uint16_t s = round(d * 65535); memcpy(p, &s, 2);
	} else if (bpc == 4) {
		uint32_t s = round(d * 4294967295); memcpy(p, &s, 4);
	} else if (bpc == 8) {
		uint64_t s = roundl(d * 18446744073709551615UL); memcpy(p, &s, 8);
	} else if (bpc == -8) {
		memcpy(p, &d, 8);
	} else if (bpc == -4) {
		float s = (float)d; memcpy(p, &s, 4);
	}
}

static double clip01(double d)              /* fix-ca.c:873-880 */
{
	if (d <= 0.0) return 0.0;
	if (d >= 1.0) return 1.0;
	return d;
}

/* ---- coordinate maps (fix-ca.c:776-820) -------------------------------- */

static int iabs(int i) { return i >= 0 ? i : -i; }

static int rnd_half_away(double d)          /* fix-ca.c:776-789 */
{
	if (d >= 0) {
		if (d > INT_MAX) return INT_MAX;
		return (int)(d + 0.5);
	}
	if (d < INT_MIN) return INT_MIN;
	return -((int)(0.5 - d));
}

/* the one coordinate expression, evaluated left to right in double */
static double src_coord(int i, int centre, double s, double shift)
{
	return (i - centre) * s + centre - shift;
}

static int map_int(int i, int centre, int size, double s, double shift)
{
	int j = rnd_half_away(src_coord(i, centre, s, shift));
	if (j <= 0) return 0;
	if (j >= size) return size - 1;
	return j;
}

static double map_dbl(int i, int centre, int size, double s, double shift)
{
	double d = src_coord(i, centre, s, shift);
	if (d <= 0.0) return 0.0;
	if (d >= size - 1) return size - 1;
	return d;
}

/* ---- interpolators ------------------------------------------------------ */

/* Catmull-Rom through four equally spaced values at parameter t
 * (fix-ca.c:905-907 and :916-918 are this same polynomial). */
static double catmull_rom(double a, double b, double c, double d, double t)
{
	return ((((-a + 3 * b - 3 * c + d) * t +
	          (2 * a - 5 * b + 4 * c - d)) * t +
	         (-a + c)) * t + (b + b)) / 2.0;
}

static double cr_row(const unsigned char *row, int bpp, int bpc, double t,
                     int cm, int c0, int c1, int c2)
{
	double a = sample_get(row + (size_t)cm * bpp, bpc);
	double b = sample_get(row + (size_t)c0 * bpp, bpc);
	double c = sample_get(row + (size_t)c1 * bpp, bpc);
	double d = sample_get(row + (size_t)c2 * bpp, bpc);
	return catmull_rom(a, b, c, d, t);
}

static void lerp2(unsigned char *out, const unsigned char *r0,
                  const unsigned char *r1, int c0, int c1,
                  int bpp, int bpc, double fx, double fy)
{
	double p00 = sample_get(r0 + (size_t)c0 * bpp, bpc);
	double p10 = sample_get(r0 + (size_t)c1 * bpp, bpc);
	double p01 = sample_get(r1 + (size_t)c0 * bpp, bpc);
	double p11 = sample_get(r1 + (size_t)c1 * bpp, bpc);
	double v = (1 - fy) * (p00 + fx * (p10 - p00))
	         + fy * (p01 + fx * (p11 - p01));       /* fix-ca.c:891-892 */
	sample_put(out, clip01(v), bpc);
}

/* ---- preview cross-hair (fix-ca.c:945-996), show_progress == 0 only ----- */

static void put_rgb(unsigned char *g, int b, int bpc, double c)
{
	sample_put(g - b, c, bpc);
	sample_put(g, c, bpc);
	sample_put(g + b, c, bpc);
}

static void crosshair(unsigned char *line, int width, int bpp, int bpc,
                      int x, int y, int xc, int yc)
{
	int i, b = iabs(bpc);
	double c = 1.0;
	unsigned char *g = line + b;            /* green of pixel 0 */

	if (y == yc) {
		i = iabs(xc - x) % 16;
		if (i < 8) c = 0.0;
		while (width-- > 0) {
			put_rgb(g, b, bpc, c);
			if (i-- < 0) {
				i = 7;
				c = (c > 0) ? 0.0 : 1.0;
			}
			g += bpp;
		}
		return;
	}
	y = (y <= yc) ? yc - y : y - yc;
	i = iabs(y) % 16;
	if (i < 8) c = 0.0;
	xc -= x;
	if (xc >= 0 && xc < width) put_rgb(g + (size_t)xc * bpp, b, bpc, c);
	x = xc - y;
	if (x >= 0 && x < width)   put_rgb(g + (size_t)x * bpp, b, bpc, c);
	x = xc + y;
	if (x >= 0 && x < width)   put_rgb(g + (size_t)x * bpp, b, bpc, c);
}

/* ---- the region loop ---------------------------------------------------- */

struct chan {              /* one resampled channel */
	int    off;            /* byte offset inside a pixel */
	double s;              /* scale about the lens centre */
	double xs, ys;         /* directional shifts */
};

/*
 * Same argument list and meaning as the reference's fix_ca_region
 * (fix-ca.c:998-1001).  Returns 0, or a negative code where the reference's
 * behaviour is undefined and the oracle refuses:
 *   -1 x1 != 0 or x2 != orig_width (the reference indexes
 *      wrongly there), -2 bad bpc/bytes, -3 max_dim + amount <= 0,
 *   -4 preview saturation requested (libgimpcolor arithmetic, unpinned).
 */
ORACLE_API int
fixca_oracle_region(const unsigned char *src, unsigned char *dst,
                    int orig_width, int orig_height, int bytes, int bpc,
                    const FixCaParams *p, int x1, int x2, int y1, int y2,
                    int show_progress)
{
	const int W = orig_width, H = orig_height;
	const int b = iabs(bpc);
	const size_t pitch = (size_t)W * bytes;
	struct chan ch[2];
	int xc, yc, max_dim, x, y, k;

	if (x1 != 0 || x2 != W) return -1;
	if (!(bpc == 1 || bpc == 2 || bpc == 4 || bpc == 8 || bpc == -4 || bpc == -8))
		return -2;
	if (bytes < 3 * b || W <= 0 || H <= 0 || y1 < 0 || y2 > H) return -2;
	if (!show_progress && p->saturation != 0.0) return -4;

	xc = (int)p->lens_x;                         /* fix-ca.c:1033-1034 */
	yc = (int)p->lens_y;
	max_dim = (xc >= yc) ? xc : yc;              /* fix-ca.c:1036-1043 */
	if (W - xc > max_dim) max_dim = W - xc;
	if (H - yc > max_dim) max_dim = H - yc;
	if (max_dim + p->blue <= 0 || max_dim + p->red <= 0) return -3;

	ch[0].off = 2 * b; ch[0].s = max_dim / (max_dim + p->blue);   /* :1044 */
	ch[0].xs = p->x_blue; ch[0].ys = p->y_blue;
	ch[1].off = 0;     ch[1].s = max_dim / (max_dim + p->red);    /* :1045 */
	ch[1].xs = p->x_red;  ch[1].ys = p->y_red;

	for (y = y1; y < y2; ++y) {
		unsigned char *out = dst + (size_t)y * pitch;

		memcpy(out, src + (size_t)y * pitch, pitch);   /* G, A  (:1098) */

		for (k = 0; k < 2; ++k) {
			const struct chan *c = &ch[k];

			if (p->interpolation == FIXCA_INTERP_NONE) {
				int sy = map_int(y, yc, H, c->s, c->ys);
				const unsigned char *row = src + (size_t)sy * pitch + c->off;
				for (x = 0; x < W; ++x) {
					int sx = map_int(x, xc, W, c->s, c->xs);
					memcpy(out + (size_t)x * bytes + c->off,
					       row + (size_t)sx * bytes, b);
				}
			} else if (p->interpolation == FIXCA_INTERP_LINEAR) {
				double yd = map_dbl(y, yc, H, c->s, c->ys);
				int r0 = floor(yd);
				double fy = yd - r0;
				int r1 = (r0 == H - 1) ? r0 : r0 + 1;
				const unsigned char *row0 = src + (size_t)r0 * pitch + c->off;
				const unsigned char *row1 = src + (size_t)r1 * pitch + c->off;
				for (x = 0; x < W; ++x) {
					double xd = map_dbl(x, xc, W, c->s, c->xs);
					int c0 = floor(xd);
					double fx = xd - c0;
					int c1 = (c0 == W - 1) ? c0 : c0 + 1;
					lerp2(out + (size_t)x * bytes + c->off, row0, row1,
					      c0, c1, bytes, bpc, fx, fy);
				}
			} else {
				double yd = map_dbl(y, yc, H, c->s, c->ys);
				int r0 = floor(yd);
				double fy = yd - r0;
				int rm = (r0 == 0) ? r0 : r0 - 1;
				int r1 = (r0 == H - 1) ? r0 : r0 + 1;
				int r2 = (r1 == H - 1) ? r1 : r1 + 1;
				const unsigned char *rowm = src + (size_t)rm * pitch + c->off;
				const unsigned char *row0 = src + (size_t)r0 * pitch + c->off;
				const unsigned char *row1 = src + (size_t)r1 * pitch + c->off;
				const unsigned char *row2 = src + (size_t)r2 * pitch + c->off;
				for (x = 0; x < W; ++x) {
					double xd = map_dbl(x, xc, W, c->s, c->xs);
					int c0 = floor(xd);
					double fx = xd - c0;
					int cm = (c0 == 0) ? c0 : c0 - 1;
					int c1 = (c0 == W - 1) ? c0 : c0 + 1;
					int c2 = (c1 == W - 1) ? c1 : c1 + 1;
					double h0 = cr_row(rowm, bytes, bpc, fx, cm, c0, c1, c2);
					double h1 = cr_row(row0, bytes, bpc, fx, cm, c0, c1, c2);
					double h2 = cr_row(row1, bytes, bpc, fx, cm, c0, c1, c2);
					double h3 = cr_row(row2, bytes, bpc, fx, cm, c0, c1, c2);
					double v = catmull_rom(h0, h1, h2, h3, fy);
					sample_put(out + (size_t)x * bytes + c->off, clip01(v), bpc);
				}
			}
		}

		if (!show_progress)                               /* :1322-1327 */
			crosshair(out, W, bytes, bpc, 0, y, xc, yc);
	}
	return 0;
}

/*
 * Row-parallel variant of the same function for the "all host cores" context
 * number.  Rows are independent, so each thread runs the scalar
 * routine on a band.  Built only when compiled with -fopenmp.
 */
ORACLE_API int
fixca_oracle_region_mt(const unsigned char *src, unsigned char *dst,
                       int orig_width, int orig_height, int bytes, int bpc,
                       const FixCaParams *p, int x1, int x2, int y1, int y2,
                       int show_progress, int nthreads)
{
	int rc = 0;
	if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
	for (int t = 0; t < nthreads; ++t) {
		long rows = (long)y2 - y1;
		int a = y1 + (int)(rows * t / nthreads);
		int e = y1 + (int)(rows * (t + 1) / nthreads);
		int r = (a < e) ? fixca_oracle_region(src, dst, orig_width, orig_height,
		                                      bytes, bpc, p, x1, x2, a, e,
		                                      show_progress) : 0;
		if (r) {
#ifdef _OPENMP
#pragma omp critical
#endif
			rc = r;
		}
	}
	return rc;
}

/*
 * The tail of preview_update() (fix-ca.c:659-671): the visible window (x, y, width, height) of
 * the corrected image `img` (orig_width pixels wide, what fix_ca_region wrote with
 * show_progress = 0) is copied into `prev`; samples wider than 8 bits go through
 * get_pixel -> set_pixel(.., 1), i.e. prev = (guchar) round(get_pixel(sample) * 255).
 * prev holds width * height * (bytes / |bpc|) bytes when |bpc| != 1, width * height * bytes otherwise.
 *
 * set_pixel stores the double into a guchar (fix-ca.c:752): for values outside 0..255 (float
 * images may hold any value in green/alpha, and in red/blue with interpolation None) C leaves the
 * conversion undefined; gcc on x86-64, the reference's platform, truncates to a 32-bit integer
 * (cvttsd2si: 0x80000000 when out of range or NaN) and keeps the low byte.  That is spelled out
 * here so that the oracle does not depend on its own compiler.
 */
static unsigned char u8_as_x86(double r)
{
	int32_t i = (r >= -2147483648.0 && r < 2147483648.0) ? (int32_t)r : INT32_MIN;
	return (unsigned char)((uint32_t)i & 0xffu);
}

ORACLE_API int
fixca_oracle_preview(const unsigned char *img, unsigned char *prev, int orig_width, int orig_height,
                     int bytes, int bpc, int x, int y, int width, int height)
{
	const int b = iabs(bpc);
	if (x < 0 || y < 0 || width < 0 || height < 0 || x + width > orig_width || y + height > orig_height) return -1;
	if (!(bpc == 1 || bpc == 2 || bpc == 4 || bpc == -4 || bpc == -8) || bytes < 3 * b) return -2;
	for (int i = 0; i < height; ++i) {
		const unsigned char *row = img + ((size_t)orig_width * (y + i) + x) * bytes;
		if (b == 1) {
			memcpy(prev + (size_t)width * i * bytes, row, (size_t)width * bytes);
		} else {
			const int n = width * bytes / b;
			for (int j = 0; j < n; ++j)
				prev[(size_t)width * i * bytes / b + j] = u8_as_x86(round(sample_get(row + (size_t)j * b, bpc) * 255));
		}
	}
	return 0;
}

ORACLE_API int fixca_oracle_has_openmp(void)
{
#ifdef _OPENMP
	return 1;
#else
	return 0;
#endif
}
