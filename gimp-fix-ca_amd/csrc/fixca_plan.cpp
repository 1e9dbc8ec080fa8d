/*
 * fixca_plan.cpp — host-side argument validation, geometry and launch planning.
 * No HIP calls here: this file also runs on a machine without a GPU (the
 * planner is exported through fixca_plan_describe and unit-tested on CPU).
 * One exception, through fixca_wide.hip: LINEAR / CUBIC on the wide sample types ask the
 * runtime for the current device and its occupancy (fixca_wide_device, fixca_wide_slots);
 * without a device they plan for a 256-CU part.
 *
 * Compiled with -ffp-contract=off: the coordinate arithmetic below must round
 * exactly like the kernels' (and the reference's) un-fused double expression.
 */
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fixca_hip.h"
#include "fixca_geom.h"
#include "fixca_internal.h"
#include "fixca_roll.h"
#include "fixca_wide.h"
#include "fixca_move.h"

static double h_coord(int i, int centre, double s, double shift)
{
	return (i - centre) * s + centre - shift;            /* fix-ca.c:801, :813 */
}

static double h_map_dbl(int i, int centre, int size, double s, double shift)
{
	double d = h_coord(i, centre, s, shift);
	if (d <= 0.0) return 0.0;
	if (d >= size - 1) return size - 1;
	return d;
}

static int h_rnd(double d)                                /* fix-ca.c:776-789 */
{
	if (d >= 0) {
		if (d > INT_MAX) return INT_MAX;
		return (int)(d + 0.5);
	}
	if (d < INT_MIN) return INT_MIN;
	return -((int)(0.5 - d));
}

static int h_map_int(int i, int centre, int size, double s, double shift)
{
	int j = h_rnd(h_coord(i, centre, s, shift));
	if (j <= 0) return 0;
	if (j >= size) return size - 1;
	return j;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int trunc_to_int(double d)
{
	if (!(d > (double)INT_MIN)) return INT_MIN + 1;       /* also NaN */
	if (d >= (double)INT_MAX) return INT_MAX;
	return (int)d;
}

extern "C" int fixca_params_check(const FixCaParams *p)
{
	/* run()'s non-interactive range check, fix-ca.c:279-295 */
	if (!p) return FIXCA_E_INVALID;
	const double m = FIXCA_INPUT_MAX;
	const double v[6] = { p->blue, p->red, p->x_blue, p->x_red, p->y_blue, p->y_red };
	for (int i = 0; i < 6; ++i)
		if (!(v[i] >= -m && v[i] <= m)) return FIXCA_E_DOMAIN;
	if (p->interpolation < 0 || p->interpolation > 2) return FIXCA_E_DOMAIN;
	return FIXCA_OK;
}

int fixca_make_geom(int W, int H, int bytes, int bpc, const FixCaParams *p,
                    int x1, int x2, int y1, int y2, int show_progress, FixcaGeom *g)
{
	if (!p || !g) return fixca_fail(FIXCA_E_INVALID, "null parameter block");
	if (W <= 0 || H <= 0) return fixca_fail(FIXCA_E_INVALID, "image size %dx%d", W, H);
	if (y1 < 0 || y2 > H || y1 > y2)
		return fixca_fail(FIXCA_E_INVALID, "row range [%d,%d) outside 0..%d", y1, y2, H);
	if (x1 != 0 || x2 != W)
		return fixca_fail(FIXCA_E_UNSUPPORTED,
		                  "column range [%d,%d) != [0,%d): the reference indexes partial columns wrongly "
		                  "(fix-ca.c:855,1084,1098); only full-width calls are defined", x1, x2, W);
	/* color_size() codes (fix-ca.c:681-711): 1, 2, 4 unsigned integers, -4 float, -8 double.
	 * 8 (u64) goes through long double in the reference (fix-ca.c:727-731, :763-766): no GPU
	 * equivalent of x87 extended precision, so it is refused. */
	const int sb = bpc < 0 ? -bpc : bpc;
	if (!(bpc == 1 || bpc == 2 || bpc == 4 || bpc == -4 || bpc == -8) || (bytes != 3 * sb && bytes != 4 * sb))
		return fixca_fail(FIXCA_E_UNSUPPORTED, "bpc=%d bytes=%d: implemented are RGB/RGBA with u8, u16, u32, float or double samples",
		                  bpc, bytes);
	/* show_progress == 0 is preview_update()'s call (fix-ca.c:656): after the row loop the
	 * reference boosts saturation (only if saturation != 0; that uses libgimpcolor's HSV
	 * conversion, which is not in the reference tree: refused) and draws the lens-centre lines. */
	if (!show_progress && p->saturation != 0.0)
		return fixca_fail(FIXCA_E_UNSUPPORTED, "preview saturation boost (saturation=%g) is not implemented", p->saturation);
	if (p->interpolation < 0 || p->interpolation > 2)
		return fixca_fail(FIXCA_E_DOMAIN, "interpolation %d not in 0..2", p->interpolation);

	memset(g, 0, sizeof *g);
	g->W = W; g->H = H; g->bpp = bytes;
	g->bpc = bpc; g->nch = bytes / sb;
	g->preview = show_progress ? 0 : 1;
	g->xc = trunc_to_int(p->lens_x);                      /* fix-ca.c:1033-1034 */
	g->yc = trunc_to_int(p->lens_y);
	g->y1 = y1; g->y2 = y2;
	g->interp = p->interpolation;

	long long max_dim = (g->xc >= g->yc) ? g->xc : g->yc;  /* fix-ca.c:1036-1043 */
	if ((long long)W - g->xc > max_dim) max_dim = (long long)W - g->xc;
	if ((long long)H - g->yc > max_dim) max_dim = (long long)H - g->yc;
	if (max_dim > INT_MAX) max_dim = INT_MAX;
	const int md = (int)max_dim;
	if (!(md + p->red > 0) || !(md + p->blue > 0))
		return fixca_fail(FIXCA_E_DOMAIN, "max_dim %d + amount <= 0 (red %g, blue %g): the reference divides by <= 0",
		                  md, p->red, p->blue);
	g->s[0] = md / (md + p->red);                         /* fix-ca.c:1045 */
	g->s[1] = md / (md + p->blue);                        /* fix-ca.c:1044 */
	g->xs[0] = p->x_red;  g->ys[0] = p->y_red;
	g->xs[1] = p->x_blue; g->ys[1] = p->y_blue;
	for (int c = 0; c < 2; ++c)
		if (!isfinite(g->s[c]) || !isfinite(g->xs[c]) || !isfinite(g->ys[c]) || !(g->s[c] > 0))
			return fixca_fail(FIXCA_E_DOMAIN, "non-finite or non-positive scale/shift");
	return FIXCA_OK;
}

static bool debug_on(void)
{
	const char *d = getenv("FIXCA_DEBUG");
	return d && *d && strcmp(d, "0") != 0;
}

int fixca_debug_env(const char *name, int dflt, int lo, int hi)
{
	if (!debug_on()) return dflt;
	const char *e = getenv(name);
	if (!e || !*e) return dflt;
	int v = atoi(e);
	if (v < lo) v = lo;
	if (v > hi) v = hi;
	return v;
}

bool fixca_debug_flag(const char *name)
{
	if (!debug_on()) return false;
	const char *e = getenv(name);
	return e && *e && strcmp(e, "0") != 0;
}

static int env_int(const char *name, int dflt, int lo, int hi) { return fixca_debug_env(name, dflt, lo, hi); }

/* FIXCA_DEBUG=1 FIXCA_PLAN_TRACE=1: why a planner refused a geometry, on stderr */
#define PLAN_REFUSE(...) do { if (fixca_debug_flag("FIXCA_PLAN_TRACE")) { fprintf(stderr, "fixca-plan: " __VA_ARGS__); fputc('\n', stderr); } return false; } while (0)

/* does every strip's source window fit the 256 columns a wave filters? */
static bool interp_fits(const FixcaGeom *g, int oc, int taps)
{
	const int back = (taps == 4) ? 1 : 0;
	for (int c = 0; c < 2; ++c)
		for (int X0 = 0; X0 < g->W; X0 += oc) {
			int xl = X0 + oc - 1;
			if (xl > g->W - 1) xl = g->W - 1;
			int S0 = (int)floor(h_map_dbl(X0, g->xc, g->W, g->s[c], g->xs[c])) - back;
			int last = (int)floor(h_map_dbl(xl, g->xc, g->W, g->s[c], g->xs[c])) - back - S0 + taps - 1;
			if (last > FIXCA_WAVE_COLS - 1) return false;
		}
	return true;
}


/* ---- plan for the LDS-ring scan kernel (fixca_scan.hip) -------------------------- */

/* does every strip's union window (identity columns + red taps + blue taps) fit 256 columns? */
static bool scan_cols_fit(const FixcaGeom *g, int oc, int taps)
{
	const int back = (taps == 4) ? 1 : 0, W = g->W;
	const int nstrips = (W + oc - 1) / oc;
	for (int st = 0; st < nstrips; ++st) {
		int X0 = st * oc;
		if (X0 > W - oc) X0 = W - oc;
		int lo = X0, hi = X0 + oc - 1;
		for (int c = 0; c < 2; ++c) {
			int a, b;
			if (taps == 0) {
				a = h_map_int(X0, g->xc, W, g->s[c], g->xs[c]);
				b = h_map_int(X0 + oc - 1, g->xc, W, g->s[c], g->xs[c]);
			} else {
				a = clampi((int)floor(h_map_dbl(X0, g->xc, W, g->s[c], g->xs[c])) - back, 0, W - 1);
				b = clampi((int)floor(h_map_dbl(X0 + oc - 1, g->xc, W, g->s[c], g->xs[c])) - back + taps - 1, 0, W - 1);
			}
			if (a < lo) lo = a;
			if (b > hi) hi = b;
		}
		int U0 = lo & ~3;
		if (U0 > W - FIXCA_WAVE_COLS) U0 = W - FIXCA_WAVE_COLS;
		if (hi > U0 + FIXCA_WAVE_COLS - 1) return false;
	}
	return true;
}

/* most source rows any output row keeps alive at once (identity row + both windows) */
static int scan_row_span(const FixcaGeom *g, int taps)
{
	const int back = (taps == 4) ? 1 : 0, H = g->H;
	int span = 1;
	for (int y = g->y1; y < g->y2; ++y) {
		int keep = y, need = y;
		for (int c = 0; c < 2; ++c) {
			int a, b;
			if (taps == 0) {
				a = b = h_map_int(y, g->yc, H, g->s[c], g->ys[c]);
			} else {
				int y0 = (int)floor(h_map_dbl(y, g->yc, H, g->s[c], g->ys[c]));
				a = clampi(y0 - back, 0, H - 1);
				b = clampi(y0 - back + taps - 1, 0, H - 1);
			}
			if (a < keep) keep = a;
			if (b > need) need = b;
		}
		if (need - keep + 1 > span) span = need - keep + 1;
	}
	return span;
}

static bool plan_scan(const FixcaGeom *g, FixcaPlan *pl)
{
	if (g->bpp != 4 || (g->W & 3) || g->W < FIXCA_WAVE_COLS) return false;
	if ((long long)g->W * g->H * 4 >= (1LL << 40)) return false;
	const int taps = (g->interp == 0) ? 0 : (g->interp == 1 ? 2 : 4);
	const int span = scan_row_span(g, taps);
	/* a block of 4 output rows keeps span + 3 rows alive; leave >= 4 slots for rows in flight */
	int ring = env_int("FIXCA_RING", 0, 0, 64);
	if (ring == 0) ring = (span + 7 <= 8) ? 8 : (span + 7 <= 16) ? 16 : 32;
	if ((ring & (ring - 1)) || span + 5 > ring) return false;
	int oc_max = env_int("FIXCA_OC", 252, 64, 252) & ~3;
	for (int oc = oc_max; oc >= 128; oc -= 4) {
		if (!scan_cols_fit(g, oc, taps)) continue;
		pl->kernel = FIXCA_K_SCAN;
		pl->oc = oc;
		pl->th = env_int("FIXCA_TH", 64, 1, 4096);
		pl->ring = ring;
		int rows = g->y2 - g->y1;
		pl->nstrips = (g->W + oc - 1) / oc;
		pl->nchunks = (rows + pl->th - 1) / pl->th;
		if (pl->nchunks < 1) pl->nchunks = 1;
		pl->nwaves = pl->nstrips * pl->nchunks;
		pl->nblocks = (pl->nwaves + 7) & ~7;
		return true;
	}
	return false;
}


/* ---- plan for the row-rolling kernel (fixca_roll.hip) ----------------------------- */

/* first and last tap column WITH WEIGHT of output column x, clamped as the reference clamps
 * them (fix-ca.c:1170-1177 linear, :1270-1298 cubic).  A coordinate exactly on a column (every
 * coordinate clamped to the image edge is) has all its weight on that column; k_roll takes the
 * same decision on the same fp64 fraction. */
static void tap_span(const FixcaGeom *g, int c, int x, int taps, int *first, int *last)
{
	const int W = g->W;
	if (taps == 1) {                                    /* NONE: the rounded source column, weight 1 */
		*first = *last = h_map_int(x, g->xc, W, g->s[c], g->xs[c]);
		return;
	}
	const double xd = h_map_dbl(x, g->xc, W, g->s[c], g->xs[c]);
	const int x0 = (int)floor(xd);
	if (xd - (double)x0 == 0.0) {
		*first = *last = x0;
		return;
	}
	if (taps == 4) {
		int c2 = x0 + 1 < W - 1 ? x0 + 1 : W - 1;
		*first = x0 - 1 > 0 ? x0 - 1 : 0;
		*last = c2 + 1 < W - 1 ? c2 + 1 : W - 1;
	} else {
		*first = x0;
		*last = x0 + 1 < W - 1 ? x0 + 1 : W - 1;
	}
}

/* Every strip must satisfy what k_roll assumes: with m = min over the strip of (tap column - x),
 * every tap with weight lands on pixel i + j of the lane's 8, 0 <= j <= 4; the lane windows
 * start at most 32 px before and end at most 32 px past the 256-column ring row (the pads: 128 bytes each);
 * every column that carries weight is inside the ring row.
 * Outputs that ARE the row's first or last pixel (a coordinate clamped to the image: all the weight on column 0 or
 * W - 1) do not count towards m and the 5-pixel reach: the ones the 5 pixels do not reach are rewritten at the end of
 * each tile (fixca_roll.hip, roll_edge_pixels; k_roll_tables takes the same decisions on the same fp64 values): they
 * must be the strip's first or last outputs, of one side only, at most 255 per channel (*edge: the plan has some). */
static int roll_win(const FixcaGeom *g) { return g->bpp == 3 ? ROLL_WIN_RGB : ROLL_WIN; }

static bool roll_cols_fit(const FixcaGeom *g, int oc, int taps, int *edge)
{
	*edge = 0;
	const int WIN = roll_win(g);
	const int back = (taps == 4) ? 1 : 0, W = g->W;
	const int nstrips = (W + oc - 1) / oc;
	for (int st = 0; st < nstrips; ++st) {
		const int X0 = st * oc;                         /* k_roll's last strip may be narrower */
		const int xe = X0 + oc < W ? X0 + oc : W;
		int lo = X0, hi = xe - 1, m[2], sides = 0;      /* sides: 1 flagged first-pixel outputs, 2 last-pixel */
		for (int c = 0; c < 2; ++c) {
			int a = taps == 1 ? h_map_int(X0, g->xc, W, g->s[c], g->xs[c])
			                  : clampi((int)floor(h_map_dbl(X0, g->xc, W, g->s[c], g->xs[c])) - back, 0, W - 1);
			if (a < lo) lo = a;
		}
		for (int c = 0; c < 2; ++c) {
			int mn = INT_MAX, mx = INT_MIN, mn_all = INT_MAX, mx_all = INT_MIN;
			for (int x = X0; x < xe; ++x) {
				int f, l;
				tap_span(g, c, x, taps, &f, &l);
				if (l > hi) hi = l;
				if (f - x < mn_all) mn_all = f - x;
				if (l - x > mx_all) mx_all = l - x;
				if (f < lo) PLAN_REFUSE("roll oc=%d strip %d channel %d: column %d taps %d, below the window start %d", oc, st, c, x, f, lo);               /* the window start assumes the first column needs the lowest tap */
				if (f == l && (f == 0 || f == W - 1)) continue;          /* an edge pixel */
				if (f - x < mn) mn = f - x;
				if (l - x > mx) mx = l - x;
			}
			if (mx_all - mn_all <= 4) { m[c] = mn_all; continue; }      /* the 5 pixels reach everything: edge pixels are outputs like any others */
			if (mn == INT_MAX) mn = mx = 0;                              /* a strip of edge pixels only */
			if (mx - mn > 4) PLAN_REFUSE("roll oc=%d strip %d channel %d: tap offsets span %d..%d (more than 5 lane pixels)", oc, st, c, mn, mx);
			int nfl = 0, fl_lo = INT_MAX, fl_hi = INT_MIN;
			for (int x = X0; x < xe; ++x) {
				int f, l;
				tap_span(g, c, x, taps, &f, &l);
				if (!(f == l && (f == 0 || f == W - 1))) continue;
				if (f - x - mn < 0 || f - x - mn > 4) {
					*edge = 1;
					sides |= (f == 0) ? 1 : 2;
					++nfl;
					if (x < fl_lo) fl_lo = x;
					if (x > fl_hi) fl_hi = x;
				}
			}
			if (sides == 3) PLAN_REFUSE("roll oc=%d strip %d: edge pixels of both sides", oc, st);    /* the tile's last step rewrites one side */
			if (nfl > 255 || (nfl && sides == 1 && (fl_lo != 0 || fl_hi != nfl - 1)) || (nfl && sides == 2 && (fl_hi != W - 1 || fl_lo != W - nfl)))
				PLAN_REFUSE("roll oc=%d strip %d channel %d: %d edge pixels in columns %d..%d are not the row's first / last", oc, st, c, nfl, fl_lo, fl_hi);
			m[c] = mn;
		}
		int U0 = lo & ~ROLL_U0_MASK;
		if (U0 > W - WIN) U0 = W - WIN;
		if (hi > U0 + WIN - 1) PLAN_REFUSE("roll oc=%d strip %d: columns %d..%d do not fit the %d-column window at %d", oc, st, lo, hi, WIN, U0);
		const int nact = (xe - X0 + 3) >> 2;            /* producing lanes (the last may be partial); lane `nact` is read as a neighbour */
		for (int c = 0; c < 2; ++c) {
			const int P = X0 + m[c] - U0;
			if (P < -32 || P + 4 * nact + 4 > WIN + 32) PLAN_REFUSE("roll oc=%d strip %d channel %d: lane windows start at %d (pads allow -32 .. %d)", oc, st, c, P, WIN + 32 - 4 * nact - 4);
		}
	}
	return true;
}

static bool plan_roll(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	/* any width from the window's up (since round 4; a width that is not a multiple of 4 ran on the register-window
	 * kernels at a third of the speed).  RGB8 rows of such an image start at any byte: gfx950's buffer loads, LDS-DMA and
	 * stores take byte addresses as they come (tools/skeletons/unaligned_probe.hip, profiles/r04_unaligned_probe.txt). */
	if ((g->bpp != 4 && g->bpp != 3) || g->W < roll_win(g)) return false;
	if (((long long)g->W * g->bpp) % 4 != 0 && !fixca_unaligned_ok()) return false;      /* rows at byte addresses: only where the device is known to move them faithfully */
	const int taps = (g->interp == 0) ? 1 : (g->interp == 1) ? 2 : 4;      /* NONE runs as a 1-tap filter */
	const int span = scan_row_span(g, g->interp == 0 ? 0 : taps);
	/* ring rows: the span + 3 rows a block of 4 keeps alive plus rows in flight.  16 when that
	 * fits (row mod 16 is one instruction); otherwise as small as 6 rows in flight allow, because
	 * every KB of ring is a wave less per CU (160 KB of LDS) */
	int ring = env_int("FIXCA_RING", 0, 0, 64);
	if (ring == 0) ring = (span + 7 <= 8) ? 8 : (span + 7 <= 16) ? 16 : span + 9;
	if (ring > 64 || span + 5 > ring) PLAN_REFUSE("roll: span %d needs a ring of %d rows (max 64)", span, ring);
	/* Strip width: a multiple of 16 pixels makes every row store start and end on a 64-byte
	 * boundary (no memory line is shared by two waves' stores; measured 6 % faster than the
	 * widest width that fits, 248); other widths only if no such width fits. */
	const int oc_top = ROLL_NEXT_FROM_LDS ? 256 : 252;
	const int oc_env = env_int("FIXCA_OC", 0, 0, oc_top) & ~3;
	/* the width search walks every column for every candidate width and does not depend on the
	 * row range: remembered per thread, so that a caller working through an image in row bands
	 * (fixca_region_host's pipeline, the preview) plans each band in microseconds */
	static thread_local struct { bool valid; FixcaGeom g; int oc_env; int oc_pick; int edge; } memo;
	FixcaGeom key = *g;
	key.y1 = key.y2 = 0;
	key.preview = 0;
	int oc_pick = 0, edge = 0;
	if (memo.valid && memo.oc_env == oc_env && memcmp(&memo.g, &key, sizeof key) == 0) {
		oc_pick = memo.oc_pick;
		edge = memo.edge;
	} else {
		if (oc_env) {
			for (int oc = oc_env; oc >= 128 && !oc_pick; oc -= 4)
				if (roll_cols_fit(g, oc, taps, &edge)) oc_pick = oc;
		} else {
			for (int oc = oc_top & ~15; oc >= 128 && !oc_pick; oc -= 16)
				if (roll_cols_fit(g, oc, taps, &edge)) oc_pick = oc;
			for (int oc = oc_top; oc >= 128 && !oc_pick; oc -= 4)
				if (roll_cols_fit(g, oc, taps, &edge)) oc_pick = oc;
		}
		memo.valid = true;
		memo.g = key;
		memo.oc_env = oc_env;
		memo.oc_pick = oc_pick;
		memo.edge = edge;
	}
	for (int oc = oc_pick; oc >= 128; oc = 0) {
		const int rows = g->y2 - g->y1;
		pl->kernel = FIXCA_K_ROLL;
		pl->edge = edge;
		pl->oc = oc;
		pl->ring = ring;
		pl->nstrips = (g->W + oc - 1) / oc;
		/* Rows per wave, chosen for a launch of ONE image (in a batch launch the waves of the
		 * next image fill the slots as they free up and th = 29..113 measure the same).  Every
		 * wave of the image should be resident at once (one round): a CU holds `per_cu` waves
		 * (LDS: ring + pads per wave, 160 KB per CU), and the waves spread evenly over the 256
		 * CUs.  A wave's time is (th + c) row times, c ~ 12 rows for its fixed cost and halo; a
		 * row time grows with the k = ceil(waves / 256) waves sharing a CU about as
		 * 0.27 + 0.05 k us (measured on MI355X, 6000x4000, k = 4..9). */
		int per_cu = (160 * 1024) / (ring * roll_win(g) * g->bpp + 512);
		if (per_cu > 7) per_cu = 7;                     /* one image alone: 7 waves per CU beat 8 and 9 at equal total rows (measured) */
		int best_th = 0;
		long best_cost = 0;
		for (int th = 16; th <= 1024; ++th) {
			const long chunks = (rows + th - 1) / th;
			const long per = (chunks * pl->nstrips + 255) / 256;
			if (per > per_cu && th < 1024) continue;
			const long cost = (27 + 5 * per) * (th + 12);
			if (!best_th || cost < best_cost) { best_th = th; best_cost = cost; }
			if (chunks == 1) break;
		}
		if (flags & FIXCA_PLAN_BATCH) {
			/* many images back to back: the GPU stays full whatever the wave count, and short waves measure best.  A tile
			 * is one rebuilt row plus blocks of 4, so heights are 4k + 1 (a leftover single row costs nearly a block).
			 * CUBIC: 4 x the rows a wave keeps alive + 5, so the halo stays ~20 % (6000x4000, span 7: th 33 -> 35.2 us per
			 * image, 44 -> 36.0, 57 -> 37.4; round 4, two boxes: 21 / 29 / 33 rows 72.4 / 72.5 / 72.5 %, 25: 71.8, 37+: 70.5).
			 * LINEAR and NONE want them shorter (round 4, profiles/r04_batch_tile_heights.txt, two boxes: LINEAR 17 / 21 /
			 * 25 / 33 rows 74.7 / 73.5-74.5 / 71.3-72.4 / 72.7-73.7 %, NONE 74.3 / 73.9-74.1 / 72.2-72.5 / 73.6). */
			best_th = (taps == 4) ? 4 * span + 5 : 4 * span + 1;
			/* RGB8 CUBIC: since the tap windows are picked apart by byte alignment (round 3) the kernel holds 181 registers, 8 waves
			 * per CU, and longer tiles win (tools/ab.py --bpp 3, one allocation: th 25 62.6-63.0 %, 33 64.2-65.1, 37 66.0, 41
			 * 65.8-66.3, 49 65.8, 57 64.6).  RGB8 LINEAR / NONE (165 / 157 registers) follow RGBA8: 17 / 21 / 25 / 29 rows
			 * 67.6 / 67.9 / 66.5-66.7 / 65.4-65.5 % LINEAR, NONE 67.9 / 68.3 / 66.6-67.0 / 65.6 %. */
			if (g->bpp == 3 && taps == 4) best_th = 6 * span - 1;
			if (best_th < 21) best_th = 21;
			if (best_th > 128) best_th = 128;
			/* 25-row tiles of a 6000x4000 image were the one height that measured 1-2 points under both neighbours in
			 * every mode: 160 chunks x 25 strips = 4000 tiles, 500 per XCD = a whole number of chunk rows, so the eight XCDs
			 * walk through the same strips in step, 500 image rows apart.  Such a height moves on by one block. */
			for (int tries = 0; tries < 4; ++tries) {
				const long chunks = (rows + best_th - 1) / best_th;
				const long per_xcd = ((chunks * pl->nstrips + 7) & ~7L) / 8;
				if (chunks < 16 || per_xcd % pl->nstrips != 0) break;
				best_th += 4;
			}
		}
		pl->th = env_int("FIXCA_TH", best_th, 1, 4096);
		/* k_roll addresses a wave's rows with 32-bit offsets into per-wave buffer descriptors */
		if ((long long)(pl->th + 2 * FIXCA_SOURCE_ROWS) * g->W * g->bpp > (1LL << 31)) return false;
		pl->nchunks = (rows + pl->th - 1) / pl->th;
		if (pl->nchunks < 1) pl->nchunks = 1;
		pl->nwaves = pl->nstrips * pl->nchunks;
		if ((long long)pl->nwaves + 8 > (1LL << 31) / ROLL_BATCH_MAX) return false;     /* grid size of a batch launch */
		pl->nblocks = (pl->nwaves + 7) & ~7;
		return true;
	}
	return false;
}

/* The tile schedule of one k_roll launch (RollSched, fixca_roll.h) */
void roll_make_sched(const FixcaGeom *g, const FixcaPlan *pl, int n_images, RollSched *sc)
{
	memset(sc, 0, sizeof *sc);
	const int rows = g->y2 - g->y1;
	sc->ncols = n_images * pl->nstrips;
	sc->rows = rows;
	sc->th = pl->th;
	sc->pool_chunks = (rows + pl->th - 1) / pl->th;
	if (sc->pool_chunks < 1) sc->pool_chunks = 1;
	sc->pool_waves = pl->nstrips * sc->pool_chunks;
	sc->pool_blocks = (sc->pool_waves + 7) & ~7;
	sc->edge_wb = env_int("FIXCA_EDGE_WB", n_images > 1 ? 1 : 0, 0, 1);
}

/* test hook (CPU): the schedule as numbers */
extern "C" __attribute__((visibility("default"))) int fixca_debug_sched(int W, int H, int bytes, const FixCaParams *p, int n_images, int *out6)
{
	FixcaGeom g;
	FixcaPlan pl;
	int rc = fixca_make_geom(W, H, bytes, 1, p, 0, W, 0, H, 1, &g);
	if (rc) return rc;
	rc = fixca_make_plan(&g, n_images > 1 ? FIXCA_PLAN_BATCH : 0u, &pl);
	if (rc) return rc;
	if (pl.kernel != FIXCA_K_ROLL) return fixca_fail(FIXCA_E_UNSUPPORTED, "not a k_roll geometry");
	RollSched sc;
	roll_make_sched(&g, &pl, n_images, &sc);
	const int v[6] = { sc.ncols, sc.rows, sc.pool_chunks, sc.pool_waves, sc.pool_blocks, sc.th };
	memcpy(out6, v, sizeof v);
	return FIXCA_OK;
}

static void finish_plan(const FixcaGeom *g, FixcaPlan *pl)
{
	int rows = g->y2 - g->y1;
	pl->nstrips = (g->W + pl->oc - 1) / pl->oc;
	pl->nchunks = (rows + pl->th - 1) / pl->th;
	if (pl->nchunks < 1) pl->nchunks = 1;
	pl->nwaves = pl->nstrips * pl->nchunks;
	pl->nblocks = ((pl->nwaves + 3) / 4 + 7) & ~7;
}

/* ---- plan for the wide-sample rolling kernel (fixca_wide.hip) ------------------------- */

/* k_wcol gives every channel its own 128-column window starting at the (even-rounded) first tap of
 * the strip's first column; every tap of every column of the strip must fall inside it */
/* 16-bit CUBIC: may k_wcol use the weight form (fixca_wide.hip)?  Its results are the reference's for every input; what
 * is decided here is only speed: a sample whose value is within 1e-6 of k + 1/2 is recomputed operation for operation,
 * and such samples are common exactly when the fractions of the source coordinates sit on a coarse dyadic grid in both
 * directions (pure shifts by halves, quarters or eighths: with fx = 1/2, fy = 0 every 16th sample is an exact k + 1/2).
 * Scales other than 1 spread the fractions over the columns and rows and such samples are a few per image. */
static bool wide_fast_ok(const FixcaGeom *g)
{
	if (g->bpc != 2 || g->interp == 0 || fixca_debug_flag("FIXCA_WCOL_REF")) return false;       /* 16-bit samples; CUBIC in k_wcol, LINEAR in k_lin */
	if (fixca_debug_flag("FIXCA_WCOL_WEIGHTS")) return true;          /* tests: the recomputation path under load */
	const int rows = g->y2 - g->y1;
	for (int c = 0; c < 2; ++c) {
		long coarse[2] = { 0, 0 }, zero[2] = { 0, 0 };
		for (int d = 0; d < 2; ++d) {
			const int lo = d ? g->y1 : 0, hi = d ? g->y2 : g->W;
			for (int i = lo; i < hi; ++i) {
				const double v = d ? h_map_dbl(i, g->yc, g->H, g->s[c], g->ys[c]) : h_map_dbl(i, g->xc, g->W, g->s[c], g->xs[c]);
				const double f8 = (v - floor(v)) * 8.0, dist = fabs(f8 - floor(f8 + 0.5));
				if (dist < 1e-4) { if (floor(f8 + 0.5) == 0.0 || floor(f8 + 0.5) == 8.0) ++zero[d]; else ++coarse[d]; }
			}
		}
		if ((coarse[0] * 8 > g->W && (coarse[1] + zero[1]) * 8 > rows) || (coarse[1] * 8 > rows && (coarse[0] + zero[0]) * 8 > g->W)) return false;
	}
	return true;
}

static bool plan_wide(const FixcaGeom *g, FixcaPlan *pl)
{
	/* RGB u16 of odd width: k_wcol moves its pixel pairs as dwords at halfword addresses (round 4) */
	if (((long long)g->W * g->bpp) % 4 != 0 && !fixca_unaligned_ok()) return false;
	if (g->bpc == 1 || g->W < 2) return false;
	/* NONE: the rolling kernel wins on 16-bit samples (6000x4000 RGBA u16 93 vs 129 us, RGB 83 vs 118); on 4- and
	 * 8-byte samples the one-thread-per-pixel gather is faster (RGBA float 151 vs 189 us, double 315 vs 348) */
	if (g->interp == 0 && g->bpc != 2) return false;
	/* the kernel addresses the image through 32-bit buffer offsets: rows below 8 MB, image below 4 GB */
	if ((long long)g->W * g->bpp >= (8LL << 20) || (long long)g->W * g->bpp * g->H >= (1LL << 32) - 4096) return false;
	/* the one case where the one-thread-per-pixel kernel measures faster: LINEAR on RGB double samples (no get_pixel
	 * division to save, and 48-byte pixel pairs stored by the lane that computed them; RGBA double moves as 16-byte
	 * units and is faster here: 328 vs 417 us for 6000x4000) */
	if (g->bpc == -8 && g->interp == 1 && g->nch != 4) return false;
	const int taps = (g->interp == 2) ? 4 : (g->interp == 1) ? 2 : 1, back = (taps == 4) ? 1 : 0, W = g->W;
	const int oc = WIDE_OC;
	for (int c = 0; c < 2; ++c)
		for (int X0 = 0; X0 < W; X0 += oc) {
			/* window origin as the kernel takes it: the strip's lowest tap, unclamped (-1 at the left edge), rounded
			 * down to even; the clamped taps of the reference are then consecutive window slots (fixca_wide.hip).
			 * NONE has one tap, the nearest column (scale(), fix-ca.c:799-809) */
			const int xl = (X0 + oc < W ? X0 + oc : W) - 1;
			int f, l;
			if (taps == 1) {
				f = h_map_int(X0, g->xc, W, g->s[c], g->xs[c]) & ~1;
				l = h_map_int(xl, g->xc, W, g->s[c], g->xs[c]);
			} else {
				f = ((int)floor(h_map_dbl(X0, g->xc, W, g->s[c], g->xs[c])) - back) & ~1;
				l = (int)floor(h_map_dbl(xl, g->xc, W, g->s[c], g->xs[c])) - back + taps - 1;
			}
			if (l - f > WIDE_WIN - 1) return false;
			/* the map is monotone for s > 0, so no column of the strip reaches below f or above l */
		}
	pl->kernel = FIXCA_K_WCOL;
	pl->oc = oc;
	pl->wide_fast = (g->interp == 2 && wide_fast_ok(g)) ? 1 : 0;      /* k_wcol has the weight form for CUBIC only */
	/* Tile height: all tiles resident at once, the same number on every SIMD.  The kernel is bound by fp64 issue and its
	 * waves all do the same work, so a launch lasts as long as its fullest SIMD (6000x4000 RGBA u16, 3 resident waves
	 * per SIMD: th 66 -> 3050 waves 132 us; th 60 -> 3350 waves 164 us; th 90 -> 2250 waves 170 us).  Several rounds
	 * when one would need tiles above WIDE_TH_MAX.  Tried and dropped: resident waves taking small tiles from an
	 * atomic queue (evens out the SIMDs, but a tile's start costs as much as 19 rows: 184-250 us). */
	{
		const int q = WIDE_SLOTS * (WIDE_SLOTS - 1), th_max = WIDE_TH_MAX / q * q, rows = g->y2 - g->y1, nstrips = (W + oc - 1) / oc;
		int slots = fixca_wide_slots(g, pl->wide_fast);
		if (slots <= 0) slots = 12 * 256;                  /* no device to ask (planning on a CPU-only machine) */
		const int per_round = slots / nstrips > 0 ? slots / nstrips : 1;
		int th = (rows + per_round - 1) / per_round;
		if (th > th_max) {
			const int rounds = (th + th_max - 1) / th_max;
			th = (rows + per_round * rounds - 1) / (per_round * rounds);
		}
		th = (th + q - 1) / q * q;
		if (th < 2 * q) th = 2 * q;
		if (th > th_max) th = th_max;
		pl->th = env_int("FIXCA_TH", th, 1, WIDE_TH_MAX);
	}
	pl->nstrips = (W + oc - 1) / oc;
	pl->nchunks = (g->y2 - g->y1 + pl->th - 1) / pl->th;
	if (pl->nchunks < 1) pl->nchunks = 1;
	pl->nwaves = pl->nstrips * pl->nchunks;
	pl->nblocks = (pl->nwaves + 7) & ~7;
	return true;
}

/* ---- plan for the NONE byte mover on wide samples (fixca_move.hip) --------------------- */

static int move_col_align(int pb) { return (pb % 16 == 0) ? 1 : (pb % 8 == 0) ? 2 : (pb % 4 == 0) ? 4 : 8; }

/* does every strip's window (its own columns for green / alpha, the red and the blue source columns) fit a ring row
 * of winb bytes that starts on a column the kernel can start it on? */
static bool move_cols_fit(const FixcaGeom *g, int ocpx, int winb)
{
	const int W = g->W, pb = g->bpp, a = move_col_align(pb);
	for (int X0 = 0; X0 < W; X0 += ocpx) {
		const int xl = (X0 + ocpx < W ? X0 + ocpx : W) - 1;
		int lo = X0, hi = xl;
		for (int c = 0; c < 2; ++c) {
			/* the map is monotone for s > 0: the strip's first column has the lowest source column, its last the highest */
			const int f = h_map_int(X0, g->xc, W, g->s[c], g->xs[c]), l = h_map_int(xl, g->xc, W, g->s[c], g->xs[c]);
			if (f < lo) lo = f;
			if (l > hi) hi = l;
		}
		const int U0 = lo & ~(a - 1);
		if ((long long)(hi + 1 - U0) * pb > winb) return false;
	}
	return true;
}

/* Strip widths of the ring kernels of the wide sample types (k_move, k_lin, k_cub).  `widest` is the widest strip whose
 * windows fit a ring row.  A strip whose byte width is a multiple of 64 starts and ends every row store on a sector
 * boundary when the pitch does too, and no sector is shared with the neighbour strip's wave (round 4,
 * profiles/r04_wide_strip_widths.txt: RGBA u16 NONE 252 / 248 / 240 columns 71.8 / 74.3 / 74.8 %, u16 LINEAR 63.9 / 65.7 /
 * 66.2 %, RGBA float CUBIC 58 / 56 columns 59.5 / 61.0 %, RGB u16 NONE 328 / 320: 71.0 / 71.7 %; 15 % narrower loses again:
 * float CUBIC 52 / 48 columns 57.5 / 54.3 %).  Among such widths within 10 % of the widest: the fewest strips, then the
 * fewest idle lanes in the image's last strip (6000 wide: 248 and 240 columns both make 25 strips, 240 makes them full). */
template <typename Fits>
static int sector_aligned_strip(int W, int pb, int step, int widest, bool forced, Fits fits)
{
	if (forced || widest <= 0) return widest;
	int best = 0;
	long best_n = 0, best_pad = 0;
	for (int oc = widest; oc >= step && 10 * oc >= 9 * widest; oc -= step) {
		if (((long long)oc * pb) % 64 != 0 || !fits(oc)) continue;
		const long n = (W + oc - 1) / oc, pad = n * oc;
		if (!best || n < best_n || (n == best_n && pad < best_pad)) { best = oc; best_n = n; best_pad = pad; }
	}
	return best ? best : widest;
}

static bool plan_move(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	if (g->interp != 0 || g->bpc == 1) return false;
	const int W = g->W, pb = g->bpp, rows = g->y2 - g->y1;
	const long long pitch = (long long)W * pb;
	/* RGB u16 of odd width: rows are not whole dwords.  gfx950 moves dwords at halfword addresses (round 4), but a buffer's
	 * range check is per dword: the dword that holds a tile's last source halfword straddles the end of the tile's
	 * descriptor and reads as 0 (tried: tools/fuzz_soak.py caught it at once), and the descriptor must not be stretched past
	 * the image.  k_wcol's one-tap instantiation loads whole pixel pairs inside the row and has no such dword. */
	if (pitch % 4) return false;
	const int span = scan_row_span(g, 0);
	const bool batch = (flags & FIXCA_PLAN_BATCH) != 0;
	/* a strip is a whole number of 16-byte units AND of pixels: a multiple of `step` columns */
	const int step = (pb % 16 == 0) ? 1 : (pb % 8 == 0) ? 2 : (pb % 4 == 0) ? 4 : 8;
	/* Shape of a launch, measured on MI355X with 6000x4000 images (tools/sweep_move.sh, one allocation per format):
	 *   one image per launch: about two waves per CU, each streaming a tall column with ~14 KiB of loads in flight, beats
	 *     every wider or shorter tiling: RGBA float 1-KiB ring rows (64 columns), 800-row tiles, 14 rows in flight 74.6 %
	 *     of the HBM peak (6 in flight, 112-row tiles: 64.0 %); RGBA double 2-KiB rows, 800-row tiles 74.6 % (4-KiB rows,
	 *     267-row tiles: 66.1 %).  16-bit samples cost a wave more LDS time per byte (a 2-byte gather per sample, 4-way
	 *     bank conflicts at a 16-byte lane stride): three waves per CU, 63.7 %.
	 *   many images per launch (fixca_batch_device): the GPU is full whatever the tiling; 32-row tiles of 2-KiB rows,
	 *     10 rows in flight: u16 65-70 %, float 72.7-73.4 %, double 69.6-74.1 %, RGB u16 63 %, RGB float 65 %, RGB double
	 *     70 % (two boxes; 4-KiB rows lose 9 points on RGB u16, 1-KiB rows 8 on RGB double). */
	const int nu_env = env_int("FIXCA_MOVE_NU", 0, 0, MOVE_NU_MAX);
	const bool light = g->nch == 4 && pb >= 16;          /* one 16-byte read and one or two gathers per unit: a wave streams at full speed */
	int nu = nu_env ? nu_env : batch ? 2 : (pb <= 8 || pb == 16) ? 1 : 2;
	if (nu == 3) nu = 2;
	for (; nu >= 1; nu >>= 1) {
		const int rb = MOVE_RB(nu);                      /* a block of rb output rows keeps span + rb - 1 source rows alive */
		const int winb = nu * 1024;
		int depth = env_int("FIXCA_MOVE_DEPTH", 0, 0, MOVE_DEPTH_MAX);
		if (depth == 0) depth = batch ? (pb <= 8 ? 14 : 10) : light ? (14 * 1024 + winb - 1) / winb : 6;    /* u16, batched: 14 rows 66.9 %, 10: 65.3, 6: 63.2 */
		if (depth < 2) depth = 2;
		int ring = env_int("FIXCA_RING", 0, 0, 96);
		if (ring == 0) {
			ring = span + rb - 1 + depth + 1;
			while ((long long)ring * winb > MOVE_LDS_MAX && depth > 2) { --depth; --ring; }     /* large shifts: fewer rows in flight */
		}
		if (ring < span + rb + 1 || (long long)ring * winb > MOVE_LDS_MAX) continue;
		const int oc_env = env_int("FIXCA_OC", 0, 0, 1 << 20);
		int oc_top = 64 * nu * 16 / pb / step * step;    /* 64 lanes x nu units */
		if (oc_env && oc_env < oc_top) oc_top = oc_env / step * step;
		int oc_pick = 0;
		for (int oc = oc_top; oc >= step && oc * pb >= winb / 2 && !oc_pick; oc -= step)
			if (move_cols_fit(g, oc, winb)) oc_pick = oc;
		if (!oc_pick) continue;
		oc_pick = sector_aligned_strip(W, pb, step, oc_pick, oc_env != 0, [&](int oc) { return move_cols_fit(g, oc, winb); });
		pl->kernel = FIXCA_K_MOVE;
		pl->nu = nu;
		pl->depth = depth;
		pl->oc = oc_pick;
		pl->ring = ring;
		pl->nstrips = (W + oc_pick - 1) / oc_pick;
		int th;
		if (batch) {
			th = 32;
		} else {
			/* waves of the launch: two per CU for RGBA pixels of 16 and 32 bytes; formats that gather every sample (RGB) or
			 * two-byte samples need more waves to hide a wave's LDS time: three to four per CU, 6 rows in flight
			 * (one box, tools/sweep_r3g.sh: RGB u16 950 waves 58.5 %, RGB float 925 waves 60.5 %, RGB double 750 waves 69.3 %,
			 * RGBA u16 flat at 56-57 % from 735 to 3087 waves; the nt copy of the same buffers: 65-73 %) */
			const int want = light ? 512 : 900;
			int per_cu = (160 * 1024) / (ring * winb + 64);          /* one generation: never more waves than slots */
			if (per_cu > 16) per_cu = 16;
			if (per_cu < 1) per_cu = 1;
			int chunks = (want + pl->nstrips / 2) / pl->nstrips;
			if (chunks > per_cu * 256 / pl->nstrips) chunks = per_cu * 256 / pl->nstrips;
			if (chunks < 1) chunks = 1;
			th = (rows + chunks - 1) / chunks;
			if (th < 32) th = 32;
		}
		pl->th = env_int("FIXCA_TH", th, 1, 4096);
		if ((long long)(pl->th + 2 * FIXCA_SOURCE_ROWS) * pitch > (1LL << 31)) {
			/* 32-bit offsets into a tile's rows: shorter tiles for very wide images */
			const long long fit = (1LL << 31) / pitch - 2 * FIXCA_SOURCE_ROWS;
			if (fit < 8) return false;
			pl->th = (int)fit;
		}
		pl->nchunks = (rows + pl->th - 1) / pl->th;
		if (pl->nchunks < 1) pl->nchunks = 1;
		pl->nwaves = pl->nstrips * pl->nchunks;
		if ((long long)pl->nwaves + 8 > (1LL << 31) / MOVE_BATCH_MAX) return false;
		pl->nblocks = (pl->nwaves + 7) & ~7;
		return true;
	}
	return false;
}

/* ---- plan for LINEAR on the wide RGBA formats through the LDS ring (fixca_lin.hip) ------------- */

/* every strip's window: its own columns (green, alpha) and both taps of the red and of the blue interpolation */
static bool lin_cols_fit(const FixcaGeom *g, int ocpx, int winb)
{
	const int W = g->W, pb = g->bpp, a = move_col_align(pb);
	for (int X0 = 0; X0 < W; X0 += ocpx) {
		const int xl = (X0 + ocpx < W ? X0 + ocpx : W) - 1;
		int lo = X0, hi = xl;
		for (int c = 0; c < 2; ++c) {
			const int f = (int)floor(h_map_dbl(X0, g->xc, W, g->s[c], g->xs[c]));
			int l = (int)floor(h_map_dbl(xl, g->xc, W, g->s[c], g->xs[c])) + 1;
			if (l > W - 1) l = W - 1;
			if (f < lo) lo = f;
			if (l > hi) hi = l;
		}
		const int U0 = lo & ~(a - 1);
		if ((long long)(hi + 1 - U0) * pb > winb) return false;
	}
	return true;
}

static bool plan_lin(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	if (g->interp != 1 || g->bpc == 1) return false;
	/* RGB: 12- and 24-byte pixels through k_lin_rgb (a lane owns 24 bytes of whole pixels, the row is put together in
	 * LDS); RGB u16 stays with k_wcol */
	if (g->nch == 3 && g->bpc == 2) return false;
	/* 16-bit samples: in the reference's operation order 4 samples per unit, 16 quotients and 70 operations per unit
	 * and row put the fp64 floor above the memory time (33-40 % against k_wcol's 55 %; FIXCA_LIN_U16=1 to measure): only
	 * in the weight form, where the planner allows it (not on shifts by halves, where rounding boundaries are common) */
	const bool fast = g->bpc == 2 && wide_fast_ok(g);
	/* ... and in a batch: 59.8 against 53.0 % for k_wcol six images per launch; one image per launch the best shape
	 * only draws (56.4 against 56.7 %), so k_wcol keeps the single call (profiles/r03_sweep_r3u.txt) */
	if (g->bpc == 2 && !((fast && (flags & FIXCA_PLAN_BATCH)) || fixca_debug_flag("FIXCA_LIN_U16"))) return false;
	const int W = g->W, pb = g->bpp, rows = g->y2 - g->y1;
	const long long pitch = (long long)W * pb;
	if (pitch % 4) return false;
	const int span = scan_row_span(g, 2);
	const bool batch = (flags & FIXCA_PLAN_BATCH) != 0;
	const int step = (pb % 16 == 0) ? 1 : (pb % 8 == 0) ? 2 : (pb % 4 == 0) ? 4 : 8;
	/* Launch shapes measured on MI355X, 6000x4000, one allocation per format (profiles/r03_sweep_r3k.txt; k_wcol on the
	 * same buffers: float 53.1 %, u32 54.3 %, double 64.3 % of the HBM peak; nt copy 65-70 %):
	 *   one image per launch: float 2-KiB rows x 200 rows (980 waves), 10 rows in flight 65.9 % (1-KiB rows x 200: 62.9,
	 *     x 400: 46.0); u32 (three more fp64 operations per sample: more waves) 1-KiB rows x 200 rows (2000 waves) 61.2 %,
	 *     2-KiB rows 55.7; double 2-KiB rows x 400 rows (1000 waves), 6 in flight 72.7 % (x 200 rows, 5 in flight: 51.3);
	 *   many images per launch: 1-KiB rows x 32 rows, 8 in flight: float 69.9 %, u32 64.7 %, double 73.2 %. */
	const bool integer = g->bpc > 0;
	int nu = env_int("FIXCA_MOVE_NU", 0, 0, MOVE_NU_MAX);
	if (nu == 0) nu = (g->bpc == 2) ? 2 : (batch || integer) ? 1 : 2;
	if (nu == 3) nu = 2;
	if (g->nch == 3) nu = 2;                             /* k_lin_rgb has one shape: 2-KiB ring rows, 1536-byte strips */
	for (; nu >= (g->nch == 3 ? 2 : 1); nu >>= 1) {
		const int winb = nu * 1024;
		int depth = env_int("FIXCA_MOVE_DEPTH", 0, 0, MOVE_DEPTH_MAX);
		/* rows in flight (profiles/r03_sweep_r3s.txt): the formats with more arithmetic per byte trade prefetch for occupancy.
		 * RGB (24 bytes and 4 or 2 computed samples per lane, a staging row in LDS): 3-4 rows (float, batched: 4 rows 65.6 %,
		 * 8 rows 55.8, 12 rows 48.1); RGBA u32 4-6 rows (batched: 6 rows 67.0 %, 8 rows 63.0); RGBA float / double 8-10 / 6 */
		if (depth == 0) depth = (g->nch == 3) ? (batch ? 4 : 3) : (g->bpc == 2) ? 3 : integer ? (batch ? 6 : 4) : batch ? 8 : (pb >= 24) ? 6 : 10;
		if (depth < 2) depth = 2;
		int ring = env_int("FIXCA_RING", 0, 0, 96);
		if (ring == 0) {
			ring = span + depth + 1;
			while ((long long)ring * winb > MOVE_LDS_MAX && depth > 2) { --depth; --ring; }
		}
		if (ring < span + 2 || (long long)ring * winb > MOVE_LDS_MAX) continue;
		const int oc_env = env_int("FIXCA_OC", 0, 0, 1 << 20);
		int oc_top = (g->nch == 3 ? 1536 : 64 * nu * 16) / pb / step * step;
		if (oc_env && oc_env < oc_top) oc_top = oc_env / step * step;
		int oc_pick = 0;
		for (int oc = oc_top; oc >= step && oc * pb >= winb / 2 && !oc_pick; oc -= step)
			if (lin_cols_fit(g, oc, winb)) oc_pick = oc;
		if (!oc_pick) continue;
		oc_pick = sector_aligned_strip(W, pb, step, oc_pick, oc_env != 0, [&](int oc) { return lin_cols_fit(g, oc, winb); });
		pl->kernel = FIXCA_K_LIN;
		pl->wide_fast = fast ? 1 : 0;
		pl->nu = nu;
		pl->depth = depth;
		pl->oc = oc_pick;
		pl->ring = ring;
		pl->nstrips = (W + oc_pick - 1) / oc_pick;
		int th;
		if (batch) {
			th = 32;
		} else if (g->nch == 3) {
			/* RGB, one image per launch: many short tiles (float 100 rows x 47 strips = 1880 waves 57.9 %, 191 rows 53.4;
			 * double 40 rows = 9400 waves 65.1 %, 364 rows 56.5): the kernel wants occupancy, not fat waves */
			th = (pb == 12) ? 100 : 40;
		} else {
			/* the kernel has fp64 work to cover (40 operations per 4-byte pixel): one wave per SIMD, two for u32.  ONE
			 * generation: a launch of 1029 waves on the 1024 slots a 32-KiB ring leaves (4 per CU) ran its last 5 waves
			 * alone: 258 instead of 146 us for RGBA float */
			const int want = integer ? 2048 : 1024;
			int per_cu = (160 * 1024) / (ring * winb + 64);
			if (per_cu > 16) per_cu = 16;
			if (per_cu < 1) per_cu = 1;
			int chunks = (want + pl->nstrips / 2) / pl->nstrips;
			if (chunks > per_cu * 256 / pl->nstrips) chunks = per_cu * 256 / pl->nstrips;
			if (chunks < 1) chunks = 1;
			th = (rows + chunks - 1) / chunks;
			if (th < 32) th = 32;
		}
		pl->th = env_int("FIXCA_TH", th, 1, 4096);
		if ((long long)(pl->th + 2 * FIXCA_SOURCE_ROWS) * pitch > (1LL << 31)) {
			const long long fit = (1LL << 31) / pitch - 2 * FIXCA_SOURCE_ROWS;
			if (fit < 8) return false;
			pl->th = (int)fit;
		}
		pl->nchunks = (rows + pl->th - 1) / pl->th;
		if (pl->nchunks < 1) pl->nchunks = 1;
		pl->nwaves = pl->nstrips * pl->nchunks;
		if ((long long)pl->nwaves + 8 > (1LL << 31) / MOVE_BATCH_MAX) return false;
		pl->nblocks = (pl->nwaves + 7) & ~7;
		return true;
	}
	return false;
}

/* ---- plan for CUBIC on the wide RGBA formats through the LDS ring (fixca_cub.hip) --------------- */

/* every strip's window: its own columns (green, alpha) and the four clamped taps of the red and of the blue filter */
static bool cub_cols_fit(const FixcaGeom *g, int ocpx, int winb)
{
	const int W = g->W, pb = g->bpp, a = move_col_align(pb);
	for (int X0 = 0; X0 < W; X0 += ocpx) {
		const int xl = (X0 + ocpx < W ? X0 + ocpx : W) - 1;
		int lo = X0, hi = xl;
		for (int c = 0; c < 2; ++c) {
			const int f = clampi((int)floor(h_map_dbl(X0, g->xc, W, g->s[c], g->xs[c])) - 1, 0, W - 1);
			const int l = clampi((int)floor(h_map_dbl(xl, g->xc, W, g->s[c], g->xs[c])) + 2, 0, W - 1);
			if (f < lo) lo = f;
			if (l > hi) hi = l;
		}
		const int U0 = lo & ~(a - 1);
		if ((long long)(hi + 1 - U0) * pb > winb) return false;
	}
	return true;
}

static bool plan_cub(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	const bool batch = (flags & FIXCA_PLAN_BATCH) != 0;
	/* Where the ring kernel beats k_wcol (profiles/r03_sweep_r3o.txt, one allocation per format): RGBA double always
	 * (64.5 against 58.0 % of the HBM peak six images per launch, 61.0 against 57.9 one image), RGBA float in a batch
	 * (55.6 against 51.2 %; one image per launch 48.7 against 52.1: k_wcol).  u16 (also in the weight form: 38-46
	 * against 45 %) and u32 (43-48 against 51) stay with k_wcol; FIXCA_CUB_ALL=1 plans floats always (tests). */
	if (g->interp != 2 || g->nch != 4) return false;
	if (!(g->bpc == -8 || (g->bpc == -4 && (batch || fixca_debug_flag("FIXCA_CUB_ALL"))))) return false;
	const int W = g->W, pb = g->bpp, rows = g->y2 - g->y1;
	const long long pitch = (long long)W * pb;
	if (pitch % 4) return false;
	const int span = scan_row_span(g, 4);
	const int step = 1;
	/* 1-KiB ring rows, 64-row tiles, many waves: the kernel is bound by its fp64 chains and wants occupancy, not fat waves
	 * (2-KiB rows: 42.9 against 55.2 % for float; 400-row tiles of 2-KiB rows 52.3 against 60.0 % for double) */
	int nu = env_int("FIXCA_MOVE_NU", 0, 0, MOVE_NU_MAX);
	if (nu == 0) nu = 1;
	if (nu == 3) nu = 2;
	for (; nu >= 1; nu >>= 1) {
		const int winb = nu * 1024;
		int depth = env_int("FIXCA_MOVE_DEPTH", 0, 0, MOVE_DEPTH_MAX);
		/* few rows in flight: every KiB of ring is occupancy the fp64 chains need (double, six per launch: 4 rows in
		 * flight 64.5 %, 8 rows 63.0, 12 rows 59.3; float 55.6 / 53.3 / 49.3) */
		if (depth == 0) depth = 4;
		if (depth < 2) depth = 2;
		int ring = env_int("FIXCA_RING", 0, 0, 96);
		if (ring == 0) {
			ring = span + depth + 1;
			while ((long long)ring * winb > MOVE_LDS_MAX && depth > 2) { --depth; --ring; }
		}
		if (ring < span + 2 || (long long)ring * winb > MOVE_LDS_MAX) continue;
		const int oc_env = env_int("FIXCA_OC", 0, 0, 1 << 20);
		int oc_top = 64 * nu * 16 / pb / step * step;
		if (oc_env && oc_env < oc_top) oc_top = oc_env / step * step;
		int oc_pick = 0;
		for (int oc = oc_top; oc >= step && oc * pb >= winb / 2 && !oc_pick; oc -= step)
			if (cub_cols_fit(g, oc, winb)) oc_pick = oc;
		if (!oc_pick) continue;
		oc_pick = sector_aligned_strip(W, pb, step, oc_pick, oc_env != 0, [&](int oc) { return cub_cols_fit(g, oc, winb); });
		pl->kernel = FIXCA_K_CUB;
		pl->nu = nu;
		pl->depth = depth;
		pl->oc = oc_pick;
		pl->ring = ring;
		pl->nstrips = (W + oc_pick - 1) / oc_pick;
		/* a tile's first row builds its windows from 4 source rows and its halo is span rows: in a batch 32 rows lose 2 points,
		 * 128 rows 1-2; one image per launch (double): 48 rows 61.0 %, 64 rows 59.7, 96 rows 57.2 */
		const int th = batch ? 64 : 48;
		pl->th = env_int("FIXCA_TH", th, 1, 4096);
		if ((long long)(pl->th + 2 * FIXCA_SOURCE_ROWS) * pitch > (1LL << 31)) {
			const long long fit = (1LL << 31) / pitch - 2 * FIXCA_SOURCE_ROWS;
			if (fit < 8) return false;
			pl->th = (int)fit;
		}
		pl->nchunks = (rows + pl->th - 1) / pl->th;
		if (pl->nchunks < 1) pl->nchunks = 1;
		pl->nwaves = pl->nstrips * pl->nchunks;
		if ((long long)pl->nwaves + 8 > (1LL << 31) / MOVE_BATCH_MAX) return false;
		pl->nblocks = (pl->nwaves + 7) & ~7;
		return true;
	}
	return false;
}

static int make_plan_uncached(const FixcaGeom *g, unsigned flags, FixcaPlan *pl);

/* The planner walks every column and row of the image (tens of microseconds for 6000x4000,
 * more than the kernel's own launch cost), and a caller such as the plug-in's preview or a
 * batch loop repeats the same geometry: the last plan is kept per thread.  The tuning
 * environment variables are part of the key so that a changed variable re-plans. */
static const char *const plan_env[] = { "FIXCA_DEBUG", "FIXCA_NO_MOVE", "FIXCA_NO_LIN", "FIXCA_NO_CUB", "FIXCA_CUB_ALL", "FIXCA_LIN_U16", "FIXCA_MOVE_NU", "FIXCA_MOVE_DEPTH", "FIXCA_NO_WCOL", "FIXCA_WCOL_REF", "FIXCA_WCOL_WEIGHTS", "FIXCA_NO_ROLL", "FIXCA_NO_SCAN", "FIXCA_RING", "FIXCA_OC",
                                        "FIXCA_TH" };

static void env_signature(char *buf, size_t n)
{
	size_t o = 0;
	for (size_t i = 0; i < sizeof plan_env / sizeof plan_env[0] && o + 1 < n; ++i) {
		const char *e = getenv(plan_env[i]);
		o += (size_t)snprintf(buf + o, n - o, "%s;", e ? e : "-");
	}
	if (o >= n) buf[n - 1] = 0;
}

int fixca_make_plan(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	static thread_local struct { bool valid; FixcaGeom g; unsigned flags; int device; char env[352]; FixcaPlan pl; } last;
	char env[352];
	env_signature(env, sizeof env);
	/* k_wcol's tile height is sized to the occupancy the CURRENT device reports (fixca_wide_slots): after
	 * fixca_set_device() the plan of the previous device must not be handed out.  Only the wide LINEAR / CUBIC plans
	 * depend on the device; the 8-bit path never touches the HIP runtime here. */
	const int device = (g->bpc != 1 && g->interp != 0) ? fixca_wide_device() : -1;
	if (last.valid && last.flags == flags && last.device == device && memcmp(&last.g, g, sizeof *g) == 0 && strcmp(last.env, env) == 0) {
		*pl = last.pl;
		return FIXCA_OK;
	}
	int rc = make_plan_uncached(g, flags, pl);
	if (rc == FIXCA_OK) {
		last.valid = true;
		last.g = *g;
		last.flags = flags;
		last.device = device;
		memcpy(last.env, env, sizeof env);
		last.pl = *pl;
	}
	return rc;
}

static int make_plan_uncached(const FixcaGeom *g, unsigned flags, FixcaPlan *pl)
{
	const unsigned want = flags & FIXCA_KERNEL_MASK;
	memset(pl, 0, sizeof *pl);
	pl->kernel = FIXCA_K_EXACT;
	if (want == FIXCA_KERNEL_EXACT) return FIXCA_OK;
	if (g->bpc != 1) {
		/* 16/32-bit and floating-point samples: the rolling kernel (fp64 for LINEAR / CUBIC, a copy for NONE),
		 * bit-exact like the one-thread-per-pixel kernel, which stays as the checker and as the fallback
		 * for scales that drift more than a window over one strip */
		if (!fixca_debug_flag("FIXCA_NO_MOVE") && plan_move(g, flags, pl)) return FIXCA_OK;
		memset(pl, 0, sizeof *pl);
		if (!fixca_debug_flag("FIXCA_NO_LIN") && plan_lin(g, flags, pl)) return FIXCA_OK;
		memset(pl, 0, sizeof *pl);
		if (!fixca_debug_flag("FIXCA_NO_CUB") && plan_cub(g, flags, pl)) return FIXCA_OK;
		memset(pl, 0, sizeof *pl);
		if (!fixca_debug_flag("FIXCA_NO_WCOL") && plan_wide(g, pl)) return FIXCA_OK;
		memset(pl, 0, sizeof *pl);
		pl->kernel = FIXCA_K_EXACT;
		if (want == FIXCA_KERNEL_FAST)
			return fixca_fail(FIXCA_E_UNSUPPORTED, "no fast kernel for bpc=%d interpolation %d at this scale", g->bpc, g->interp);
		return FIXCA_OK;
	}

	if (!fixca_debug_flag("FIXCA_NO_ROLL") && plan_roll(g, flags, pl)) return FIXCA_OK;
	memset(pl, 0, sizeof *pl);
	if (!fixca_debug_flag("FIXCA_NO_SCAN") && plan_scan(g, pl)) return FIXCA_OK;
	memset(pl, 0, sizeof *pl);
	pl->kernel = FIXCA_K_EXACT;

	/* the register-window kernels load 4 pixels at a time and index rows/columns with int */
	const bool size_ok = g->W >= 8 && (long long)g->W * g->H * g->bpp < (1LL << 40);
	const int th = env_int("FIXCA_TH", 32, 1, 64);

	if (size_ok && g->interp == 0) {
		pl->kernel = FIXCA_K_NONE;
		pl->oc = FIXCA_WAVE_COLS;
		pl->th = th;
		finish_plan(g, pl);
		return FIXCA_OK;
	}
	if (size_ok && g->interp != 0) {
		const int taps = (g->interp == 2) ? 4 : 2;
		int oc_max = env_int("FIXCA_OC", 252, 16, 252) & ~3;
		for (int oc = oc_max; oc >= 64; oc -= 4) {
			if (!interp_fits(g, oc, taps)) continue;
			pl->kernel = FIXCA_K_INTERP;
			pl->oc = oc;
			pl->th = th;
			finish_plan(g, pl);
			return FIXCA_OK;
		}
	}
	if (want == FIXCA_KERNEL_FAST)
		return fixca_fail(FIXCA_E_UNSUPPORTED, "no fast plan for %dx%d scale %.6g/%.6g (width < 8 or scale too far from 1)",
		                  g->W, g->H, g->s[0], g->s[1]);
	return FIXCA_OK;
}

void fixca_plan_text(const FixcaGeom *g, const FixcaPlan *pl, char *buf, size_t n)
{
	static const char *kn[] = { "?", "exact-fp64", "fast-none", "fast-interp", "scan", "(removed)", "roll", "wide-roll-fp64", "move-none", "ring-linear-fp64", "ring-cubic-fp64" };
	static const char *in[] = { "none", "linear", "cubic" };
	if (!buf || !n) return;
	if (pl->kernel == FIXCA_K_EXACT)
		snprintf(buf, n, "%s %s bpp=%d bpc=%d grid=%dx%d", kn[1], in[g->interp], g->bpp, g->bpc, (g->W + 255) / 256, g->y2 - g->y1);
	else if (pl->kernel == FIXCA_K_WCOL)
		snprintf(buf, n, "%s %s bpp=%d bpc=%d oc=%d th=%d strips=%d chunks=%d waves=%d%s", kn[pl->kernel], in[g->interp], g->bpp, g->bpc,
		         pl->oc, pl->th, pl->nstrips, pl->nchunks, pl->nwaves, pl->wide_fast ? " weights+recheck" : "");
	else if (pl->kernel == FIXCA_K_MOVE || pl->kernel == FIXCA_K_LIN || pl->kernel == FIXCA_K_CUB)
		snprintf(buf, n, "%s %s bpp=%d bpc=%d oc=%d th=%d ring=%dx%dK depth=%d strips=%d chunks=%d waves=%d%s", kn[pl->kernel], in[g->interp], g->bpp, g->bpc,
		         pl->oc, pl->th, pl->ring, pl->nu, pl->depth, pl->nstrips, pl->nchunks, pl->nwaves,
		         (pl->kernel == FIXCA_K_LIN && pl->wide_fast) ? " weights+recheck" : "");
	else if (pl->kernel == FIXCA_K_SCAN || pl->kernel == FIXCA_K_ROLL)
		snprintf(buf, n, "%s %s bpp=%d oc=%d th=%d ring=%d strips=%d chunks=%d waves=%d%s", kn[pl->kernel],
		         in[g->interp], g->bpp, pl->oc, pl->th, pl->ring, pl->nstrips, pl->nchunks, pl->nwaves,
		         (pl->kernel == FIXCA_K_ROLL && pl->edge) ? " edge-pixels" : "");
	else
		snprintf(buf, n, "%s %s bpp=%d oc=%d th=%d strips=%d chunks=%d waves=%d blocks=%d", kn[pl->kernel],
		         in[g->interp], g->bpp, pl->oc, pl->th, pl->nstrips, pl->nchunks, pl->nwaves, pl->nblocks);
}

extern "C" int fixca_plan_describe(int W, int H, int bytes, int bpc, const FixCaParams *p,
                                   int x1, int x2, int y1, int y2, int show_progress,
                                   unsigned flags, char *buf, size_t buflen)
{
	FixcaGeom g;
	FixcaPlan pl;
	int rc = fixca_make_geom(W, H, bytes, bpc, p, x1, x2, y1, y2, show_progress, &g);
	if (rc) return rc;
	rc = fixca_make_plan(&g, flags, &pl);
	if (rc) return rc;
	fixca_plan_text(&g, &pl, buf, buflen);
	return pl.kernel;
}
