/*
 * weightbound.c — the error bound k_wcol's weight form rests on (fixca_wide.hip, profiles/NOTES.md).
 *
 * For 16-bit samples the reference computes a CUBIC sample as cubicX(cubicY(rows)) on p = s / 65535 and
 * rounds clip(v) * 65535 (fix-ca.c:896-920, 748-761, 873-880; restated below operation for operation,
 * compile with -ffp-contract=off).  The kernel computes x = sum_t wy_t * sum_i wx_i * s_ti with the
 * Catmull-Rom weights and recomputes a sample the reference's way when fast_round_u16 (restated below, verbatim) flags it: fract(x + 1/2) within 1e-6 of 0 or 1.
 * This program measures max |x_ref - x_weights| over random, extreme-step, nearly flat and ramp windows,
 * random and dyadic fractions, and counts the samples whose two roundings differ although the weight
 * form did not flag them (must be 0).  Prints: <max difference> <unflagged mismatches> <flagged>.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { rs += 0x9E3779B97F4A7C15ull; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double unit(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

static double cubic_ref(double a, double b, double c, double d, double t)     /* fix-ca.c:905-907, :916-918 */
{
	return ((((-a + 3.0 * b - 3.0 * c + d) * t + (2.0 * a - 5.0 * b + 4.0 * c - d)) * t + (-a + c)) * t + (b + b)) / 2.0;
}
static void weights(double t, double w[4])
{
	const double t2 = t * t, t3 = t2 * t;
	w[0] = 0.5 * (-t3 + 2.0 * t2 - t); w[1] = 0.5 * (3.0 * t3 - 5.0 * t2 + 2.0);
	w[2] = 0.5 * (-3.0 * t3 + 4.0 * t2 + t); w[3] = 0.5 * (t3 - t2);
}

/* fast_round_u16 of gimp-fix-ca_amd/csrc/fixca_codec.h, verbatim: t = clip(x) + 1/2 truncated, the band test on fract(t)
 * (v_fract_f64 of a non-negative t is t - floor(t)); near = recompute this sample operation for operation */
static unsigned fast_round_u16(double x, int *near_boundary)
{
	const double t = fmin(fmax(x + 0.5, 0.5), 65535.5);
	const double fr = t - floor(t);
	*near_boundary = fabs(fr - 0.5) > 0.5 - 1e-6;
	return (unsigned)t;
}

int main(void)
{
	double worst = 0.0;
	long unflagged = 0, flagged = 0;
	for (long n = 0; n < 12000000; ++n) {
		double s[4][4];
		const int mode = (int)(n % 4);
		const int base = (int)(rnd() % 65536);
		for (int t = 0; t < 4; ++t)
			for (int i = 0; i < 4; ++i) {
				int v;
				if (mode == 0) v = (int)(rnd() % 65536);
				else if (mode == 1) v = (rnd() & 1) ? 65535 : 0;                      /* extreme steps */
				else if (mode == 2) v = base + (int)(rnd() % 7) - 3;                  /* nearly flat */
				else v = base + 37 * i + 11 * t;                                      /* ramps: exact halves at t = 1/2 */
				s[t][i] = (double)(v < 0 ? 0 : v > 65535 ? 65535 : v);
			}
		double fx = unit(), fy = unit();
		if (n % 8 >= 6) { fx = floor(fx * 8.0) / 8.0; fy = floor(fy * 8.0) / 8.0; }   /* coarse dyadic fractions */
		if (n % 64 == 13) {                                                           /* x within 2^-40 .. 2^-52 of k + 1/2: x + 1/2 itself rounds */
			const int k = (int)(rnd() % 65535);
			for (int i = 0; i < 4; ++i) { s[0][i] = s[1][i] = k; s[2][i] = s[3][i] = k + 1; }
			fx = 0.0;
			fy = 0.5 + ((rnd() & 1) ? 1.0 : -1.0) * ldexp(1.0, -40 - (int)(rnd() % 13));
		}
		double h[4];
		for (int t = 0; t < 4; ++t) h[t] = cubic_ref(s[t][0] / 65535.0, s[t][1] / 65535.0, s[t][2] / 65535.0, s[t][3] / 65535.0, fx);
		double v = cubic_ref(h[0], h[1], h[2], h[3], fy);
		v = v < 0.0 ? 0.0 : v > 1.0 ? 1.0 : v;                                        /* clip_d */
		const double xr = v * 65535.0;
		const long ref = (long)round(xr);                                             /* set_pixel: C round() */
		double wx[4], wy[4], H[4];
		weights(fx, wx); weights(fy, wy);
		for (int t = 0; t < 4; ++t) H[t] = wx[3] * s[t][3] + (wx[2] * s[t][2] + (wx[1] * s[t][1] + wx[0] * s[t][0]));
		double x = wy[3] * H[3] + (wy[2] * H[2] + (wy[1] * H[1] + wy[0] * H[0]));
		x = x < 0.0 ? 0.0 : x > 65535.0 ? 65535.0 : x;
		const double d = fabs(x - xr);
		if (d > worst) worst = d;
		int near_boundary;
		const long fast = (long)fast_round_u16(x, &near_boundary);
		if (near_boundary) ++flagged;
		else if (fast != ref) ++unflagged;
	}
	printf("%.3e %ld %ld\n", worst, unflagged, flagged);
	return 0;
}
