/*
 * weightbound_linear.c — the error bound the 16-bit LINEAR weight form rests on (k_lin<u16, .., FAST>, fixca_lin.hip; fixca_codec.h).
 *
 * The reference computes (1 - fy) * (p00 + fx * (p10 - p00)) + fy * (p01 + fx * (p11 - p01)) on p = s / 65535 and rounds
 * clip(v) * 65535 (fix-ca.c:882-894, 748-761, 873-880; restated below operation for operation, compile with
 * -ffp-contract=off).  The kernel computes h_t = fma(fx, s_t1 - s_t0, s_t0), x = fma(fy, h_1 - h_0, h_0) on the integer
 * samples and recomputes a sample the reference's way when fast_round_u16 (restated below, verbatim) flags it: fract(x + 1/2) within 1e-6 of 0 or 1.  This program measures
 * max |x_ref - x_weights| over random, extreme-step, nearly flat and ramp windows, random and dyadic fractions, and
 * counts the samples whose two roundings differ although the weight form did not flag them (must be 0).
 * Prints: <max difference> <unflagged mismatches> <flagged>.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { rs += 0x9E3779B97F4A7C15ull; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double unit(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

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
	for (long n = 0; n < 24000000; ++n) {
		double s[2][2];
		const int mode = (int)(n % 4);
		const int base = (int)(rnd() % 65536);
		for (int t = 0; t < 2; ++t)
			for (int i = 0; i < 2; ++i) {
				int v;
				if (mode == 0) v = (int)(rnd() % 65536);
				else if (mode == 1) v = (rnd() & 1) ? 65535 : 0;                      /* extreme steps */
				else if (mode == 2) v = base + (int)(rnd() % 7) - 3;                  /* nearly flat */
				else v = base + 37 * i + 11 * t;                                      /* ramps: exact halves at t = 1/2 */
				s[t][i] = (double)(v < 0 ? 0 : v > 65535 ? 65535 : v);
			}
		double fx = unit(), fy = unit();
		if (n % 8 >= 6) { fx = floor(fx * 8.0) / 8.0; fy = floor(fy * 8.0) / 8.0; }   /* coarse dyadic fractions */
		if (n % 64 == 5) fx = 0.0;
		if (n % 64 == 9) fy = 0.0;
		if (n % 64 == 13) {                                                           /* x within 2^-40 .. 2^-52 of k + 1/2: x + 1/2 itself rounds */
			const int k = (int)(rnd() % 65535);
			s[0][0] = s[0][1] = k; s[1][0] = s[1][1] = k + 1;
			fx = 0.0;
			fy = 0.5 + ((rnd() & 1) ? 1.0 : -1.0) * ldexp(1.0, -40 - (int)(rnd() % 13));
		}
		const double p00 = s[0][0] / 65535.0, p10 = s[0][1] / 65535.0, p01 = s[1][0] / 65535.0, p11 = s[1][1] / 65535.0;
		double v = (1.0 - fy) * (p00 + fx * (p10 - p00)) + fy * (p01 + fx * (p11 - p01));      /* fix-ca.c:891-892 */
		v = v < 0.0 ? 0.0 : v > 1.0 ? 1.0 : v;                                        /* clip_d */
		const double xr = v * 65535.0;
		const long ref = (long)round(xr);                                             /* set_pixel: C round() */
		const double h0 = fma(fx, s[0][1] - s[0][0], s[0][0]), h1 = fma(fx, s[1][1] - s[1][0], s[1][0]);
		double x = fma(fy, h1 - h0, h0);
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
