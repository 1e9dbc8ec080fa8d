/*
 * fixca_codec.h — get_pixel / set_pixel (fix-ca.c:713-774) for the sample types beyond 8 bits, as the fp64 kernels use
 * them: sample <-> double in [0, 1], bit for bit the reference's results; and the 16-bit CUBIC weight form's pieces (the
 * Catmull-Rom weights, the operation-for-operation recomputation of one sample).  Shared by fixca_wide.hip, fixca_lin.hip
 * and fixca_cub.hip.
 */
#ifndef FIXCA_CODEC_H
#define FIXCA_CODEC_H

#include "fixca_device.h"

namespace fixca_codec {

template <typename S> struct Codec;
/* get_pixel divides by 2^n - 1 (fix-ca.c:720-726).  q = x * r, q' = fma(fma(-q, d, x), r, q) with r = RN(1 / d) is the
 * correctly rounded quotient x / d for EVERY u16 and EVERY u32 value x (checked exhaustively against the division:
 * tests/test_oracle_golden.py::test_reciprocal_division_is_exact), at 3 operations instead of the ~10 of a division. */
__device__ __forceinline__ double div_exact(double x, double d, double r)
{
	const double q = __dmul_rn(x, r);
	return __fma_rn(__fma_rn(-q, d, x), r, q);
}
/* C round() (half away from zero, set_pixel fix-ca.c:752-761) of a value in [0, 2^32): the integer part plus one if
 * the (exactly computed) fraction is at least one half.  Not floor(x + 0.5): that rounds 0.49999999999999994 up. */
__device__ __forceinline__ uint32_t round_nonneg(double x)
{
	/* v_fract_f64 is x - floor(x), exact for these magnitudes; the conversion truncates */
	return (uint32_t)x + (__builtin_amdgcn_fract(x) >= 0.5 ? 1u : 0u);
}
template <> struct Codec<uint16_t> {
	static __device__ __forceinline__ double get(uint16_t s) { return div_exact((double)s, 65535.0, 1.0 / 65535.0); }
	static __device__ __forceinline__ uint16_t put(double d) { return (uint16_t)round_nonneg(__dmul_rn(d, 65535.0)); }
	static constexpr bool INTEGER = true;
};
template <> struct Codec<uint32_t> {
	static __device__ __forceinline__ double get(uint32_t s) { return div_exact((double)s, 4294967295.0, 1.0 / 4294967295.0); }
	static __device__ __forceinline__ uint32_t put(double d) { return round_nonneg(__dmul_rn(d, 4294967295.0)); }
	static constexpr bool INTEGER = true;
};
template <> struct Codec<float> {
	static __device__ __forceinline__ double get(float s) { return (double)s; }
	static __device__ __forceinline__ float put(double d) { return (float)d; }
	static constexpr bool INTEGER = false;
};
template <> struct Codec<double> {
	static __device__ __forceinline__ double get(double s) { return s; }
	static __device__ __forceinline__ double put(double d) { return d; }
	static constexpr bool INTEGER = false;
};


__device__ __forceinline__ int tap_index_rt(int c0, int t, int n) { return max(0, min(c0 - 1 + t, n - 1)); }   /* cubic taps, see tap_index */
/* ---- 16-bit samples, CUBIC: the same results from a quarter of the arithmetic ---------------------
 * The reference's value of a sample is v = cubicX(cubicY(row 0..3)) on p = s / 65535, then round(clip(v) * 65535)
 * (fix-ca.c:896-920, 748-761): 38 fp64 operations in a fixed order plus 16 divisions.  In real arithmetic v * 65535 is
 * x = sum_t wy_t(fy) * sum_i wx_i(fx) * s_ti with the Catmull-Rom weights below, 8 fused multiply-adds on the integer
 * samples.  The two differ only by rounding errors: every one of the reference's 19 + 19 operations and 16 divisions
 * rounds a value of magnitude <= 12 (in units of p) by <= 2^-53 relative, the second filter multiplies the first one's
 * errors by at most sum |w| <= 1.5, so |x_ref - x| <= (230 + 1.5 * 230 + 16) * 12 * 1.2e-16 * 65535 = 5.6e-8 samples
 * (a worst-case bound: every operation at full magnitude with errors of one sign), and the weight form's own error is
 * below 1e-9 (measured over 6e7 random, extreme-step and nearly flat windows, numpy against the operation-for-operation
 * form: 9.8e-11; tests/c/weightbound.c measures a largest difference of 8.4e-11 between the two forms and asserts
 * < 1e-8).  The rounded sample can differ only when a rounding boundary k + 0.5 lies between the two values (clipping
 * to [0, 65535] moves both to the same integer): so x is rounded as the reference rounds, and the rare sample whose
 * fraction is within WIDE_FAST_EPS = 1e-6 of 0.5 (18 x the analytic bound, 10^4 x the measured difference) is
 * recomputed operation for operation from the source image (exact_cubic_sample, the one-thread-per-pixel kernel's
 * code).  The planner (plan_wide) keeps the operation-for-operation variant for geometries whose fractions sit on a
 * coarse dyadic grid (pure shifts by halves or quarters), where such samples are common. */
#define WIDE_FAST_EPS 1e-6
static_assert(WIDE_FAST_EPS >= 10 * 5.6e-8, "the recheck band must stay at least 10 x the analytic bound on |x_ref - x|");

/* The weight forms' last step for 16-bit samples: x (in sample units, the reference's clip(x_ref) * 65535 up to the bound
 * above) rounded as the reference rounds (set_pixel: round(), half away from zero; x >= 0 after the clip), and whether
 * x is within WIDE_FAST_EPS of a rounding boundary k + 1/2, i.e. whether the sample has to be recomputed operation for
 * operation.  t = clip(x) + 1/2 is truncated (the addition is exact to 2^-37 at this magnitude, far inside the band),
 * and x is near k + 1/2 exactly when t is near an integer: fract(t) near 0 or 1.  Five vector instructions fewer per
 * sample than fract / compare / add-with-carry on x itself; the comparison's lane mask stays in scalar registers. */
__device__ __forceinline__ unsigned fast_round_u16(double x, bool &near_boundary)
{
	const double t = __builtin_fmin(__builtin_fmax(x + 0.5, 0.5), 65535.5);
	const double fr = __builtin_amdgcn_fract(t);
	near_boundary = __builtin_fabs(fr - 0.5) > 0.5 - WIDE_FAST_EPS;
	return (unsigned)t;
}
__device__ __forceinline__ void cr_weights(double t, double (&w)[4])
{
	const double t2 = t * t, t3 = t2 * t;
	w[0] = 0.5 * (-t3 + 2.0 * t2 - t);
	w[1] = 0.5 * (3.0 * t3 - 5.0 * t2 + 2.0);
	w[2] = 0.5 * (-3.0 * t3 + 4.0 * t2 + t);
	w[3] = 0.5 * (t3 - t2);
}
/* one output sample of channel c (0 red, 1 blue) at (x, y), operation for operation (fix-ca.c:1260-1318) */
template <typename S, int NCH>
__device__ __forceinline__ S exact_cubic_sample(const S *__restrict__ src, const FixcaGeom &g, int x, int y, int c)
{
	typedef Codec<S> C;
	const size_t pitch = (size_t)g.W * NCH;
	const double yd = map_dbl(y, g.yc, g.H, g.s[c], g.ys[c]), xd = map_dbl(x, g.xc, g.W, g.s[c], g.xs[c]);
	const int r0 = (int)floor(yd), c0 = (int)floor(xd);
	const double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
	double h[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const S *row = src + (size_t)tap_index_rt(r0, t, g.H) * pitch + 2 * c;
		h[t] = cr_exact(C::get(row[(size_t)tap_index_rt(c0, 0, g.W) * NCH]), C::get(row[(size_t)tap_index_rt(c0, 1, g.W) * NCH]),
		                C::get(row[(size_t)tap_index_rt(c0, 2, g.W) * NCH]), C::get(row[(size_t)tap_index_rt(c0, 3, g.W) * NCH]), fx);
	}
	return C::put(clip01(cr_exact(h[0], h[1], h[2], h[3], fy)));
}


/* one output sample of channel c (0 red, 1 blue) at (x, y) for LINEAR, operation for operation (fix-ca.c:1160-1185,
 * bilinear :882-894): what the 16-bit LINEAR weight form (fixca_lin.hip) falls back to near a rounding boundary.  The
 * form computes h = fma(fx, s1 - s0, s0) per row and x = fma(fy, h1 - h0, h0) on the integer samples; the reference's
 * value differs from it by rounding errors only: 4 quotients s / 65535, 3 + 3 + 4 operations on values <= 2 and the final
 * multiplication, each <= 2^-53 relative: |x_ref - x| < 15 * 2.2e-16 * 65535 = 2.2e-10 samples (measured over 24 million
 * windows: 2.2e-11, tests/c/weightbound_linear.c), the form's own error < 4 * 2^-53 * 65535 = 3e-11: far inside
 * WIDE_FAST_EPS. */
template <typename S, int NCH>
__device__ __forceinline__ S exact_linear_sample(const S *__restrict__ src, const FixcaGeom &g, int x, int y, int c)
{
	typedef Codec<S> C;
	const size_t pitch = (size_t)g.W * NCH;
	const double yd = map_dbl(y, g.yc, g.H, g.s[c], g.ys[c]), xd = map_dbl(x, g.xc, g.W, g.s[c], g.xs[c]);
	const int r0 = (int)floor(yd), c0 = (int)floor(xd);
	const double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
	const int r1 = (r0 == g.H - 1) ? r0 : r0 + 1, c1 = (c0 == g.W - 1) ? c0 : c0 + 1;
	const S *row0 = src + (size_t)r0 * pitch + 2 * c, *row1 = src + (size_t)r1 * pitch + 2 * c;
	const double p00 = C::get(row0[(size_t)c0 * NCH]), p10 = C::get(row0[(size_t)c1 * NCH]);
	const double p01 = C::get(row1[(size_t)c0 * NCH]), p11 = C::get(row1[(size_t)c1 * NCH]);
	const double a = __dadd_rn(p00, __dmul_rn(fx, __dsub_rn(p10, p00)));
	const double b = __dadd_rn(p01, __dmul_rn(fx, __dsub_rn(p11, p01)));
	return C::put(clip01(__dadd_rn(__dmul_rn(__dsub_rn(1.0, fy), a), __dmul_rn(fy, b))));
}

}   /* namespace fixca_codec */

#endif
