/*
 * fixca_codec.h — get_pixel / set_pixel (fix-ca.c:713-774) for the sample types beyond 8 bits, as the fp64 kernels use
 * them: sample <-> double in [0, 1], bit for bit the reference's results.  Shared by fixca_wide.hip and fixca_lin.hip.
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


}   /* namespace fixca_codec */

#endif
