/*
 * fixca_device.h — device helpers shared by the gfx950 kernel files: the fp64
 * coordinate arithmetic of the reference (never contracted to FMA), the tap
 * weights, v_perm_b32 and the XCD-aware tile remap.
 */
#ifndef FIXCA_DEVICE_H
#define FIXCA_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fixca_geom.h"

#define WAVE 64
#define LANE_PX 4

/* ------------------------------------------------------------------------ */
/* fp64 helpers: explicit round-to-nearest ops are never contracted to FMA   */
/* ------------------------------------------------------------------------ */

__device__ __forceinline__ double src_coord(int i, int centre, double s, double shift)
{
	/* (i - centre) * s + centre - shift, left to right (fix-ca.c:801, :813) */
	return __dsub_rn(__dadd_rn(__dmul_rn((double)(i - centre), s), (double)centre), shift);
}

__device__ __forceinline__ int rnd_half_away(double d)      /* fix-ca.c:776-789 */
{
	if (d >= 0) {
		if (d > 2147483647.0) return 2147483647;
		return (int)__dadd_rn(d, 0.5);
	}
	if (d < -2147483648.0) return (-2147483647 - 1);
	return -((int)__dsub_rn(0.5, d));
}

__device__ __forceinline__ int map_int(int i, int centre, int size, double s, double shift)
{
	int j = rnd_half_away(src_coord(i, centre, s, shift));   /* fix-ca.c:799-809 */
	if (j <= 0) return 0;
	if (j >= size) return size - 1;
	return j;
}

__device__ __forceinline__ double map_dbl(int i, int centre, int size, double s, double shift)
{
	double d = src_coord(i, centre, s, shift);               /* fix-ca.c:811-820 */
	if (d <= 0.0) return 0.0;
	if (d >= (double)(size - 1)) return (double)(size - 1);
	return d;
}

__device__ __forceinline__ double clip01(double d)
{
	if (d <= 0.0) return 0.0;
	if (d >= 1.0) return 1.0;
	return d;
}

/* Catmull-Rom in the reference's operation order (fix-ca.c:905-907, :916-918) */
__device__ __forceinline__ double cr_exact(double a, double b, double c, double d, double t)
{
	double k3 = __dadd_rn(__dsub_rn(__dadd_rn(-a, __dmul_rn(3.0, b)), __dmul_rn(3.0, c)), d);
	double k2 = __dsub_rn(__dadd_rn(__dsub_rn(__dmul_rn(2.0, a), __dmul_rn(5.0, b)), __dmul_rn(4.0, c)), d);
	double k1 = __dadd_rn(-a, c);
	double r = __dadd_rn(__dmul_rn(k3, t), k2);
	r = __dadd_rn(__dmul_rn(r, t), k1);
	r = __dadd_rn(__dmul_rn(r, t), __dadd_rn(b, b));
	return __ddiv_rn(r, 2.0);
}


/* v_perm_b32: result byte i = byte sel[i] of the 64-bit value {hi, lo}
 * (selector 0-3 -> lo bytes 0-3, 4-7 -> hi bytes 0-3, 0x0c -> 0x00). */
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
	return __builtin_amdgcn_perm(hi, lo, sel);
}

/* XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so
 * give every XCD a contiguous range of tiles; nblocks is a multiple of 8. */
__device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
	return (bid & 7) * (nblocks >> 3) + (bid >> 3);
}

/* Catmull-Rom tap weights at parameter t: value = sum w[j] * p[j], p = taps -1,0,+1,+2.
 * Same polynomial as fix-ca.c:905-907 regrouped by sample; computed in fp64, used in fp32. */
__device__ __forceinline__ void cr_weights(double t, float w[4])
{
	double t2 = t * t, t3 = t2 * t;
	w[0] = (float)((-t3 + 2.0 * t2 - t) * 0.5);
	w[1] = (float)((3.0 * t3 - 5.0 * t2 + 2.0) * 0.5);
	w[2] = (float)((-3.0 * t3 + 4.0 * t2 + t) * 0.5);
	w[3] = (float)((t3 - t2) * 0.5);
}

template <int TAPS>
__device__ __forceinline__ void tap_weights(double t, float w[TAPS])
{
	if (TAPS == 4) {
		cr_weights(t, w);
	} else {
		w[0] = (float)(1.0 - t);
		w[1] = (float)t;
	}
}

/* The same tap weights evaluated in fp32 from the (exactly computed) fp64 fraction: used where
 * many weights are needed per wave and 1e-7 relative error is far inside the 1-LSB budget.
 * A fraction of exactly 0 gives exactly (0,1,0,0) / (1,0). */
template <int TAPS>
__device__ __forceinline__ void tap_weights_f32(float t, float w[TAPS])
{
	if (TAPS == 4) {
		float t2 = t * t, t3 = t2 * t;
		w[0] = (-t3 + 2.0f * t2 - t) * 0.5f;
		w[1] = (3.0f * t3 - 5.0f * t2 + 2.0f) * 0.5f;
		w[2] = (-3.0f * t3 + 4.0f * t2 + t) * 0.5f;
		w[3] = (t3 - t2) * 0.5f;
	} else {
		w[0] = 1.0f - t;
		w[1] = t;
	}
}

#endif
