/*
 * fixca_wide.hip — the row-rolling kernel for the sample types beyond 8 bits (SURVEY §8f row 1:
 * get_pixel / set_pixel for u16, u32, float and double samples, fix-ca.c:713-774; color_size codes
 * 2, 4, -4, -8), LINEAR and CUBIC.
 *
 * The arithmetic is the reference's, in fp64, operation for operation (the same helpers as the
 * one-thread-per-pixel kernel k_wide, which stays as the checker and the fallback): the results
 * are bit-identical to the CPU oracle.  What changes is how often each piece is computed.  The
 * map is separable (x' depends on the column only, y' on the row only), so
 *
 *   - get_pixel runs ONCE per source sample and row visit, not once per tap: a lane converts the
 *     samples of its own two window columns and the wave shares the doubles through 2 KB of LDS
 *     (32 get_pixel divisions per output sample in the reference, 1 here);
 *   - the horizontal filter of a source row (cubicY, fix-ca.c:896-909; the inner lerp of
 *     bilinear, fix-ca.c:891) is evaluated once per output column and kept in a rolling window of
 *     TAPS rows in registers: when the output row advances, the window slides by one source row
 *     (clamping is monotone, so the slide is valid at the image edges too) and only the new row is
 *     filtered; the vertical filter (cubicX, fix-ca.c:911-920) runs on the window.  Re-using the
 *     value of an expression is not a change of rounding: every operation still sees the operands
 *     the reference's would.
 *
 * One wave per tile of `oc` = 120 output columns x `th` rows, two pixels per lane (a 16-byte load
 * and store per lane for RGBA u16).  Each channel has its own 128-column source window, so any
 * shift fits; only the drift of the tap offset over a strip is limited (the planner checks it and
 * falls back to k_wide).  Global memory is read with plain loads: a source row is visited up to
 * three times (as red row, blue row, identity row), a few rows apart; the repeats are L1/L2 hits.
 */
#include "fixca_device.h"
#include "fixca_wide.h"

namespace {

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
__device__ __forceinline__ double round_nonneg(double x)
{
	const double t = trunc(x);
	return (__dsub_rn(x, t) >= 0.5) ? __dadd_rn(t, 1.0) : t;
}
template <> struct Codec<uint16_t> {
	static __device__ __forceinline__ double get(uint16_t s) { return div_exact((double)s, 65535.0, 1.0 / 65535.0); }
	static __device__ __forceinline__ uint16_t put(double d) { return (uint16_t)(unsigned)round_nonneg(__dmul_rn(d, 65535.0)); }
};
template <> struct Codec<uint32_t> {
	static __device__ __forceinline__ double get(uint32_t s) { return div_exact((double)s, 4294967295.0, 1.0 / 4294967295.0); }
	static __device__ __forceinline__ uint32_t put(double d) { return (uint32_t)round_nonneg(__dmul_rn(d, 4294967295.0)); }
};
template <> struct Codec<float> {
	static __device__ __forceinline__ double get(float s) { return (double)s; }
	static __device__ __forceinline__ float put(double d) { return (float)d; }
};
template <> struct Codec<double> {
	static __device__ __forceinline__ double get(double s) { return s; }
	static __device__ __forceinline__ double put(double d) { return d; }
};

/* two pixels of NCH samples, as stored: loaded and stored as one unit (16 bytes for RGBA u16) */
/* guaranteed alignment of an even pixel's address: 4 samples per pixel -> a multiple of 4 * sizeof(S) (at most 16 is of
 * use); 3 samples per pixel -> only that of a sample */
template <typename S, int NCH> struct __attribute__((packed, aligned(NCH == 4 ? (4 * sizeof(S) < 16 ? 4 * sizeof(S) : 16) : sizeof(S)))) Px2 { S v[2 * NCH]; };

/* My two pixels are 2 * NCH * sizeof(S) contiguous bytes.  They are moved as 16-byte pieces whenever the address is a
 * multiple of 16 (RGBA: always for even image widths, every other row for odd ones): a wave's store instruction then
 * writes whole 64-byte sectors.  With the 8-byte pieces the type's guaranteed alignment allows, every sector is written
 * twice, half each time, and the write traffic to HBM doubles (measured: WRITE_SIZE 433 MB for a 192 MB image). */
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename S, int NCH>
__device__ __forceinline__ void load_px2(const S *p, Px2<S, NCH> &out)
{
	constexpr int BYTES = 2 * NCH * (int)sizeof(S);
	if constexpr (BYTES % 16 == 0) {
		if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
			const u32x4_t *q = reinterpret_cast<const u32x4_t *>(p);
			u32x4_t *o = reinterpret_cast<u32x4_t *>(&out);
#pragma unroll
			for (int i = 0; i < BYTES / 16; ++i) o[i] = q[i];
			return;
		}
	}
	out = *reinterpret_cast<const Px2<S, NCH> *>(p);
}
template <typename S, int NCH>
__device__ __forceinline__ void store_px2(S *p, const Px2<S, NCH> &in)
{
	constexpr int BYTES = 2 * NCH * (int)sizeof(S);
	if constexpr (BYTES % 16 == 0) {
		if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
			u32x4_t *q = reinterpret_cast<u32x4_t *>(p);
			const u32x4_t *o = reinterpret_cast<const u32x4_t *>(&in);
#pragma unroll
			for (int i = 0; i < BYTES / 16; ++i) __builtin_nontemporal_store(o[i], &q[i]);
			return;
		}
	}
	*reinterpret_cast<Px2<S, NCH> *>(p) = in;
}

template <int TAPS>
__device__ __forceinline__ double hfilter(const double (&p)[TAPS], double fx)
{
	if constexpr (TAPS == 4) return cr_exact(p[0], p[1], p[2], p[3], fx);              /* cubicY, fix-ca.c:905-907 */
	else return __dadd_rn(p[0], __dmul_rn(fx, __dsub_rn(p[1], p[0])));                   /* fix-ca.c:891, inner */
}
template <int TAPS>
__device__ __forceinline__ double vfilter(const double (&h)[TAPS], double fy)
{
	if constexpr (TAPS == 4) return cr_exact(h[0], h[1], h[2], h[3], fy);              /* cubicX, fix-ca.c:916-918 */
	else return __dadd_rn(__dmul_rn(__dsub_rn(1.0, fy), h[0]), __dmul_rn(fy, h[1]));     /* fix-ca.c:891, outer */
}

/* tap t of the window that starts at integer coordinate c0, clamped as the reference clamps it:
 * cubic  max(c0-1,0), c0, min(c0+1,n-1), min(min(c0+1,n-1)+1,n-1)  (fix-ca.c:1218-1256, 1270-1298)
 * linear c0, (c0 == n-1) ? c0 : c0+1                               (fix-ca.c:1149-1158, 1170-1177)
 * both are clamp(c0 - BACK + t, 0, n-1) because 0 <= c0 <= n-1 */
template <int TAPS>
__device__ __forceinline__ int tap_index(int c0, int t, int n)
{
	constexpr int BACK = (TAPS == 4) ? 1 : 0;
	return max(0, min(c0 - BACK + t, n - 1));
}

template <typename S, int NCH, int TAPS>
__global__ __launch_bounds__(64) void
k_wcol(const S *__restrict__ src, S *__restrict__ dst, FixcaGeom g, WidePlan wp)
{
	typedef Codec<S> C;
	typedef Px2<S, NCH> P2;
	constexpr int T = TAPS;
	__shared__ double hbuf[2][128];                       /* one source row's converted samples per channel */
	__shared__ double row_fy[2][WIDE_TH_MAX + 1];         /* per output row of the tile (and one past it): fraction ... */
	__shared__ int    row_y0[2][WIDE_TH_MAX + 1];         /* ... and floor of the source coordinate               */

	const int lane = threadIdx.x;
	const int tile = xcd_remap(blockIdx.x, wp.nblocks);
	if (tile >= wp.nwaves) return;
	const int strip = tile % wp.nstrips, chunk = tile / wp.nstrips;
	const int W = g.W, H = g.H;
	const int X0 = strip * wp.oc;
	const int Y0 = g.y1 + chunk * wp.th;
	const int nrows = min(wp.th, g.y2 - Y0);
	const size_t pitch = (size_t)W * NCH;                /* in samples */
	const int xe = min(X0 + wp.oc, W);                   /* one past my strip's last output column */
	const int x = X0 + 2 * lane;                         /* my two output columns: x, x + 1 */
	const bool act0 = x < xe, act1 = x + 1 < xe;

	/* ---- per-lane column constants: tap slots in the channel's window, fractions --------------- */
	int    U[2];                                          /* window start per channel (even)          */
	int    slot[2][2][T];                                 /* [channel][pixel][tap] -> index in hbuf    */
	double fx[2][2];
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		const double xd0 = map_dbl(X0, g.xc, W, g.s[c], g.xs[c]);
		U[c] = tap_index<T>((int)floor(xd0), 0, W) & ~1;    /* the strip's first column has the lowest tap (the map is monotone) */
		U[c] = __builtin_amdgcn_readfirstlane(U[c]);
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int xx = min(x + j, W - 1);
			const double xd = map_dbl(xx, g.xc, W, g.s[c], g.xs[c]);
			const int c0 = (int)floor(xd);
			fx[c][j] = __dsub_rn(xd, (double)c0);
#pragma unroll
			for (int t = 0; t < T; ++t) slot[c][j][t] = min(tap_index<T>(c0, t, W) - U[c], 127);
		}
	}

	/* the horizontally filtered window rows of my two columns, per channel */
	double hh[2][2][T];
	int    y0prev[2] = { -1000000, -1000000 };

	/* the raw samples of my two window columns of source row r, channel c (loads only: no wait here) */
	auto fetch = [&](int c, int r, S (&raw)[2]) {
		const S *row = src + (size_t)r * pitch + 2 * c;
#pragma unroll
		for (int j = 0; j < 2; ++j) raw[j] = row[(size_t)min(U[c] + 2 * lane + j, W - 1) * NCH];
	};
	/* get_pixel on them, shared with the wave through LDS, then the horizontal filter of my two columns */
	auto absorb = [&](int c, const S (&raw)[2], double (&out)[2]) {
#pragma unroll
		for (int j = 0; j < 2; ++j) hbuf[c][2 * lane + j] = C::get(raw[j]);
		/* one wave: LDS operations of a wave complete in order, the reads below see the writes above */
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			double p[T];
#pragma unroll
			for (int t = 0; t < T; ++t) p[t] = hbuf[c][slot[c][j][t]];
			out[j] = hfilter<T>(p, fx[c][j]);
		}
	};
	/* The row coordinates are the same for every lane: lane L computes those of row Y0 + L (and
	 * Y0 + 64 + L) once per tile instead of every lane computing all of them every row. */
	for (int i = lane; i <= nrows; i += 64) {
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const double yd = map_dbl(min(Y0 + i, H - 1), g.yc, H, g.s[c], g.ys[c]);
			const int y0 = (int)floor(yd);
			row_y0[c][i] = y0;
			row_fy[c][i] = __dsub_rn(yd, (double)y0);
		}
	}

	/* Software pipeline: while row y is computed, the loads of row y + 1 (the one new source row per
	 * channel of a sliding window, and the pixels to copy green/alpha from) are already in flight. */
	S   pend[2][2];                                       /* prefetched raw samples per channel      */
	int pend_row[2] = { -1, -1 };                         /* the source row they belong to           */
	P2  idt;                                              /* my two pixels of the current output row */
	const size_t xoff = (size_t)min(x, W - 2) * NCH;      /* W >= 2 is guaranteed by the planner      */
	load_px2<S, NCH>(src + (size_t)Y0 * pitch + xoff, idt);

	for (int r = 0; r < nrows; ++r) {
		const int y = Y0 + r;
		int    y0[2];
		double fy[2];
		S      cur[2][2];
		int    cur_row[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			y0[c] = __builtin_amdgcn_readfirstlane(row_y0[c][r]);
			fy[c] = row_fy[c][r];
			cur_row[c] = pend_row[c];
			cur[c][0] = pend[c][0]; cur[c][1] = pend[c][1];
		}
		/* ---- issue the next row's loads ------------------------------------------------------ */
		P2 idt_next = idt;
		if (r + 1 < nrows) {
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const int y0n = __builtin_amdgcn_readfirstlane(row_y0[c][r + 1]);
				pend_row[c] = -1;
				if (y0n == y0[c] + 1) {
					pend_row[c] = tap_index<T>(y0n, T - 1, H);
					fetch(c, pend_row[c], pend[c]);
				}
			}
			load_px2<S, NCH>(src + (size_t)(y + 1) * pitch + xoff, idt_next);
		}
		/* ---- this row ------------------------------------------------------------------------ */
		double v[2][2];                                    /* [channel][pixel] */
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			if (y0[c] == y0prev[c] + 1) {
				/* slide by one source row: row k of the new window is row k + 1 of the old one */
				const int need = tap_index<T>(y0[c], T - 1, H);
				if (cur_row[c] != need) fetch(c, need, cur[c]);        /* never in steady state */
				double nw[2];
				absorb(c, cur[c], nw);
#pragma unroll
				for (int j = 0; j < 2; ++j) {
#pragma unroll
					for (int t = 0; t + 1 < T; ++t) hh[c][j][t] = hh[c][j][t + 1];
					hh[c][j][T - 1] = nw[j];
				}
			} else if (y0[c] != y0prev[c]) {
				/* first row of the tile, or a scale step that skips a row: rebuild the window */
#pragma unroll
				for (int t = 0; t < T; ++t) {
					S raw[2];
					double nw[2];
					fetch(c, tap_index<T>(y0[c], t, H), raw);
					absorb(c, raw, nw);
					hh[c][0][t] = nw[0]; hh[c][1][t] = nw[1];
				}
			}                                               /* y0 == y0prev: the same window again */
			y0prev[c] = y0[c];
#pragma unroll
			for (int j = 0; j < 2; ++j) v[c][j] = clip01(vfilter<T>(hh[c][j], fy[c]));   /* clip_d, fix-ca.c:873-880 */
		}
		/* green and alpha are copied (fix-ca.c:1098), red and blue replaced */
		if (act0) {
			S *out = dst + (size_t)y * pitch + (size_t)x * NCH;
			if (act1) {
				P2 px = idt;
				px.v[0] = C::put(v[0][0]); px.v[2] = C::put(v[1][0]);
				px.v[NCH] = C::put(v[0][1]); px.v[NCH + 2] = C::put(v[1][1]);
				store_px2<S, NCH>(out, px);
			} else {
				/* the odd last column of an odd-width image: x == W - 1, idt was loaded from column W - 2 */
				S px[NCH];
#pragma unroll
				for (int k = 0; k < NCH; ++k) px[k] = idt.v[NCH + k];
				px[0] = C::put(v[0][0]); px[2] = C::put(v[1][0]);
#pragma unroll
				for (int k = 0; k < NCH; ++k) out[k] = px[k];
			}
		}
		idt = idt_next;
	}
}

template <typename S>
hipError_t launch_typed(const void *d_src, void *d_dst, const FixcaGeom *g, const WidePlan *wp, hipStream_t stream)
{
	const S *s = static_cast<const S *>(d_src);
	S *d = static_cast<S *>(d_dst);
	const dim3 grid((unsigned)wp->nblocks), block(64);
	if (g->interp == 2) {
		if (g->nch == 4) hipLaunchKernelGGL((k_wcol<S, 4, 4>), grid, block, 0, stream, s, d, *g, *wp);
		else             hipLaunchKernelGGL((k_wcol<S, 3, 4>), grid, block, 0, stream, s, d, *g, *wp);
	} else {
		if (g->nch == 4) hipLaunchKernelGGL((k_wcol<S, 4, 2>), grid, block, 0, stream, s, d, *g, *wp);
		else             hipLaunchKernelGGL((k_wcol<S, 3, 2>), grid, block, 0, stream, s, d, *g, *wp);
	}
	return hipGetLastError();
}

}   /* namespace */

extern "C" hipError_t
fixca_launch_wide(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	if (g->y2 <= g->y1) return hipSuccess;
	if (g->interp != 1 && g->interp != 2) return hipErrorInvalidValue;
	WidePlan wp;
	wp.oc = pl->oc; wp.th = pl->th; wp.nstrips = pl->nstrips; wp.nwaves = pl->nwaves; wp.nblocks = pl->nblocks;
	switch (g->bpc) {
	case 2:  return launch_typed<uint16_t>(d_src, d_dst, g, &wp, stream);
	case 4:  return launch_typed<uint32_t>(d_src, d_dst, g, &wp, stream);
	case -4: return launch_typed<float>(d_src, d_dst, g, &wp, stream);
	case -8: return launch_typed<double>(d_src, d_dst, g, &wp, stream);
	default: return hipErrorInvalidValue;
	}
}
