/*
 * fixca_kernels.hip — gfx950 (MI355X, CDNA4) kernels for the Fix-CA pixel loop.
 *
 * What is computed (reference fix-ca.c:998-1348; arithmetic per SURVEY.md App. A):
 * green and alpha are copied, red and blue are resampled from a copy of
 * themselves scaled about the lens centre and shifted:
 *     x' = (x - xc) * s_c + xc - xshift_c      (depends on the column only)
 *     y' = (y - yc) * s_c + yc - yshift_c      (depends on the row only)
 * with none / bilinear / Catmull-Rom interpolation.  The map is separable, which
 * is what the fast kernels are built on.
 *
 * Three kernels:
 *   k_exact   one thread per pixel, fp64, every operation in the reference's
 *             order with no FMA contraction: bit-exact in all three modes.  Also
 *             the fallback for geometries the fast plan does not cover.
 *   k_none    interpolation NONE: a byte gather.  One lane = 4 adjacent pixels of
 *             a 256-column strip, walking down up to 64 rows; column indices are
 *             per-lane constants, row indices come from a per-wave row table.
 *             Bit-exact (same fp64 coordinate arithmetic as k_exact).
 *   k_interp  LINEAR / CUBIC row scan, fp32 sample arithmetic on fp64
 *             coordinates (within 1 LSB of the reference):
 *               - a wave owns `oc` output columns x up to 64 rows; lanes are
 *                 independent of other waves (no workgroup barrier);
 *               - per-row quantities (source row, vertical weights) are computed
 *                 once per wave, one row per lane, and broadcast with readlane;
 *               - per-column quantities (tap position, horizontal weights) are
 *                 lane constants computed once;
 *               - the vertical window (2 or 4 source rows of this channel's bytes
 *                 for the lane's 4 source columns) rolls through registers, one
 *                 new 16-byte coalesced row segment per output row and channel,
 *                 prefetched one row ahead;
 *               - the vertical filter runs first on the lane's own columns, the
 *                 256 results per channel are exchanged through LDS and every
 *                 lane gathers the 2 / 4 horizontal taps of its 4 outputs.
 *             HBM traffic is one read of the source and one write of the result;
 *             re-reads of a source row by the red window, the blue window and
 *             the identity copy are L2 hits.
 *
 * No MFMA: this is a gather/warp, not a contraction.  Wave size is 64 throughout.
 */
#include "fixca_device.h"

/* ------------------------------------------------------------------------ */
/* k_exact: one thread per pixel, fp64, bit-exact                            */
/* ------------------------------------------------------------------------ */

template <int BPP>
__global__ __launch_bounds__(256) void
k_exact(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, FixcaGeom g)
{
	/* sample/255 exactly as get_pixel computes it (fix-ca.c:717-719) */
	__shared__ double unit[256];
	unit[threadIdx.x] = __ddiv_rn((double)threadIdx.x, 255.0);
	__syncthreads();

	const int x = blockIdx.x * 256 + threadIdx.x;
	if (x >= g.W) return;
	const size_t pitch = (size_t)g.W * BPP;

	for (int y = g.y1 + blockIdx.y; y < g.y2; y += gridDim.y) {
	const uint8_t *in = src + (size_t)y * pitch + (size_t)x * BPP;
	uint8_t px[4];
	px[0] = in[0]; px[1] = in[1]; px[2] = in[2];
	if (BPP == 4) px[3] = in[3];

#pragma unroll
	for (int c = 0; c < 2; ++c) {
		const int off = 2 * c;
		const double s = g.s[c], xs = g.xs[c], ys = g.ys[c];
		if (g.interp == 0) {
			int sy = map_int(y, g.yc, g.H, s, ys);
			int sx = map_int(x, g.xc, g.W, s, xs);
			px[off] = src[(size_t)sy * pitch + (size_t)sx * BPP + off];
		} else if (g.interp == 1) {
			double yd = map_dbl(y, g.yc, g.H, s, ys);
			double xd = map_dbl(x, g.xc, g.W, s, xs);
			int r0 = (int)floor(yd), c0 = (int)floor(xd);
			double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
			int r1 = (r0 == g.H - 1) ? r0 : r0 + 1;
			int c1 = (c0 == g.W - 1) ? c0 : c0 + 1;
			const uint8_t *row0 = src + (size_t)r0 * pitch + off;
			const uint8_t *row1 = src + (size_t)r1 * pitch + off;
			double p00 = unit[row0[(size_t)c0 * BPP]], p10 = unit[row0[(size_t)c1 * BPP]];
			double p01 = unit[row1[(size_t)c0 * BPP]], p11 = unit[row1[(size_t)c1 * BPP]];
			double a = __dadd_rn(p00, __dmul_rn(fx, __dsub_rn(p10, p00)));
			double b = __dadd_rn(p01, __dmul_rn(fx, __dsub_rn(p11, p01)));
			double v = __dadd_rn(__dmul_rn(__dsub_rn(1.0, fy), a), __dmul_rn(fy, b));
			px[off] = (uint8_t)round(__dmul_rn(clip01(v), 255.0));
		} else {
			double yd = map_dbl(y, g.yc, g.H, s, ys);
			double xd = map_dbl(x, g.xc, g.W, s, xs);
			int r0 = (int)floor(yd), c0 = (int)floor(xd);
			double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
			int rr[4], cc[4];
			rr[1] = r0; rr[0] = (r0 == 0) ? r0 : r0 - 1;
			rr[2] = (r0 == g.H - 1) ? r0 : r0 + 1;
			rr[3] = (rr[2] == g.H - 1) ? rr[2] : rr[2] + 1;
			cc[1] = c0; cc[0] = (c0 == 0) ? c0 : c0 - 1;
			cc[2] = (c0 == g.W - 1) ? c0 : c0 + 1;
			cc[3] = (cc[2] == g.W - 1) ? cc[2] : cc[2] + 1;
			double h[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint8_t *row = src + (size_t)rr[j] * pitch + off;
				h[j] = cr_exact(unit[row[(size_t)cc[0] * BPP]], unit[row[(size_t)cc[1] * BPP]],
				                unit[row[(size_t)cc[2] * BPP]], unit[row[(size_t)cc[3] * BPP]], fx);
			}
			double v = cr_exact(h[0], h[1], h[2], h[3], fy);
			px[off] = (uint8_t)round(__dmul_rn(clip01(v), 255.0));
		}
	}

	uint8_t *out = dst + (size_t)y * pitch + (size_t)x * BPP;
	out[0] = px[0]; out[1] = px[1]; out[2] = px[2];
	if (BPP == 4) out[3] = px[3];
	}
}

/* ------------------------------------------------------------------------ */
/* k_wide: the same fp64 arithmetic for 16/32-bit and floating-point samples  */
/* (get_pixel / set_pixel, fix-ca.c:713-774): bit-exact, one thread per pixel  */
/* ------------------------------------------------------------------------ */

template <typename S> struct SampleCodec;
template <> struct SampleCodec<uint16_t> {
	static __device__ __forceinline__ double get(uint16_t s) { return __ddiv_rn((double)s, 65535.0); }
	static __device__ __forceinline__ uint16_t put(double d) { return (uint16_t)round(__dmul_rn(d, 65535.0)); }
};
template <> struct SampleCodec<uint32_t> {
	static __device__ __forceinline__ double get(uint32_t s) { return __ddiv_rn((double)s, 4294967295.0); }
	static __device__ __forceinline__ uint32_t put(double d) { return (uint32_t)round(__dmul_rn(d, 4294967295.0)); }
};
template <> struct SampleCodec<float> {
	static __device__ __forceinline__ double get(float s) { return (double)s; }
	static __device__ __forceinline__ float put(double d) { return (float)d; }
};
template <> struct SampleCodec<double> {
	static __device__ __forceinline__ double get(double s) { return s; }
	static __device__ __forceinline__ double put(double d) { return d; }
};

template <typename S, int NCH>
__global__ __launch_bounds__(256) void
k_wide(const S *__restrict__ src, S *__restrict__ dst, FixcaGeom g)
{
	typedef SampleCodec<S> C;
	const int x = blockIdx.x * 256 + threadIdx.x;
	if (x >= g.W) return;
	const size_t pitch = (size_t)g.W * NCH;              /* in samples */

	for (int y = g.y1 + blockIdx.y; y < g.y2; y += gridDim.y) {
		const S *in = src + (size_t)y * pitch + (size_t)x * NCH;
		S px[NCH];
#pragma unroll
		for (int k = 0; k < NCH; ++k) px[k] = in[k];

#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const int off = 2 * c;
			const double s = g.s[c], xs = g.xs[c], ys = g.ys[c];
			if (g.interp == 0) {
				int sy = map_int(y, g.yc, g.H, s, ys);
				int sx = map_int(x, g.xc, g.W, s, xs);
				px[off] = src[(size_t)sy * pitch + (size_t)sx * NCH + off];   /* a copy of the sample's bytes */
			} else if (g.interp == 1) {
				double yd = map_dbl(y, g.yc, g.H, s, ys);
				double xd = map_dbl(x, g.xc, g.W, s, xs);
				int r0 = (int)floor(yd), c0 = (int)floor(xd);
				double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
				int r1 = (r0 == g.H - 1) ? r0 : r0 + 1;
				int c1 = (c0 == g.W - 1) ? c0 : c0 + 1;
				const S *row0 = src + (size_t)r0 * pitch + off;
				const S *row1 = src + (size_t)r1 * pitch + off;
				double p00 = C::get(row0[(size_t)c0 * NCH]), p10 = C::get(row0[(size_t)c1 * NCH]);
				double p01 = C::get(row1[(size_t)c0 * NCH]), p11 = C::get(row1[(size_t)c1 * NCH]);
				double a = __dadd_rn(p00, __dmul_rn(fx, __dsub_rn(p10, p00)));
				double b = __dadd_rn(p01, __dmul_rn(fx, __dsub_rn(p11, p01)));
				double v = __dadd_rn(__dmul_rn(__dsub_rn(1.0, fy), a), __dmul_rn(fy, b));
				px[off] = C::put(clip01(v));
			} else {
				double yd = map_dbl(y, g.yc, g.H, s, ys);
				double xd = map_dbl(x, g.xc, g.W, s, xs);
				int r0 = (int)floor(yd), c0 = (int)floor(xd);
				double fy = __dsub_rn(yd, (double)r0), fx = __dsub_rn(xd, (double)c0);
				int rr[4], cc[4];
				rr[1] = r0; rr[0] = (r0 == 0) ? r0 : r0 - 1;
				rr[2] = (r0 == g.H - 1) ? r0 : r0 + 1;
				rr[3] = (rr[2] == g.H - 1) ? rr[2] : rr[2] + 1;
				cc[1] = c0; cc[0] = (c0 == 0) ? c0 : c0 - 1;
				cc[2] = (c0 == g.W - 1) ? c0 : c0 + 1;
				cc[3] = (cc[2] == g.W - 1) ? cc[2] : cc[2] + 1;
				double h[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const S *row = src + (size_t)rr[j] * pitch + off;
					h[j] = cr_exact(C::get(row[(size_t)cc[0] * NCH]), C::get(row[(size_t)cc[1] * NCH]),
					                C::get(row[(size_t)cc[2] * NCH]), C::get(row[(size_t)cc[3] * NCH]), fx);
				}
				double v = cr_exact(h[0], h[1], h[2], h[3], fy);
				px[off] = C::put(clip01(v));
			}
		}

		S *out = dst + (size_t)y * pitch + (size_t)x * NCH;
#pragma unroll
		for (int k = 0; k < NCH; ++k) out[k] = px[k];
	}
}

template <typename S>
static void launch_wide(const void *d_src, void *d_dst, const FixcaGeom *g, dim3 grid, hipStream_t stream)
{
	const S *s = static_cast<const S *>(d_src);
	S *d = static_cast<S *>(d_dst);
	if (g->nch == 4) hipLaunchKernelGGL((k_wide<S, 4>), grid, dim3(256), 0, stream, s, d, *g);
	else             hipLaunchKernelGGL((k_wide<S, 3>), grid, dim3(256), 0, stream, s, d, *g);
}

/* ------------------------------------------------------------------------ */
/* shared pieces of the fast kernels                                         */
/* ------------------------------------------------------------------------ */

/* 4 pixels of an interleaved row, 16 B (RGBA) or 12 B (RGB); w[3] unused for RGB.
 * Pixel addresses are only 4-byte (RGBA) / 1-byte (RGB) aligned. */
struct Px4 { uint32_t w[4]; };

struct __attribute__((packed, aligned(4))) U4A4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) U3A1 { uint32_t a, b, c; };

template <int BPP>
__device__ __forceinline__ Px4 load_px4(const uint8_t *p)
{
	Px4 r;
	if (BPP == 4) {
		U4A4 v = *reinterpret_cast<const U4A4 *>(p);
		r.w[0] = v.a; r.w[1] = v.b; r.w[2] = v.c; r.w[3] = v.d;
	} else {
		U3A1 v = *reinterpret_cast<const U3A1 *>(p);
		r.w[0] = v.a; r.w[1] = v.b; r.w[2] = v.c; r.w[3] = 0;
	}
	return r;
}

template <int BPP>
__device__ __forceinline__ void store_px4(uint8_t *p, const Px4 &r)
{
	if (BPP == 4) {
		U4A4 v; v.a = r.w[0]; v.b = r.w[1]; v.c = r.w[2]; v.d = r.w[3];
		*reinterpret_cast<U4A4 *>(p) = v;
	} else {
		U3A1 v; v.a = r.w[0]; v.b = r.w[1]; v.c = r.w[2];
		*reinterpret_cast<U3A1 *>(p) = v;
	}
}

/*
 * Puts new red and blue bytes (4 pixels each, packed little-endian in rq / bq)
 * into 4 loaded pixels, keeping green (and alpha).
 * RGBA: pixel i is dword i = [R G B A].
 * RGB : 12 bytes [R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3].
 */
template <int BPP>
__device__ __forceinline__ Px4 merge_rb(const Px4 &idt, uint32_t rq, uint32_t bq)
{
	Px4 o;
	if (BPP == 4) {
		/* {hi = idt, lo = rq/bq}: take R from rq byte i, G from idt byte 1, B from bq, A from idt byte 3 */
		uint32_t rb0 = perm(bq, rq, 0x0c040c00);  /* [R0 0 B0 0] */
		uint32_t rb1 = perm(bq, rq, 0x0c050c01);
		uint32_t rb2 = perm(bq, rq, 0x0c060c02);
		uint32_t rb3 = perm(bq, rq, 0x0c070c03);
		o.w[0] = (idt.w[0] & 0xFF00FF00u) | rb0;
		o.w[1] = (idt.w[1] & 0xFF00FF00u) | rb1;
		o.w[2] = (idt.w[2] & 0xFF00FF00u) | rb2;
		o.w[3] = (idt.w[3] & 0xFF00FF00u) | rb3;
	} else {
		/* dword0 = [R0 G0 B0 R1]: R0 = rq.0, B0 = bq.0, R1 = rq.1 */
		uint32_t m0 = perm(bq, rq, 0x01040c00);   /* [R0 0 B0 R1] */
		/* dword1 = [G1 B1 R2 G2]: B1 = bq.1, R2 = rq.2 */
		uint32_t m1 = perm(bq, rq, 0x0c02050c);   /* [0 B1 R2 0] */
		/* dword2 = [B2 R3 G3 B3]: B2 = bq.2, R3 = rq.3, B3 = bq.3 */
		uint32_t m2 = perm(bq, rq, 0x070c0306);   /* [B2 R3 0 B3] */
		o.w[0] = (idt.w[0] & 0x0000FF00u) | m0;
		o.w[1] = (idt.w[1] & 0xFF0000FFu) | m1;
		o.w[2] = (idt.w[2] & 0x00FF0000u) | m2;
		o.w[3] = 0;
	}
	return o;
}

/* ------------------------------------------------------------------------ */
/* k_none: interpolation NONE, byte gather, bit-exact                        */
/* ------------------------------------------------------------------------ */

template <int BPP>
__global__ __launch_bounds__(256) void
k_none(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, FixcaGeom g, FixcaPlan pl)
{
	const int lane = threadIdx.x & (WAVE - 1);
	const int wid = xcd_remap(blockIdx.x, pl.nblocks) * 4 + (threadIdx.x >> 6);
	if (wid >= pl.nwaves) return;                       /* wave-uniform */
	const int strip = wid % pl.nstrips, chunk = wid / pl.nstrips;
	const int x = strip * pl.oc + LANE_PX * lane;
	const int Y0 = g.y1 + chunk * pl.th;
	const int nrows = min(pl.th, g.y2 - Y0);
	const size_t pitch = (size_t)g.W * BPP;
	const bool active = (LANE_PX * lane < pl.oc) && (x < g.W);
	const bool full = active && (x + 3 < g.W);

	/* per-lane column table: source column of each of my 4 pixels, as byte offsets */
	uint32_t sxo[2][4];
#pragma unroll
	for (int c = 0; c < 2; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			int xi = min(x + i, g.W - 1);
			sxo[c][i] = (uint32_t)map_int(xi, g.xc, g.W, g.s[c], g.xs[c]) * BPP + 2 * c;
		}

	/* per-wave row table: lane r holds the source rows of output row Y0 + r */
	int syt[2];
	{
		int yr = min(Y0 + lane, g.H - 1);
		syt[0] = map_int(yr, g.yc, g.H, g.s[0], g.ys[0]);
		syt[1] = map_int(yr, g.yc, g.H, g.s[1], g.ys[1]);
	}

	for (int r = 0; r < nrows; ++r) {
		const int y = Y0 + r;
		const uint8_t *rowR = src + (size_t)__builtin_amdgcn_readlane(syt[0], r) * pitch;
		const uint8_t *rowB = src + (size_t)__builtin_amdgcn_readlane(syt[1], r) * pitch;
		if (!active) continue;
		uint32_t rq = 0, bq = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			rq |= (uint32_t)rowR[sxo[0][i]] << (8 * i);
			bq |= (uint32_t)rowB[sxo[1][i]] << (8 * i);
		}
		const size_t o = (size_t)y * pitch + (size_t)x * BPP;
		if (full) {
			Px4 idt = load_px4<BPP>(src + o);
			store_px4<BPP>(dst + o, merge_rb<BPP>(idt, rq, bq));
		} else {
			for (int i = 0; i < 4 && x + i < g.W; ++i) {
				const uint8_t *in = src + o + (size_t)i * BPP;
				uint8_t *out = dst + o + (size_t)i * BPP;
				out[0] = (uint8_t)(rq >> (8 * i));
				out[1] = in[1];
				out[2] = (uint8_t)(bq >> (8 * i));
				if (BPP == 4) out[3] = in[3];
			}
		}
	}
}

/* ------------------------------------------------------------------------ */
/* k_interp: LINEAR (TAPS 2) / CUBIC (TAPS 4) row scan                       */
/* ------------------------------------------------------------------------ */

/* this channel's byte of 4 loaded pixels, permuted by the lane's edge-clamp selector */
template <int BPP>
__device__ __forceinline__ uint32_t channel_bytes(const Px4 &p, int c, uint32_t sel)
{
	if (BPP == 4) {
		/* byte c*2 of each dword */
		uint32_t pick = c ? 0x0c0c0602u : 0x0c0c0400u;
		uint32_t t01 = perm(p.w[1], p.w[0], pick);      /* [c0 c1 0 0] */
		uint32_t t23 = perm(p.w[3], p.w[2], pick);      /* [c2 c3 0 0] */
		return perm(t23, t01, sel);                     /* sel: 0,1 -> c0,c1; 4,5 -> c2,c3 */
	} else if (c == 0) {
		/* R at bytes 0,3,6,9: w0.0, w0.3, w1.2, w2.1 */
		uint32_t t = perm(p.w[1], p.w[0], 0x0c060300u); /* [R0 R1 R2 0] */
		return perm(p.w[2], t, sel);                    /* sel: 0,1,2 -> R0..R2; 5 -> R3 */
	} else {
		/* B at bytes 2,5,8,11: w0.2, w1.1, w2.0, w2.3 */
		uint32_t t = perm(p.w[1], p.w[0], 0x0c0c0502u); /* [B0 B1 0 0] */
		return perm(p.w[2], t, sel);                    /* sel: 0,1 -> B0,B1; 4 -> B2; 7 -> B3 */
	}
}

/* selector value that picks source pixel idx (0..3) in channel_bytes' final perm */
template <int BPP>
__device__ __forceinline__ uint32_t sel_code(int c, int idx)
{
	if (BPP == 4) return idx < 2 ? idx : idx + 2;
	if (c == 0) return idx < 3 ? idx : 5;
	return idx < 2 ? idx : (idx == 2 ? 4 : 7);
}

__device__ __forceinline__ float ubyte_f(uint32_t v, int k)
{
	return (float)((v >> (8 * k)) & 0xffu);             /* v_cvt_f32_ubyteK */
}

template <int BPP, int TAPS>
__global__ __launch_bounds__(256) void
k_interp(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, FixcaGeom g, FixcaPlan pl)
{
	/* per wave: the vertically filtered value of 256 source columns, per channel */
	__shared__ __attribute__((aligned(16))) float vx[4][2][FIXCA_WAVE_COLS];

	const int lane = threadIdx.x & (WAVE - 1);
	const int wib = threadIdx.x >> 6;
	const int wid = xcd_remap(blockIdx.x, pl.nblocks) * 4 + wib;
	if (wid >= pl.nwaves) return;                       /* wave-uniform */
	const int strip = wid % pl.nstrips, chunk = wid / pl.nstrips;
	const int X0 = strip * pl.oc;
	const int x = X0 + LANE_PX * lane;
	const int Y0 = g.y1 + chunk * pl.th;
	const int nrows = min(pl.th, g.y2 - Y0);
	const size_t pitch = (size_t)g.W * BPP;
	const bool active = (LANE_PX * lane < pl.oc) && (x < g.W);
	const bool full = active && (x + 3 < g.W);
	constexpr int BACK = (TAPS == 4) ? 1 : 0;           /* taps start at floor - BACK */

	/* ---- per-lane column constants ---------------------------------------- */
	uint32_t tapo[2][4];        /* LDS byte offset of the first tap of output pixel i */
	float    wx[2][4][TAPS];    /* horizontal weights                                  */
	uint32_t lbo[2];            /* byte offset in a row of this lane's 4 source pixels */
	uint32_t sel[2];            /* edge-clamp selector for channel_bytes               */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		const int S0 = (int)floor(map_dbl(min(X0, g.W - 1), g.xc, g.W, g.s[c], g.xs[c])) - BACK;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			int xi = min(x + i, g.W - 1);
			double xd = map_dbl(xi, g.xc, g.W, g.s[c], g.xs[c]);
			int x0 = (int)floor(xd);
			tap_weights<TAPS>(xd - (double)x0, wx[c][i]);
			int t = x0 - BACK - S0;
			t = max(0, min(t, FIXCA_WAVE_COLS - TAPS));     /* never leaves the wave's LDS slice */
			tapo[c][i] = (uint32_t)((((wib * 2 + c) * FIXCA_WAVE_COLS) + t) * 4);
		}
		const int sc = S0 + LANE_PX * lane;             /* first source column of this lane */
		const int lb = max(0, min(sc, g.W - 4));
		lbo[c] = (uint32_t)lb * BPP;
		uint32_t s4 = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			int idx = max(0, min(sc + k, g.W - 1)) - lb;    /* replicate the edge column */
			idx = max(0, min(idx, 3));
			s4 |= sel_code<BPP>(c, idx) << (8 * k);
		}
		sel[c] = s4;
	}

	/* ---- per-wave row table: lane r describes output row Y0 + r ------------ */
	int   rbase[2];             /* virtual source row of the first vertical tap        */
	float wy[2][TAPS];          /* vertical weights                                    */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		int yr = min(Y0 + lane, g.H - 1);
		double yd = map_dbl(yr, g.yc, g.H, g.s[c], g.ys[c]);
		int y0 = (int)floor(yd);
		tap_weights<TAPS>(yd - (double)y0, wy[c]);
		rbase[c] = y0 - BACK;
	}

	const uint8_t *srcc[2] = { src + lbo[0], src + lbo[1] };
	auto load_row = [&](int c, int v) -> Px4 {          /* virtual row v, edge rows replicated */
		int row = max(0, min(v, g.H - 1));
		return load_px4<BPP>(srcc[c] + (size_t)row * pitch);
	};

	/* ---- vertical window, in registers ------------------------------------- */
	uint32_t win[2][TAPS];      /* this channel's bytes of my 4 source columns, TAPS rows */
	int      vb[2];             /* virtual row held in win[c][0]                        */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		vb[c] = __builtin_amdgcn_readlane(rbase[c], 0);
#pragma unroll
		for (int j = 0; j < TAPS; ++j)
			win[c][j] = channel_bytes<BPP>(load_row(c, vb[c] + j), c, sel[c]);
	}

	float *const lds = &vx[0][0][0];
	const uint32_t myo = (uint32_t)((wib * 2 * FIXCA_WAVE_COLS) + LANE_PX * lane) * 4;

	for (int r = 0; r < nrows; ++r) {
		const int y = Y0 + r;

		/* prefetch the rows the next output row adds to each window */
		int adv[2] = { 0, 0 };
		Px4 pend[2][2];
		if (r + 1 < nrows) {
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				adv[c] = __builtin_amdgcn_readlane(rbase[c], r + 1) - vb[c];
				if (adv[c] >= 1 && adv[c] <= 2) pend[c][0] = load_row(c, vb[c] + TAPS);
				if (adv[c] == 2) pend[c][1] = load_row(c, vb[c] + TAPS + 1);
			}
		}

		const size_t o = (size_t)y * pitch + (size_t)x * BPP;
		Px4 idt;
		if (full) idt = load_px4<BPP>(src + o);

		/* vertical filter on my own 4 source columns, then publish to the wave */
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			float v[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
			for (int j = 0; j < TAPS; ++j) {
				float w = __builtin_bit_cast(float,
					__builtin_amdgcn_readlane(__builtin_bit_cast(int, wy[c][j]), r));
#pragma unroll
				for (int k = 0; k < 4; ++k)
					v[k] = fmaf(w, ubyte_f(win[c][j], k), v[k]);
			}
			float4 q = make_float4(v[0], v[1], v[2], v[3]);
			*reinterpret_cast<float4 *>(reinterpret_cast<char *>(lds) + myo + c * FIXCA_WAVE_COLS * 4) = q;
		}
		/* LDS operations of one wave complete in order; only the compiler needs the fence */
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		/* horizontal taps of my 4 output pixels */
		uint32_t q8[2] = { 0, 0 };
#pragma unroll
		for (int c = 0; c < 2; ++c) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const float *t = reinterpret_cast<const float *>(reinterpret_cast<const char *>(lds) + tapo[c][i]);
				float acc = 0.f;
#pragma unroll
				for (int j = 0; j < TAPS; ++j)
					acc = fmaf(wx[c][i][j], t[j], acc);
				acc = fminf(fmaxf(acc, 0.f), 255.f) + 0.5f;     /* clip, then round half up */
				q8[c] |= (uint32_t)acc << (8 * i);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();

		if (full) {
			store_px4<BPP>(dst + o, merge_rb<BPP>(idt, q8[0], q8[1]));
		} else if (active) {
			for (int i = 0; i < 4 && x + i < g.W; ++i) {
				const uint8_t *in = src + o + (size_t)i * BPP;
				uint8_t *out = dst + o + (size_t)i * BPP;
				out[0] = (uint8_t)(q8[0] >> (8 * i));
				out[1] = in[1];
				out[2] = (uint8_t)(q8[1] >> (8 * i));
				if (BPP == 4) out[3] = in[3];
			}
		}

		/* roll the windows forward to the next output row */
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			if (adv[c] == 1 || adv[c] == 2) {
#pragma unroll
				for (int j = 0; j + 1 < TAPS; ++j) win[c][j] = win[c][j + 1];
				win[c][TAPS - 1] = channel_bytes<BPP>(pend[c][0], c, sel[c]);
				if (adv[c] == 2) {
#pragma unroll
					for (int j = 0; j + 1 < TAPS; ++j) win[c][j] = win[c][j + 1];
					win[c][TAPS - 1] = channel_bytes<BPP>(pend[c][1], c, sel[c]);
				}
				vb[c] += adv[c];
			} else if (adv[c] != 0) {                   /* scale > 2: reload the whole window */
				vb[c] += adv[c];
#pragma unroll
				for (int j = 0; j < TAPS; ++j)
					win[c][j] = channel_bytes<BPP>(load_row(c, vb[c] + j), c, sel[c]);
			}
		}
	}
}

/* ------------------------------------------------------------------------ */
/* k_centerline: the preview's dashed lens-centre lines (fix-ca.c:945-996),   */
/* drawn over rows [y1,y2) of the result.  One workgroup per row.             */
/* ------------------------------------------------------------------------ */

template <typename S>
__global__ __launch_bounds__(256) void
k_centerline(S *__restrict__ dst, FixcaGeom g, S one)
{
	const int y = g.y1 + blockIdx.x;
	const int W = g.W, nch = g.nch;
	S *row = dst + (size_t)y * W * nch;
	if (y == g.yc) {
		/* the reference walks the row with a counter: the first run is i0 + 2 pixels long, every
		 * later run 9 pixels, alternating 0 and 1 (i = |xc| % 16; starts at 0 if i < 8) */
		const int i0 = abs(g.xc) % 16;
		const int c0 = (i0 < 8) ? 0 : 1;
		for (int x = threadIdx.x; x < W; x += 256) {
			const int c = (x <= i0 + 1) ? c0 : (c0 ^ ((((x - (i0 + 2)) / 9) + 1) & 1));
			const S v = c ? one : (S)0;
			row[(size_t)x * nch] = v; row[(size_t)x * nch + 1] = v; row[(size_t)x * nch + 2] = v;
		}
		return;
	}
	if (threadIdx.x < 3) {
		const int d = abs(y - g.yc);
		const S v = ((d % 16) < 8) ? (S)0 : one;
		const int x = threadIdx.x == 0 ? g.xc : (threadIdx.x == 1 ? g.xc - d : g.xc + d);
		if (x >= 0 && x < W) {
			row[(size_t)x * nch] = v; row[(size_t)x * nch + 1] = v; row[(size_t)x * nch + 2] = v;
		}
	}
}

extern "C" hipError_t
fixca_launch_centerline(void *d_dst, const FixcaGeom *g, hipStream_t stream)
{
	const int rows = g->y2 - g->y1;
	if (rows <= 0) return hipSuccess;
	const dim3 grid(rows), block(256);
	switch (g->bpc) {
	case 1:  hipLaunchKernelGGL(k_centerline<uint8_t>, grid, block, 0, stream, static_cast<uint8_t *>(d_dst), *g, (uint8_t)255); break;
	case 2:  hipLaunchKernelGGL(k_centerline<uint16_t>, grid, block, 0, stream, static_cast<uint16_t *>(d_dst), *g, (uint16_t)65535); break;
	case 4:  hipLaunchKernelGGL(k_centerline<uint32_t>, grid, block, 0, stream, static_cast<uint32_t *>(d_dst), *g, 4294967295u); break;
	case -4: hipLaunchKernelGGL(k_centerline<float>, grid, block, 0, stream, static_cast<float *>(d_dst), *g, 1.0f); break;
	case -8: hipLaunchKernelGGL(k_centerline<double>, grid, block, 0, stream, static_cast<double *>(d_dst), *g, 1.0); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* k_preview: the tail of preview_update() (fix-ca.c:659-671): the visible      */
/* window of the corrected image, samples wider than 8 bits converted with      */
/* get_pixel -> set_pixel(.., 1).  One thread per output sample.                */
/* ------------------------------------------------------------------------ */

/* (guchar) round(d * 255) as gcc/x86-64 evaluates it (fix-ca.c:752): in range of a 32-bit integer -> its
 * low byte, otherwise (or NaN) 0x80000000 -> 0.  In-range values 0..255 are what every image in
 * [0,1] produces; the rest only pins down what the reference does with out-of-range floats. */
__device__ __forceinline__ uint8_t u8_as_x86(double r)
{
	const int i = (r >= -2147483648.0 && r < 2147483648.0) ? (int)r : (-2147483647 - 1);
	return (uint8_t)((uint32_t)i & 0xffu);
}

template <typename S>
__global__ __launch_bounds__(256) void
k_preview(const S *__restrict__ img, uint8_t *__restrict__ prev, int W, int nch, int x, int y, int width, int height)
{
	const size_t n = (size_t)width * nch;           /* samples per window row */
	const int i = blockIdx.y;
	const S *row = img + ((size_t)W * (y + i) + x) * nch;
	for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t)gridDim.x * 256)
		prev[(size_t)i * n + j] = u8_as_x86(round(__dmul_rn(SampleCodec<S>::get(row[j]), 255.0)));
}

extern "C" hipError_t
fixca_launch_preview(const void *d_img, void *d_prev, int W, int bytes, int bpc, int x, int y, int width, int height, hipStream_t stream)
{
	if (width <= 0 || height <= 0) return hipSuccess;
	const int b = bpc < 0 ? -bpc : bpc, nch = bytes / b;
	if (b == 1)                                      /* a plain crop (fix-ca.c:661-664) */
		return hipMemcpy2DAsync(d_prev, (size_t)width * bytes, static_cast<const uint8_t *>(d_img) + ((size_t)W * y + x) * bytes,
		                        (size_t)W * bytes, (size_t)width * bytes, height, hipMemcpyDeviceToDevice, stream);
	const size_t n = (size_t)width * nch;
	const dim3 grid((unsigned)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), (unsigned)(height < 65535 ? height : 65535)), block(256);
	if (height > 65535) return hipErrorInvalidValue;         /* a preview window is a few hundred rows */
	uint8_t *prev = static_cast<uint8_t *>(d_prev);
	switch (bpc) {
	case 2:  hipLaunchKernelGGL(k_preview<uint16_t>, grid, block, 0, stream, static_cast<const uint16_t *>(d_img), prev, W, nch, x, y, width, height); break;
	case 4:  hipLaunchKernelGGL(k_preview<uint32_t>, grid, block, 0, stream, static_cast<const uint32_t *>(d_img), prev, W, nch, x, y, width, height); break;
	case -4: hipLaunchKernelGGL(k_preview<float>, grid, block, 0, stream, static_cast<const float *>(d_img), prev, W, nch, x, y, width, height); break;
	case -8: hipLaunchKernelGGL(k_preview<double>, grid, block, 0, stream, static_cast<const double *>(d_img), prev, W, nch, x, y, width, height); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* bandwidth yardstick: 16 bytes per lane straight copy                      */
/* ------------------------------------------------------------------------ */

__global__ __launch_bounds__(256) void
k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	typedef unsigned int v4u __attribute__((ext_vector_type(4)));
	const v4u *s = reinterpret_cast<const v4u *>(src);
	v4u *d = reinterpret_cast<v4u *>(dst);
	/* nt both ways: the fastest plain copy measured (tools/kbench.hip) */
	for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&s[i]), &d[i]);
}

/* SURVEY §8d's synthetic input on the device: byte i of noise(seed) is byte (i mod 8) of SplitMix64
 * word i/8 + 1 = mix(seed + (i/8 + 1) * GOLDEN), little-endian: the same bytes as
 * fixca/synth.py noise_bytes (tests/test_gpu_parity.py compares them). */
__global__ __launch_bounds__(256) void
k_noise(uint8_t *__restrict__ dst, size_t nbytes, unsigned long long seed)
{
	const size_t nwords = (nbytes + 7) / 8;
	for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (size_t)gridDim.x * 256) {
		unsigned long long z = seed + (unsigned long long)(w + 1) * 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		if (8 * w + 8 <= nbytes) {
			*reinterpret_cast<unsigned long long *>(dst + 8 * w) = z;         /* buffers from hipMalloc are 256-byte aligned */
		} else {
			for (size_t b = 8 * w; b < nbytes; ++b) dst[b] = (uint8_t)(z >> (8 * (b - 8 * w)));
		}
	}
}

extern "C" hipError_t fixca_launch_noise(void *d_dst, size_t nbytes, unsigned long long seed, hipStream_t stream)
{
	if (!nbytes) return hipSuccess;
	if (reinterpret_cast<uintptr_t>(d_dst) & 7) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_noise, dim3(8192), dim3(256), 0, stream, static_cast<uint8_t *>(d_dst), nbytes, seed);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* launchers (called from fixca_abi.hip)                                     */
/* ------------------------------------------------------------------------ */

extern "C" hipError_t fixca_launch_scan(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_roll(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_roll_rgb(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_wide(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_move(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_lin(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);
extern "C" hipError_t fixca_launch_cub(const void *, void *, const FixcaGeom *, const FixcaPlan *, hipStream_t);

extern "C" hipError_t
fixca_launch(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	const uint8_t *s = static_cast<const uint8_t *>(d_src);
	uint8_t *d = static_cast<uint8_t *>(d_dst);
	if (g->y2 <= g->y1) return hipSuccess;
	switch (pl->kernel) {
	case FIXCA_K_EXACT: {
		int rows = g->y2 - g->y1;
		dim3 grid((g->W + 255) / 256, rows < 65535 ? rows : 65535);
		if (g->bpc == 2)       launch_wide<uint16_t>(d_src, d_dst, g, grid, stream);
		else if (g->bpc == 4)  launch_wide<uint32_t>(d_src, d_dst, g, grid, stream);
		else if (g->bpc == -4) launch_wide<float>(d_src, d_dst, g, grid, stream);
		else if (g->bpc == -8) launch_wide<double>(d_src, d_dst, g, grid, stream);
		else if (g->bpp == 4)  hipLaunchKernelGGL(k_exact<4>, grid, dim3(256), 0, stream, s, d, *g);
		else                   hipLaunchKernelGGL(k_exact<3>, grid, dim3(256), 0, stream, s, d, *g);
		break;
	}
	case FIXCA_K_NONE:
		if (g->bpp == 4) hipLaunchKernelGGL(k_none<4>, dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
		else             hipLaunchKernelGGL(k_none<3>, dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
		break;
	case FIXCA_K_INTERP:
		if (g->interp == 1) {
			if (g->bpp == 4) hipLaunchKernelGGL((k_interp<4, 2>), dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
			else             hipLaunchKernelGGL((k_interp<3, 2>), dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
		} else {
			if (g->bpp == 4) hipLaunchKernelGGL((k_interp<4, 4>), dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
			else             hipLaunchKernelGGL((k_interp<3, 4>), dim3(pl->nblocks), dim3(256), 0, stream, s, d, *g, *pl);
		}
		break;
	case FIXCA_K_SCAN:
		return fixca_launch_scan(d_src, d_dst, g, pl, stream);
	case FIXCA_K_WCOL:
		return fixca_launch_wide(d_src, d_dst, g, pl, stream);
	case FIXCA_K_MOVE:
		return fixca_launch_move(d_src, d_dst, g, pl, stream);
	case FIXCA_K_LIN:
		return fixca_launch_lin(d_src, d_dst, g, pl, stream);
	case FIXCA_K_CUB:
		return fixca_launch_cub(d_src, d_dst, g, pl, stream);
	case FIXCA_K_ROLL:
		return g->bpp == 3 ? fixca_launch_roll_rgb(d_src, d_dst, g, pl, stream) : fixca_launch_roll(d_src, d_dst, g, pl, stream);
	default:
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

extern "C" hipError_t
fixca_launch_copy(const void *d_src, void *d_dst, size_t nbytes, hipStream_t stream)
{
	size_t n16 = nbytes / 16;
	hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, stream,
	                   static_cast<const uint4 *>(d_src), static_cast<uint4 *>(d_dst), n16);
	return hipGetLastError();
}
