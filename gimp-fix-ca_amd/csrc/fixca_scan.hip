/*
 * fixca_scan.hip — the LDS-ring row-scan kernel (RGBA8): the planner's second choice, for 8-bit geometries the
 * row-rolling kernel (fixca_roll.hip, the main kernel of the path) refuses: tap offsets that drift by more than
 * one column over a strip (scales far from 1).
 *
 * One wave (a 64-thread workgroup) owns `oc` output columns x `th` output rows and
 * streams the source rows it needs exactly once:
 *
 *   HBM/L2 --LDS-DMA--> ring of K source rows in LDS --> registers --> HBM
 *
 *  - global_load_lds_dwordx4: every lane moves its 4 pixels (16 B) of a row
 *    segment of 256 columns straight into LDS, no VGPRs, up to K - span rows ahead
 *    of the row being produced (span = rows the red window, the blue window and
 *    the identity copy keep alive, 8 for +-2 px).  The wave waits with a counted
 *    s_waitcnt vmcnt(n) for just the row it needs; younger loads and the stores
 *    stay in flight.
 *  - The one 256-column window [U0, U0+256) serves all three consumers of a
 *    source row: the red taps, the blue taps and the green/alpha copy, so L2 and
 *    the fabric see each row segment once per wave.
 *  - Per-row values (source rows, vertical weights, ring bounds) are computed
 *    once per wave, one output row per lane, and broadcast with v_readlane;
 *    per-column values (tap positions, horizontal weights) are lane constants.
 *  - LINEAR/CUBIC: the vertical window of each channel lives in registers as
 *    floats in a rotating set of TAPS slots (the row loop is unrolled by TAPS, so
 *    advancing by one source row costs one LDS read and 4 conversions and no
 *    moves); the vertical filter runs on the lane's own 4 columns for red and blue
 *    together (packed fp32), the 256 results per channel cross the wave through a
 *    bank-conflict-free transposed LDS array, and each lane gathers the horizontal
 *    taps of its 4 outputs.  fp32 sample arithmetic on fp64 coordinates: within 1 LSB
 *    of the reference (fix-ca.c:882-920), green/alpha bit-exact.
 *  - NONE: a byte gather out of the ring with the reference's exact fp64 index
 *    arithmetic (fix-ca.c:799-809): bit-exact.
 *
 * The planner (fixca_plan.cpp) only selects this kernel when W % 4 == 0, W >= 256,
 * every strip's union window fits 256 columns and span + 2 <= K; everything else
 * goes to the register-window kernel or the fp64 kernel in fixca_kernels.hip.
 */
#include "fixca_device.h"
#include "fixca_internal.h"
#include <mutex>
#include <type_traits>

typedef float        f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define RING_ROW   1024                 /* bytes per ring row: 256 px x 4 B             */
#define VX_PS      66                   /* floats per plane of the exchange array        */
#define VX_CH      (4 * VX_PS)          /* floats per channel: index n = 4m+q at [q][m+1] */

template <int N> __device__ __forceinline__ void wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

/* v_cvt_pk_u8_f32 rounds to nearest EVEN; the reference rounds half away from zero
 * (set_pixel, fix-ca.c:752) and on exact .5 ties its fp64 value lands on or above .5 about
 * nine times in ten, so ties must go up: a bias of 2^-13 (8 fp32 ulps at 255) does that. */
#define ROUND_BIAS 0.0001220703125f

#define DEPTH 6                         /* source rows kept in flight ahead of the newest row in use */

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(lo, min(v, hi)); }

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rlf(float v, int lane)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

/* byte address of exchange index n (may be -1 .. 257) inside one channel's array */
__device__ __forceinline__ uint32_t vx_addr(int n)
{
	return (uint32_t)(((n & 3) * VX_PS + (n >> 2) + 1) * 4);
}

/* everything a wave carries from one output row to the next */
template <int TAPS>
struct ScanState {
	static constexpr int T = (TAPS == 0) ? 1 : TAPS;
	static constexpr int NT = T + 3;
	static constexpr int NH = (T > 1) ? T - 1 : 1;
	/* lane constants */
	f32x2    wx[4][T];          /* (red, blue) horizontal weights of output i, tap t          */
	uint32_t ta[2][NT];         /* LDS address of tap position t (consecutive-tap lanes)      */
	int      nb[2][4];          /* exchange index of the first tap of output i                */
	uint32_t go[2][4];          /* NONE: byte offset of the source pixel inside a ring row     */
	uint32_t my_ring, my_idt, my_vx;
	bool     active, allcons, edgeL, edgeR;
	int      lane;
	/* row table: lane j describes output row Y0 + 64*blk + j */
	int      t_rb[2], t_keep, t_need;
	float    t_wy[2][T];
	unsigned long long regmask; /* bit j: both windows of row j slide down by exactly one unclamped row */
	/* ring */
	uint8_t *ring, *vx;
	int      KM, H;
	int      ld;                /* highest source row issued                                   */
	int      lastneed;          /* highest source row this wave ever needs                     */
	int      seq, seqtab;       /* loads issued; lane s: seq of the load in slot s              */
	const uint8_t *gld;         /* global address of my 16 bytes of source row ld + 1          */
	uint8_t *gout;              /* global address of my 16 bytes of the next output row        */
	size_t   pitch;
	/* the newest T-1 source rows of each window, (red, blue) floats of my 4 columns */
	f32x2    hist[NH][4];
};

template <int TAPS>
__device__ __forceinline__ void scan_issue(ScanState<TAPS> &st)
{
	++st.ld;
	const int slot = st.ld & st.KM;
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)st.gld,
	                                 (__attribute__((address_space(3))) void *)(st.ring + slot * RING_ROW), 16, 0, 0);
	st.gld += st.pitch;
	st.seqtab = (st.lane == slot) ? st.seq : st.seqtab;     /* v_writelane by hand */
	++st.seq;
}

/* keep DEPTH rows in flight beyond `need` (within the ring's capacity), then wait until row
 * `need` has landed.
 * vmcnt counts loads and stores together, and on gfx9-family hardware loads complete in order
 * among loads and stores among stores, but NOT across the two kinds (a young store may be
 * acknowledged before an old load has landed; seen on MI355X as partially landed ring rows).
 * So the wait may only rely on loads: with L loads younger than the one needed, "at most L
 * operations outstanding" implies "at most L loads outstanding", which implies the needed
 * load is done.  seq therefore counts loads only, and stores in flight merely make the wait
 * stricter, never wrong. */
template <int TAPS>
__device__ __forceinline__ void scan_fetch(ScanState<TAPS> &st, int keep, int need)
{
	const int target = min(min(need + DEPTH, keep + st.KM), st.lastneed);
	while (st.ld < target) scan_issue(st);
	const int younger = st.seq - rl(st.seqtab, need & st.KM) - 1;
	if (younger >= 6) wait_vm<6>();
	else if (younger >= 4) wait_vm<4>();
	else if (younger >= 2) wait_vm<2>();
	else wait_vm<0>();
}

template <int TAPS>
__device__ __forceinline__ u32x4 scan_ring_px(const ScanState<TAPS> &st, int row, uint32_t off)
{
	return *reinterpret_cast<const u32x4 *>(st.ring + (row & st.KM) * RING_ROW + off);
}

__device__ __forceinline__ f32x2 rb_floats(uint32_t pr, uint32_t pb)
{
	f32x2 f = { (float)(pr & 0xffu), (float)((pb >> 16) & 0xffu) };   /* v_cvt_f32_ubyte0 / ubyte2 */
	return f;
}

/* publish one row of vertically filtered values: exchange index 4*lane + k -> plane k, entry lane + 1 */
template <int TAPS>
__device__ __forceinline__ void scan_publish(const ScanState<TAPS> &st, const f32x2 v[4])
{
	float *const vr = reinterpret_cast<float *>(st.vx + st.my_vx);
	float *const vb = reinterpret_cast<float *>(st.vx + st.my_vx + VX_CH * 4);
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		vr[k * VX_PS] = v[k].x;
		vb[k * VX_PS] = v[k].y;
	}
	if (st.edgeL && st.lane == 0) {                 /* column -1 replicates column 0 */
		vr[3 * VX_PS - 1] = v[0].x;
		vb[3 * VX_PS - 1] = v[0].y;
	}
	if (st.edgeR && st.lane == 63) {                /* columns W, W+1 replicate column W-1 */
		vr[0 * VX_PS + 1] = v[3].x; vr[1 * VX_PS + 1] = v[3].x;
		vb[0 * VX_PS + 1] = v[3].y; vb[1 * VX_PS + 1] = v[3].y;
	}
	/* LDS operations of one wave execute in order; only the compiler must not reorder them */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* gather the horizontal taps of my 4 outputs and filter */
template <int TAPS>
__device__ __forceinline__ void scan_gather(const ScanState<TAPS> &st, f32x2 acc[4])
{
	constexpr int NT = ScanState<TAPS>::NT;
#pragma unroll
	for (int i = 0; i < 4; ++i) acc[i] = f32x2{ ROUND_BIAS, ROUND_BIAS };
	if (st.allcons) {
		f32x2 tp[NT];
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			tp[t].x = *reinterpret_cast<const float *>(st.vx + st.ta[0][t]);
			tp[t].y = *reinterpret_cast<const float *>(st.vx + st.ta[1][t]);
		}
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int t = 0; t < TAPS; ++t)
				acc[i] = __builtin_elementwise_fma(st.wx[i][t], tp[i + t], acc[i]);
	} else {
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int t = 0; t < TAPS; ++t) {
				f32x2 tp;
				tp.x = *reinterpret_cast<const float *>(st.vx + vx_addr(st.nb[0][i] + t));
				tp.y = *reinterpret_cast<const float *>(st.vx + vx_addr(st.nb[1][i] + t) + VX_CH * 4);
				acc[i] = __builtin_elementwise_fma(st.wx[i][t], tp, acc[i]);
			}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

/* saturating round-to-nearest conversion straight into the red and blue bytes, then the row store */
template <int TAPS>
__device__ __forceinline__ void scan_store(ScanState<TAPS> &st, const f32x2 acc[4], const u32x4 &idt)
{
	u32x4 out;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].x, 0, idt[k]);
		out[k] = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].y, 2, o);
	}
	if (st.active)
		__builtin_nontemporal_store(out, reinterpret_cast<u32x4 *>(st.gout));
	st.gout += st.pitch;
	/* stores are NOT counted in seq: see the fetch function */
}

/* NONE: NR output rows, each a byte gather out of the ring */
template <int NR>
__device__ __forceinline__ void scan_rows_none(ScanState<0> &st, int r, int y)
{
	const int j = r & 63;
	scan_fetch(st, rl(st.t_keep, j), rl(st.t_need, j + NR - 1));
	u32x4 idt[NR];
	uint32_t rq[NR], bq[NR];
#pragma unroll
	for (int n = 0; n < NR; ++n) {
		idt[n] = scan_ring_px(st, y + n, st.my_idt);
		const uint8_t *rowR = st.ring + (rl(st.t_rb[0], j + n) & st.KM) * RING_ROW;
		const uint8_t *rowB = st.ring + (rl(st.t_rb[1], j + n) & st.KM) * RING_ROW;
		rq[n] = bq[n] = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			rq[n] |= (uint32_t)rowR[st.go[0][i]] << (8 * i);
			bq[n] |= (uint32_t)rowB[st.go[1][i]] << (8 * i);
		}
	}
#pragma unroll
	for (int n = 0; n < NR; ++n) {
		u32x4 out;
		out.x = (idt[n].x & 0xFF00FF00u) | perm(bq[n], rq[n], 0x0c040c00);
		out.y = (idt[n].y & 0xFF00FF00u) | perm(bq[n], rq[n], 0x0c050c01);
		out.z = (idt[n].z & 0xFF00FF00u) | perm(bq[n], rq[n], 0x0c060c02);
		out.w = (idt[n].w & 0xFF00FF00u) | perm(bq[n], rq[n], 0x0c070c03);
		if (st.active)
			__builtin_nontemporal_store(out, reinterpret_cast<u32x4 *>(st.gout));
		st.gout += st.pitch;
	}
}

/* LINEAR / CUBIC: one output row, windows rebuilt from the ring (first row of a wave, scale
 * steps, rows whose window touches the top or bottom edge) */
template <int TAPS>
__device__ __forceinline__ void scan_row_any(ScanState<TAPS> &st, int r, int y)
{
	constexpr int T = TAPS;
	const int j = r & 63;
	scan_fetch(st, rl(st.t_keep, j), rl(st.t_need, j));
	const u32x4 idt = scan_ring_px(st, y, st.my_idt);
	const int rbR = rl(st.t_rb[0], j), rbB = rl(st.t_rb[1], j);
	f32x2 w[T][4];
#pragma unroll
	for (int t = 0; t < T; ++t) {
		const u32x4 pr = scan_ring_px(st, clampi(rbR + t, 0, st.H - 1), st.my_ring);   /* edge rows replicate */
		const u32x4 pb = scan_ring_px(st, clampi(rbB + t, 0, st.H - 1), st.my_ring);
#pragma unroll
		for (int k = 0; k < 4; ++k) w[t][k] = rb_floats(pr[k], pb[k]);
	}
	f32x2 v[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
#pragma unroll
	for (int t = 0; t < T; ++t) {
		const f32x2 wy = { rlf(st.t_wy[0][t], j), rlf(st.t_wy[1][t], j) };
#pragma unroll
		for (int k = 0; k < 4; ++k) v[k] = __builtin_elementwise_fma(wy, w[t][k], v[k]);
	}
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) st.hist[t][k] = w[t + 1][k];
	f32x2 acc[4];
	scan_publish(st, v);
	scan_gather(st, acc);
	scan_store(st, acc, idt);
}

/* LINEAR / CUBIC: 4 consecutive output rows whose windows each slide down by one unclamped
 * source row.  All LDS traffic of the 4 rows is issued together so one wave overlaps its own
 * latencies: 12 ring reads, then 4 x (publish, gather), then 4 stores. */
template <int TAPS>
__device__ __forceinline__ void scan_block4(ScanState<TAPS> &st, int r, int y)
{
	constexpr int T = TAPS;
	const int j = r & 63;
	scan_fetch(st, rl(st.t_keep, j), rl(st.t_need, j + 3));
	const int nR = rl(st.t_rb[0], j) + T - 1, nB = rl(st.t_rb[1], j) + T - 1;   /* newest row of output row r */

	u32x4 idt[4];
	f32x2 rows[T - 1 + 4][4];                       /* hist rows, then the 4 new rows */
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) rows[t][k] = st.hist[t][k];
#pragma unroll
	for (int n = 0; n < 4; ++n) {
		idt[n] = scan_ring_px(st, y + n, st.my_idt);
		const u32x4 pr = scan_ring_px(st, nR + n, st.my_ring);
		const u32x4 pb = scan_ring_px(st, nB + n, st.my_ring);
#pragma unroll
		for (int k = 0; k < 4; ++k) rows[T - 1 + n][k] = rb_floats(pr[k], pb[k]);
	}
	f32x2 acc[4][4];
#pragma unroll
	for (int n = 0; n < 4; ++n) {
		f32x2 v[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
#pragma unroll
		for (int t = 0; t < T; ++t) {
			const f32x2 wy = { rlf(st.t_wy[0][t], j + n), rlf(st.t_wy[1][t], j + n) };
#pragma unroll
			for (int k = 0; k < 4; ++k) v[k] = __builtin_elementwise_fma(wy, rows[n + t][k], v[k]);
		}
		scan_publish(st, v);
		scan_gather(st, acc[n]);
	}
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) st.hist[t][k] = rows[4 + t][k];
#pragma unroll
	for (int n = 0; n < 4; ++n) scan_store(st, acc[n], idt[n]);
}

template <int TAPS>          /* 0 = NONE, 2 = LINEAR, 4 = CUBIC */
__global__ __launch_bounds__(64) void
k_scan(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, FixcaGeom g, FixcaPlan pl)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	constexpr int T = ScanState<TAPS>::T;
	constexpr int NT = ScanState<TAPS>::NT;
	constexpr int BACK = (TAPS == 4) ? 1 : 0;

	const int lane = threadIdx.x;
	const int tile = xcd_remap(blockIdx.x, pl.nblocks);
	if (tile >= pl.nwaves) return;
	const int strip = tile % pl.nstrips, chunk = tile / pl.nstrips;
	const int W = g.W, H = g.H, oc = pl.oc;
	int X0 = strip * oc;
	if (X0 > W - oc) X0 = W - oc;                   /* the last strip slides left: every strip is full */
	const int Y0 = g.y1 + chunk * pl.th;
	const int nrows = min(pl.th, g.y2 - Y0);

	ScanState<TAPS> st;
	st.lane = lane;
	st.KM = pl.ring - 1;                            /* ring rows: power of two */
	st.H = H;
	st.pitch = (size_t)W * 4;
	st.ring = lds;
	st.vx = lds + (size_t)pl.ring * RING_ROW;       /* [2][VX_CH] floats */

	/* ---- the 256-column source window of this strip (wave-uniform) ------------------- */
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		int a;
		if constexpr (TAPS == 0) a = map_int(X0, g.xc, W, g.s[c], g.xs[c]);
		else a = clampi((int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])) - BACK, 0, W - 1);
		lo = min(lo, a);
	}
	const int U0 = __builtin_amdgcn_readfirstlane(min(lo & ~3, W - FIXCA_WAVE_COLS));
	st.edgeL = (U0 == 0);
	st.edgeR = (U0 + FIXCA_WAVE_COLS == W);

	/* ---- row mapping (shared by the row table and the bounds below) ------------------- */
	auto row_of = [&](int y, int c, int &rb, float *wy) {
		if constexpr (TAPS == 0) {
			rb = map_int(y, g.yc, H, g.s[c], g.ys[c]);
		} else {
			double yd = map_dbl(y, g.yc, H, g.s[c], g.ys[c]);
			int y0 = (int)floor(yd);
			if (wy) tap_weights<T>(yd - (double)y0, wy);
			rb = y0 - BACK;
		}
	};
	auto build_rows = [&](int rbase) {               /* lane j <- output row Y0 + rbase + j */
		const int yr = min(Y0 + rbase + lane, H - 1);
		int keep = yr, need = yr;
		bool reg = (rbase + lane) > 0;
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			row_of(yr, c, st.t_rb[c], st.t_wy[c]);
			int prev;
			row_of(max(yr - 1, 0), c, prev, nullptr);
			reg = reg && (st.t_rb[c] == prev + 1) && (st.t_rb[c] >= 0) && (st.t_rb[c] + T - 1 <= H - 1);
			keep = min(keep, clampi(st.t_rb[c], 0, H - 1));
			need = max(need, clampi(st.t_rb[c] + T - 1, 0, H - 1));
		}
		st.t_keep = keep;
		st.t_need = need;
		st.regmask = __ballot(reg);
	};

	/* the last source row this wave will ever need */
	{
		const int yl = Y0 + nrows - 1;
		int ln = yl;
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			int rb;
			row_of(yl, c, rb, nullptr);
			ln = max(ln, clampi(rb + T - 1, 0, H - 1));
		}
		st.lastneed = __builtin_amdgcn_readfirstlane(ln);
	}

	build_rows(0);
	st.ld = rl(st.t_keep, 0) - 1;
	st.seq = 0;
	st.seqtab = 0;
	st.gld = src + (size_t)(st.ld + 1) * st.pitch + (size_t)(U0 + 4 * lane) * 4;
	/* start filling the ring before the (fp64-heavy) lane constants are computed */
	{
		const int keep0 = rl(st.t_keep, 0), need0 = rl(st.t_need, 0);
		const int target = min(min(need0 + DEPTH, keep0 + st.KM), st.lastneed);
		while (st.ld < target) scan_issue(st);
	}

	/* ---- per-lane column constants ------------------------------------------------ */
	const int nact = oc >> 2;                        /* lanes that produce output */
	st.active = lane < nact;
	const int xl = X0 + 4 * min(lane, nact - 1);     /* first output column of this lane */
	bool cons = true;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if constexpr (TAPS == 0) {
				int sx = map_int(xl + i, g.xc, W, g.s[c], g.xs[c]);
				st.go[c][i] = (uint32_t)clampi(sx - U0, 0, FIXCA_WAVE_COLS - 1) * 4 + 2 * c;
			} else {
				double xd = map_dbl(xl + i, g.xc, W, g.s[c], g.xs[c]);
				int x0 = (int)floor(xd);
				float w[T];
				tap_weights<T>(xd - (double)x0, w);
#pragma unroll
				for (int t = 0; t < T; ++t) {
					if (c == 0) st.wx[i][t].x = w[t]; else st.wx[i][t].y = w[t];
				}
				st.nb[c][i] = clampi(x0 - BACK - U0, -1, FIXCA_WAVE_COLS + 2 - T);
				cons = cons && (st.nb[c][i] == st.nb[c][0] + i);
			}
		}
		if constexpr (TAPS != 0) {
#pragma unroll
			for (int t = 0; t < NT; ++t)
				st.ta[c][t] = vx_addr(st.nb[c][0] + t) + c * VX_CH * 4;
		}
	}
	st.allcons = (TAPS == 0) ? true : (__all(cons) != 0);
	st.my_ring = (uint32_t)lane * 16;                                   /* my 4 source columns */
	st.my_idt = (uint32_t)((X0 - U0) * 4) + (uint32_t)min(lane, nact - 1) * 16;
	st.my_vx = (uint32_t)(lane + 1) * 4;                                /* exchange [q][lane]  */

	st.gout = dst + (size_t)Y0 * st.pitch + (size_t)xl * 4;

	int r = 0;
	while (r < nrows) {
		const int j = r & 63;
		if (r && j == 0) build_rows(r);
		if constexpr (TAPS == 0) {
			if (j + 3 < 64 && r + 3 < nrows) { scan_rows_none<4>(st, r, Y0 + r); r += 4; }
			else { scan_rows_none<1>(st, r, Y0 + r); r += 1; }
		} else {
			if (j + 3 < 64 && r + 3 < nrows && ((st.regmask >> j) & 0xFull) == 0xFull) {
				scan_block4<TAPS>(st, r, Y0 + r);
				r += 4;
			} else {
				scan_row_any<TAPS>(st, r, Y0 + r);
				r += 1;
			}
		}
	}
}

extern "C" hipError_t
fixca_launch_scan(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	const uint8_t *s = static_cast<const uint8_t *>(d_src);
	uint8_t *d = static_cast<uint8_t *>(d_dst);
	const size_t lds = (size_t)pl->ring * RING_ROW + 2 * VX_CH * 4;
	const void *fn = g->interp == 0 ? (const void *)k_scan<0> : g->interp == 1 ? (const void *)k_scan<2> : (const void *)k_scan<4>;
	hipError_t ea = (hipError_t)fixca_dyn_lds(fn, 64 * 1024);
	if (ea != hipSuccess) return ea;
	if (g->interp == 0)      hipLaunchKernelGGL(k_scan<0>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	else if (g->interp == 1) hipLaunchKernelGGL(k_scan<2>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	else                     hipLaunchKernelGGL(k_scan<4>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	return hipGetLastError();
}
