/*
 * fixca_scan.hip — the LDS-ring row-scan kernel (RGBA8), the main kernel of the path.
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
	unsigned long long regmask; /* bit j: both windows of row j advance by exactly one row    */
	/* ring */
	uint8_t *ring, *vx;
	int      KM, H;
	int      ld;                /* highest source row issued                                   */
	int      lastneed;          /* highest source row this wave ever needs                     */
	int      seq, seqtab;       /* vector-memory ops issued; lane s: seq of the load in slot s */
	const uint8_t *gld;         /* global address of my 16 bytes of source row ld + 1          */
	uint8_t *gout;              /* global address of my 16 bytes of the next output row        */
	size_t   pitch;
	/* vertical windows: (red, blue) floats of my 4 columns, T rotating slots */
	f32x2    win[T][4];
	int      prev_rb[2];
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

template <int TAPS>
__device__ __forceinline__ u32x4 scan_ring_row(const ScanState<TAPS> &st, int v)
{
	const int a = clampi(v, 0, st.H - 1);                    /* edge rows replicate */
	return *reinterpret_cast<const u32x4 *>(st.ring + (a & st.KM) * RING_ROW + st.my_ring);
}

/* one output row; P = the window slot that receives the newest source row */
template <int TAPS, int P>
__device__ __forceinline__ void scan_row(ScanState<TAPS> &st, int r, int y)
{
	constexpr int T = ScanState<TAPS>::T;
	constexpr int NT = ScanState<TAPS>::NT;
	const int j = r & 63;
	const int keep = rl(st.t_keep, j), need = rl(st.t_need, j);

	/* keep DEPTH rows in flight beyond the newest row in use, within the ring's capacity */
	const int target = min(min(need + DEPTH, keep + st.KM), st.lastneed);
	while (st.ld < target) scan_issue(st);

	/* wait for row `need`: the operations issued after its load may stay outstanding */
	const int younger = st.seq - rl(st.seqtab, need & st.KM) - 1;
	if (younger >= 2 * DEPTH) wait_vm<2 * DEPTH>();
	else if (younger >= DEPTH) wait_vm<DEPTH>();
	else wait_vm<0>();

	const u32x4 idt = *reinterpret_cast<const u32x4 *>(st.ring + (y & st.KM) * RING_ROW + st.my_idt);
	u32x4 out;

	if constexpr (TAPS == 0) {
		const uint8_t *rowR = st.ring + (rl(st.t_rb[0], j) & st.KM) * RING_ROW;
		const uint8_t *rowB = st.ring + (rl(st.t_rb[1], j) & st.KM) * RING_ROW;
		uint32_t rq = 0, bq = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			rq |= (uint32_t)rowR[st.go[0][i]] << (8 * i);
			bq |= (uint32_t)rowB[st.go[1][i]] << (8 * i);
		}
		out.x = (idt.x & 0xFF00FF00u) | perm(bq, rq, 0x0c040c00);
		out.y = (idt.y & 0xFF00FF00u) | perm(bq, rq, 0x0c050c01);
		out.z = (idt.z & 0xFF00FF00u) | perm(bq, rq, 0x0c060c02);
		out.w = (idt.w & 0xFF00FF00u) | perm(bq, rq, 0x0c070c03);
	} else {
		const int rbR = rl(st.t_rb[0], j), rbB = rl(st.t_rb[1], j);
		if ((st.regmask >> j) & 1ull) {
			/* both windows slide down by one source row: one LDS read and 4 conversions each */
			const u32x4 pr = scan_ring_row(st, rbR + TAPS - 1);
			const u32x4 pb = scan_ring_row(st, rbB + TAPS - 1);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				st.win[P][k].x = (float)(pr[k] & 0xffu);
				st.win[P][k].y = (float)((pb[k] >> 16) & 0xffu);
			}
		} else {
			/* anything else (first row, scale steps, edge rows): refill the windows in slot order */
#pragma unroll
			for (int t = 0; t < TAPS; ++t) {
				const u32x4 pr = scan_ring_row(st, rbR + t);
				const u32x4 pb = scan_ring_row(st, rbB + t);
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					st.win[(P + 1 + t) % T][k].x = (float)(pr[k] & 0xffu);
					st.win[(P + 1 + t) % T][k].y = (float)((pb[k] >> 16) & 0xffu);
				}
			}
		}

		/* vertical filter, red and blue together, on my own 4 columns */
		f32x2 v[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
#pragma unroll
		for (int t = 0; t < TAPS; ++t) {
			const f32x2 w = { rlf(st.t_wy[0][t], j), rlf(st.t_wy[1][t], j) };
#pragma unroll
			for (int k = 0; k < 4; ++k)
				v[k] = __builtin_elementwise_fma(w, st.win[(P + 1 + t) % T][k], v[k]);
		}
		/* publish: exchange index n = 4*lane + k lives at plane k, entry lane + 1 */
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
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		/* horizontal taps of my 4 outputs */
		f32x2 acc[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
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

		/* saturating round-to-nearest conversion straight into the red and blue bytes */
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].x, 0, idt[k]);
			out[k] = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].y, 2, o);
		}
	}

	if (st.active)
		__builtin_nontemporal_store(out, reinterpret_cast<u32x4 *>(st.gout));
	st.gout += st.pitch;
	++st.seq;                                        /* the store is one vector-memory operation */
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

	/* ---- the 256-column source window of this strip ------------------------------ */
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		int a;
		if constexpr (TAPS == 0) a = map_int(X0, g.xc, W, g.s[c], g.xs[c]);
		else a = clampi((int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])) - BACK, 0, W - 1);
		lo = min(lo, a);
	}
	const int U0 = min(lo & ~3, W - FIXCA_WAVE_COLS);
	st.edgeL = (U0 == 0);
	st.edgeR = (U0 + FIXCA_WAVE_COLS == W);

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
			reg = reg && (st.t_rb[c] == prev + 1);
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
		st.lastneed = yl;
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			int rb;
			row_of(yl, c, rb, nullptr);
			st.lastneed = max(st.lastneed, clampi(rb + T - 1, 0, H - 1));
		}
	}

	build_rows(0);
	st.ld = rl(st.t_keep, 0) - 1;
	st.seq = 0;
	st.seqtab = 0;
	st.gld = src + (size_t)(st.ld + 1) * st.pitch + (size_t)(U0 + 4 * lane) * 4;
	st.gout = dst + (size_t)Y0 * st.pitch + (size_t)xl * 4;
	st.prev_rb[0] = st.prev_rb[1] = 0;

	for (int r = 0; r < nrows; r += T) {
		if (r && (r & 63) == 0) build_rows(r);       /* T divides 64 */
		scan_row<TAPS, 0>(st, r, Y0 + r);
		if constexpr (T >= 2) { if (r + 1 < nrows) scan_row<TAPS, 1 % T>(st, r + 1, Y0 + r + 1); }
		if constexpr (T >= 4) {
			if (r + 2 < nrows) scan_row<TAPS, 2 % T>(st, r + 2, Y0 + r + 2);
			if (r + 3 < nrows) scan_row<TAPS, 3 % T>(st, r + 3, Y0 + r + 3);
		}
	}
}

extern "C" hipError_t
fixca_launch_scan(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	const uint8_t *s = static_cast<const uint8_t *>(d_src);
	uint8_t *d = static_cast<uint8_t *>(d_dst);
	const size_t lds = (size_t)pl->ring * RING_ROW + 2 * VX_CH * 4;
	static bool attr_done = false;
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void *)k_scan<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
		(void)hipFuncSetAttribute((const void *)k_scan<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
		(void)hipFuncSetAttribute((const void *)k_scan<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
		attr_done = true;
	}
	if (g->interp == 0)      hipLaunchKernelGGL(k_scan<0>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	else if (g->interp == 1) hipLaunchKernelGGL(k_scan<2>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	else                     hipLaunchKernelGGL(k_scan<4>, dim3(pl->nblocks), dim3(64), lds, stream, s, d, *g, *pl);
	return hipGetLastError();
}
