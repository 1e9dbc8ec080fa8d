/*
 * exp/fixca_roll_exp.h — timing instrumentation of the row-rolling kernel.  NOT PART OF THE PRODUCT:
 * compiled in only with -DFIXCA_EXPERIMENTS (tools/stamps.sh builds such a library under tools/bin/),
 * and a library built that way OVERWRITES THE OUTPUT IMAGE with the numbers it collects.
 *
 * What it records, per wave (one tile), with s_memtime (shader clock) and s_memrealtime (100 MHz):
 *   acc[0] outside blocks   acc[1] fetch: DMA issue + vmcnt wait   acc[2] ring reads issued and landed
 *   acc[3] filter arithmetic   acc[4] conversions + store issue   acc[5] prologue (tables, first ring fill)
 * and writes 8 x u64 per workgroup to the start of the image: acc[0..3], start (100 MHz ticks), lifetime
 * (100 MHz ticks), total shader cycles, and a tag: gen flag | strip << 8 | first row << 20 | XCC id << 40 | HW_ID[15:4] << 44.
 * tools/stamps.py prints the summary.
 */
#ifndef FIXCA_ROLL_EXP_H
#define FIXCA_ROLL_EXP_H

#define ROLL_EXP_FIELDS unsigned long long last, acc[8], t_begin, rt_begin;

#define STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st.acc[i] += t_ - st.last; st.last = t_; } while (0)

#define ROLL_EXP_BEGIN(st) do { (st).last = __builtin_amdgcn_s_memtime(); (st).t_begin = (st).last; \
		(st).rt_begin = __builtin_amdgcn_s_memrealtime(); for (int i_ = 0; i_ < 8; ++i_) (st).acc[i_] = 0; } while (0)

/* make the arithmetic finish before the stamp that follows it */
#define ROLL_EXP_SINK(acc, NR) do { float sink_ = 0.f; \
		_Pragma("unroll") for (int n_ = 0; n_ < (NR); ++n_) \
		_Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) sink_ += (acc)[n_][k_].x + (acc)[n_][k_].y; \
		asm volatile("v_readfirstlane_b32 s0, %0" :: "v"(sink_) : "s0"); } while (0)

#define ROLL_EXP_END(st, dst, lane, gen) do { STAMP(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
		if ((lane) < 8) { \
			unsigned long long v_ = (st).acc[0]; \
			for (int i_ = 1; i_ < 8; ++i_) v_ = ((lane) == i_) ? (st).acc[i_] : v_; \
			if ((lane) == 6) v_ = (st).last - (st).t_begin; \
			if ((lane) == 7) v_ = (gen) | ((unsigned long long)(__builtin_amdgcn_s_getreg(6164) & 15u) << 40) /* HW_REG_XCC_ID[3:0] */ \
				| ((unsigned long long)((__builtin_amdgcn_s_getreg(63492) >> 4) & 0xfffu) << 44); /* HW_REG_HW_ID[15:4]: SIMD, CU, SH, SE */ \
			if ((lane) == 5) v_ = __builtin_amdgcn_s_memrealtime() - (st).rt_begin; \
			if ((lane) == 4) v_ = (st).rt_begin; \
			reinterpret_cast<unsigned long long *>(dst)[(size_t)blockIdx.x * 8 + (lane)] = v_; \
		} } while (0)

#endif
