// dpbench — issue rate of the fp64 VALU operations k_wcol is made of (v_add_f64, v_mul_f64, v_fma_f64, v_mov_b64) per SIMD,
// for 1..4 waves per SIMD and dependent / independent chains.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dpbench tools/dpbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int OP, int CH>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
	double x[CH];
#pragma unroll
	for (int i = 0; i < CH; ++i) x[i] = a + threadIdx.x + i;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 16; ++u)
#pragma unroll
			for (int i = 0; i < CH; ++i) {
				if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
				if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
				if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
				if (OP == 3) asm volatile("v_mov_b64 %0, %1" : "+v"(x[i]) : "v"(b));
				if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&x[i]) : "v"((float)b));
				if (OP == 5) asm volatile("v_mul_f64 %0, %0, 0.5 clamp" : "+v"(x[i]));
			}
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < CH; ++i) s += x[i];
	if (s == 123.456) out[0] = s;
}
template <int OP, int CH> int run(const char *name, double *d)
{
	const int iters = 2000;
	for (int waves : { 1024, 2048, 3072, 4096 }) {
		hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		hipLaunchKernelGGL((k<OP, CH>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0, 1.0000001);
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k<OP, CH>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0, 1.0000001);
		CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		const double n = (double)iters * 16 * CH * (waves / 1024.0);      /* instructions per SIMD */
		printf("%-28s chains %d waves/SIMD %d: %8.1f us  %.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, CH, waves / 1024, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 2.4);
	}
	return 0;
}
// the same dependent-chain kernel launched as workgroups of 4 waves (a workgroup's waves go to the CU's 4 SIMDs in turn)
template <int BS> int run_blocks(double *d)
{
	const int iters = 2000;
	for (int waves : { 1024, 2048, 3072, 4096 }) {
		hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		hipLaunchKernelGGL((k<0, 1>), dim3(waves * 64 / BS), dim3(BS), 0, 0, d, iters, 1.0, 1.0000001);
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k<0, 1>), dim3(waves * 64 / BS), dim3(BS), 0, 0, d, iters, 1.0, 1.0000001);
		CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("v_add_f64 dependent, workgroups of %3d threads, %d waves per SIMD: %8.1f us\n", BS, waves / 1024, ms * 1e3);
	}
	return 0;
}
int main()
{
	double *d; CK(hipMalloc(&d, 64));
	run_blocks<64>(d); run_blocks<128>(d); run_blocks<256>(d);
	run<0, 1>("v_add_f64 dependent", d); run<0, 4>("v_add_f64", d); run<1, 4>("v_mul_f64", d); run<2, 4>("v_fma_f64", d);
	run<3, 4>("v_mov_b64", d); run<4, 4>("v_add_f32", d); run<5, 4>("v_mul_f64 clamp", d); run<1, 1>("v_mul_f64 dependent", d); run<1, 2>("v_mul_f64 2 chains", d);
	return 0;
}
