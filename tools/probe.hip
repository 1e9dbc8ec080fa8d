// probe — semantics of v_cvt_pk_u8_f32 on gfx950 (rounding, saturation)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float *in, unsigned *out, int n)
{
	int i = threadIdx.x;
	if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
int main()
{
	float h[] = { -5.f, -0.5f, -0.0f, 0.49f, 0.5f, 0.51f, 1.49f, 1.5f, 2.5f, 3.5f, 126.5f, 127.5f, 254.49f, 254.5f, 254.51f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, NAN, INFINITY };
	int n = sizeof h / sizeof h[0];
	float *d; unsigned *o, ho[64];
	hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 4);
	hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
	hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
	for (int i = 0; i < n; ++i) printf("%12g -> byte %3u  (word %08x)\n", h[i], (ho[i] >> 8) & 0xff, ho[i]);
	return 0;
}
