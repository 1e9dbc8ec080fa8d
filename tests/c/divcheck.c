/* Exhaustive check behind fixca_wide.hip's div_exact: for d = 2^n - 1 and r = RN(1/d),
 * fma(fma(-(x*r), d, x), r, x*r) == x / d for every n-bit unsigned x (n = 8, 16, 32).
 * Prints the number of mismatches per width; exit code 0 iff all are zero. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>

static long check(uint64_t maxv)
{
	const double d = (double)maxv, r = 1.0 / d;
	long bad = 0;
	for (uint64_t v = 0; v <= maxv; ++v) {
		const double x = (double)v, q = x * r;
		if (fma(fma(-q, d, x), r, q) != x / d) ++bad;
	}
	return bad;
}

int main(void)
{
	const long b8 = check(255u), b16 = check(65535u), b32 = check(4294967295ull);
	printf("u8 %ld u16 %ld u32 %ld\n", b8, b16, b32);
	return (b8 | b16 | b32) ? 1 : 0;
}
