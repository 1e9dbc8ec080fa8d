"""Integer-defined synthetic images, so that C, numpy and HIP
agree byte for byte.

noise(seed): SplitMix64 stream; word k (k >= 1) = mix(seed + k*GOLDEN) with the
30/27/31 xorshift-multiply finaliser, emitted little-endian, filling the packed
H x W x C buffer in memory order.  Check: seed 1 -> 193,92,2,137,236,45,10,145.

smooth: v(x,y,c) = (3x + 5y + 40c + ((x*y) >> 4)) & 0xff.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def noise_bytes(seed: int, nbytes: int) -> np.ndarray:
    """First `nbytes` bytes of the SplitMix64 stream for `seed` (uint8)."""
    nwords = (nbytes + 7) // 8
    out = np.empty(nwords, dtype=np.uint64)
    chunk = 1 << 22
    with np.errstate(over="ignore"):
        for start in range(0, nwords, chunk):
            n = min(chunk, nwords - start)
            k = np.arange(start + 1, start + 1 + n, dtype=np.uint64)
            z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
            out[start:start + n] = z
    return out.view(np.uint8)[:nbytes] if out.dtype.byteorder != ">" else \
        out.astype("<u8").view(np.uint8)[:nbytes]


def noise_image(seed: int, width: int, height: int, channels: int) -> np.ndarray:
    """Packed H x W x C uint8 noise image."""
    return noise_bytes(seed, width * height * channels).reshape(height, width, channels).copy()


def smooth_image(width: int, height: int, channels: int) -> np.ndarray:
    x = np.arange(width, dtype=np.int64)[None, :, None]
    y = np.arange(height, dtype=np.int64)[:, None, None]
    c = np.arange(channels, dtype=np.int64)[None, None, :]
    v = (3 * x + 5 * y + 40 * c + ((x * y) >> 4)) & 0xFF
    return v.astype(np.uint8)
