"""NumPy twin of the device's counter-based draws (pta_replicator_amd/csrc/pta_rng.h).

TEST INFRASTRUCTURE ONLY.  Throughput mode does not use the reference's RNG (NumPy's legacy
global MT19937 stream cannot be generated in parallel), so there is nothing in the reference
to restate here; this file restates the published Philox-4x32-10 algorithm (Salmon, Moraes,
Dror & Shaw, SC'11, "Parallel random numbers: as easy as 1, 2, 3"; constants of the Random123
distribution, v1.14) and is pinned by Random123's known-answer vectors in
tests/test_hostcheck.py.  With it a test can regenerate on the host the exact deviates a fused
kernel used and push them through the reference algebra (oracle/pta_oracle.py).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_GWB, STREAM_RN, STREAM_WN, STREAM_ECORR, STREAM_TD = 1, 2, 3, 4, 5


def stream_id(kind, pulsar):
    return ((kind << 24) | (pulsar & 0xFFFFFF)) & 0xFFFFFFFF


def philox4x32_10(ctr, key):
    """ctr: [n,4] uint32, key: [2] uint32 -> [n,4] uint32."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def _blocks(seed, realisation, stream, npairs, pair0=0):
    ctr = np.zeros((npairs, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(pair0, pair0 + npairs, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = np.uint32(stream)
    ctr[:, 2] = np.uint32(realisation & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32((realisation >> 32) & 0xFFFFFFFF)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def uniform_pairs(seed, realisation, stream, npairs, pair0=0):
    """u1 in (0,1], u2 in [0,1): 52 random bits dropped into the mantissa of a double in [1,2)."""
    v = _blocks(seed, realisation, stream, npairs, pair0).astype(np.uint64)
    one = np.uint64(0x3FF0000000000000)
    a = (((v[:, 0] << np.uint64(32)) | v[:, 1]) >> np.uint64(12)) | one
    b = (((v[:, 2] << np.uint64(32)) | v[:, 3]) >> np.uint64(12)) | one
    u1 = 2.0 - a.view(np.float64)
    u2 = b.view(np.float64) - 1.0
    return u1, u2


def normal_pairs(seed, realisation, stream, npairs, pair0=0, fast=False):
    """(z0[npairs], z1[npairs]): Box-Muller of the two uniforms of each Philox block.  The device evaluates
    -2 ln u and sin/cos(2 pi u) with its own < 1 ulp kernels (pta_rng.h); libm here agrees to ~1e-16.
    fast=True mirrors the opt-in fp32 transform (hardware log/sqrt/sin/cos): agreement ~1e-6."""
    u1, u2 = uniform_pairs(seed, realisation, stream, npairs, pair0)
    if fast:
        rad = np.sqrt(np.float32(-1.3862943611198906) * np.log2(u1.astype(np.float32)))
        x = np.float32(6.2831853071795864769) * u2.astype(np.float32)
        return (rad * np.cos(x)).astype(np.float64), (rad * np.sin(x)).astype(np.float64)
    rad = np.sqrt(np.maximum(-2.0 * np.log(u1), 1e-300))
    q = np.rint(4.0 * u2)                  # exact reduction to the nearest quarter turn, like the device
    x = 6.283185307179586 * (u2 - 0.25 * q)
    sr, cr = np.sin(x), np.cos(x)
    qi = q.astype(np.int64) & 3
    s = np.choose(qi, [sr, cr, -sr, -cr])
    c = np.choose(qi, [cr, -sr, -cr, sr])
    return rad * c, rad * s


def normals(seed, realisation, stream, n):
    """single-deviate view: element e = pair e>>1, branch e&1."""
    z0, z1 = normal_pairs(seed, realisation, stream, (n + 1) // 2)
    out = np.empty(2 * len(z0))
    out[0::2], out[1::2] = z0, z1
    return out[:n]
