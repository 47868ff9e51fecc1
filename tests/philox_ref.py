"""NumPy restatement of the normals the running estimator makes on the device (k_error.hip: xi_fill_kernel).

TESTS ONLY.  Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) with
key = the 64-bit seed and counter = (sample id low, sample id high, draw pair, 0); the four output words give two
53-bit uniforms in (0, 1) and Box-Muller turns them into two draws of that sample (call j = 32 b + r, r < 32, makes
draws 64 b + r and 64 b + r + 32: the two rows one thread of the device GEMM stages of a 64-draw tile).  The generator itself
is pinned by its published known-answer vectors (test_host_logic.py), the device against this file
(test_gpu_kernels.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32-valued, key: (2,) -> (..., 4) uint64 array of 32-bit words."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        m0, m1 = M0 * c[0], M1 * c[2]
        c = [(m1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), m1 & MASK,
             (m0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), m0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1)


def normals(seed, sample_ids, draws=1024):
    """Xi[d][k] for the given sample ids: (draws, len(sample_ids)) float64."""
    ids = np.asarray(sample_ids, dtype=np.uint64)
    j = np.arange(draws // 2, dtype=np.uint64)
    ctr = np.zeros((len(j), len(ids), 4), dtype=np.uint64)
    ctr[..., 0] = (ids & MASK)[None, :]
    ctr[..., 1] = (ids >> np.uint64(32))[None, :]
    ctr[..., 2] = j[:, None]
    seed = int(seed) & (2 ** 64 - 1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    u1 = (((w[..., 0] >> np.uint64(5)) << np.uint64(26)) | (w[..., 1] >> np.uint64(6))).astype(np.float64)
    u2 = (((w[..., 2] >> np.uint64(5)) << np.uint64(26)) | (w[..., 3] >> np.uint64(6))).astype(np.float64)
    u1 = (u1 + 0.5) * 2.0 ** -53
    u2 = (u2 + 0.5) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    # call j = 32 b + r (r < 32) gives draws 64 b + r and 64 b + r + 32
    out = np.empty((draws, len(ids)))
    d = 64 * (np.arange(draws // 2) // 32) + np.arange(draws // 2) % 32
    out[d] = rad * np.cos(2.0 * np.pi * u2)
    out[d + 32] = rad * np.sin(2.0 * np.pi * u2)
    return out


# Known-answer vectors of Philox4x32-10 (Random123 distribution, kat_vectors): counter, key -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
