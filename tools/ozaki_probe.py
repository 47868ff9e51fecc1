"""fp64-by-int8 (Ozaki-type) contraction: what the arithmetic delivers, on the CPU (round-4 review, item 9).

An fp64 dot product over k terms is emulated the way an int8 matrix pipe would compute it: every row of A and every
column of B is scaled by a power of two to |x| < 1/2 and cut into s slices of b bits (round-to-nearest digits,
|d| <= 2^(b-1): int8 for b <= 7), the slice products A_t . B_u are EXACT integers (k 2^(2b-2) < 2^31 for k = 1024, b <= 7), and
those with t + u < s are summed with their weights 2^(-b (t + u + 2)).  Reported: the largest error of the result
relative to sum |a||b| (the scale both this scheme's and fp64's own error bounds are stated in) against the fp64
dot product's, for the k-loop shape of the panel kernel (k = 1024) on Gaussian and on ill-scaled operands, and the
number of int8 products per fp64 product, s (s + 1) / 2.

    python3 tools/ozaki_probe.py
"""
import numpy as np
from fractions import Fraction


def slices(x, s, b):
    """x: float64 array with |x| < 1 -> s integer digit arrays d_t with x ~ sum_t d_t 2^(-b (t + 1))."""
    out, r = [], x.astype(np.longdouble)
    for _ in range(s):
        r = r * (2 ** b)
        d = np.floor(r + 0.5)
        d = np.clip(d, -(2 ** (b - 1)), 2 ** (b - 1))      # round-to-nearest digits: |d| <= 2^(b-1), an int8 for b <= 7
        out.append(d.astype(np.int64))
        r = r - d
    return out


def emulate(A, B, s, b):
    ea = np.ceil(np.log2(np.abs(A).max(axis=1) * (1 + 2.0 ** -40))) + 1      # row scales of A (powers of two)
    eb = np.ceil(np.log2(np.abs(B).max(axis=0) * (1 + 2.0 ** -40))) + 1      # column scales of B
    As, Bs = slices(A / 2.0 ** ea[:, None], s, b), slices(B / 2.0 ** eb[None, :], s, b)
    C = np.zeros((A.shape[0], B.shape[1]), dtype=np.longdouble)
    for order in range(2 * s - 1):
        if order >= s:
            break                                   # truncated: only t + u < s
        acc = np.zeros(C.shape, dtype=np.int64)     # slice products of one order add up exactly in integers
        for t in range(order + 1):
            acc += As[t] @ Bs[order - t]
        C += acc.astype(np.longdouble) * np.longdouble(2.0) ** (-b * (order + 2))
    return (C * 2.0 ** ea[:, None] * 2.0 ** eb[None, :]).astype(np.float64)


def exact(A, B):
    out = np.empty((A.shape[0], B.shape[1]))
    for i in range(A.shape[0]):
        for j in range(B.shape[1]):
            out[i, j] = float(sum(Fraction(float(a)) * Fraction(float(c)) for a, c in zip(A[i], B[:, j])))
    return out


def main():
    rng = np.random.default_rng(0)
    k, m = 1024, 12
    cases = {"gaussian": (rng.standard_normal((m, k)), rng.standard_normal((k, m))),
             "ill-scaled (entries over 8 decades)": (rng.standard_normal((m, k)) * 10.0 ** rng.uniform(-4, 4, (m, k)),
                                                     rng.standard_normal((k, m)) * 10.0 ** rng.uniform(-4, 4, (k, m)))}
    for name, (A, B) in cases.items():
        ref = exact(A, B)
        scale = np.abs(A) @ np.abs(B)
        e64 = np.abs(A @ B - ref) / scale
        print(f"{name}: fp64 dot (numpy)               max err / sum|a||b| = {e64.max():.2e}")
        for b in (6, 7):
            for s in range(6, 11):
                err = np.abs(emulate(A, B, s, b) - ref) / scale
                print(f"   {b}-bit slices, s = {s:2d}: {s * (s + 1) // 2:3d} int8 products per fp64 product, "
                      f"max err / sum|a||b| = {err.max():.2e}")


if __name__ == "__main__":
    main()
