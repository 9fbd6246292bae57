#!/usr/bin/env python3
"""tools/gen_rfft_exact.py N M NIN > csrc/<name>.inc -- the reference's rfft (x, N, M) as a straight-line codelet for ONE lane.

etsi/cpp/rfft.c:45-180 unrolled by running its loop nest on NAMES instead of numbers: the digit-reverse counter becomes a
renaming, every butterfly a handful of float statements in the reference's own order with the reference's own expressions
(twiddles as the float the reference computes: (float)cos((double)(float)(j * e)), emitted as hexadecimal literals; the
division by M_SQRT2 of a float sum in double as the multiplication sea_selftest_pi4 proves identical for every float).  Inputs
through the caller's macro CC_E(i), i < NIN (input i = 0 for i >= NIN: the zero padding), outputs X[0..N-1] in the reference's order Re(0..N/2), Im(N/2-1..1).

The only liberty: an operation on a LITERAL zero is folded (0 + 0, x + 0, x - 0 -> x, 0 - x -> -x, 0 * c): that can change the
SIGN of a zero result and nothing else, and every consumer of the spectrum squares it (CompCeps.c:451-459).  No FMA is formed,
no sum reassociated: compiled with -ffp-contract=off the lane performs the reference's arithmetic operation by operation, which
is what keeps csrc/cc_kernel.hip's per-lane CompCeps bit-identical to the LDS-transform form and to the oracle.
"""
import math
import struct
import sys


def f32(v):
    return struct.unpack("<f", struct.pack("<f", v))[0]


class Emit:
    def __init__(self):
        self.lines = []
        self.n = 0
        self.fetched = set()

    def tmp(self, expr):
        import re
        for m in re.finditer(r"\be(\d+)\b", expr):
            i = int(m.group(1))
            if i not in self.fetched:
                self.fetched.add(i)
                self.lines.append(f"const float e{i} = CC_E({i});")
        self.n += 1
        name = f"v{self.n}"
        self.lines.append(f"const float {name} = {expr};")
        return name


ZERO = "0"


def neg(a):
    if a == ZERO:
        return ZERO
    return a[1:] if a.startswith("-") else "-" + a


def add(E, a, b):
    if a == ZERO:
        return b
    if b == ZERO:
        return a
    if b.startswith("-"):
        return E.tmp(f"{a} - {b[1:]}")
    return E.tmp(f"{a} + {b}")


def sub(E, a, b):
    return add(E, a, neg(b))


def mulc(E, a, c):
    """a * literal float c"""
    if a == ZERO:
        return ZERO
    return E.tmp(f"{a} * {float(c).hex()}f")


def main():
    N, M, NIN = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    E = Emit()
    # inputs are fetched lazily: the statement `const float eI = CC_E(I);` is emitted right before the first use of input I, so
    # that a caller can make CC_E(I) the whole expression that produces the sample (loads included) and nothing is held early
    x = [f"e{i}" if i < NIN else ZERO for i in range(N)]
    # digit reverse counter (rfft.c:57-79)
    j = 0
    for i in range(N - 1):
        if i < j:
            x[i], x[j] = x[j], x[i]
        k = N >> 1
        while k <= j:
            j -= k
            k >>= 1
        j += k
    # length two butterflies (:82-96)
    is_, id_ = 0, 4
    while is_ < N - 1:
        for i0 in range(is_, N, id_):
            i1 = i0 + 1
            a0, a1 = x[i0], x[i1]
            x[i0] = add(E, a0, a1)
            x[i1] = sub(E, a0, a1)
        is_ = (id_ << 1) - 2
        id_ <<= 2
    # L shaped butterflies (:99-178)
    n2 = 2
    for k in range(1, M):
        n2 <<= 1
        n4, n8 = n2 >> 2, n2 >> 3
        e = f32((math.pi * 2) / n2)
        is_, id_ = 0, n2 << 1
        while is_ < N:
            for i in range(is_, N, id_):
                i1, i2 = i, i + n4
                i3, i4 = i2 + n4, i2 + 2 * n4
                t1 = add(E, x[i4], x[i3])
                x4n = sub(E, x[i4], x[i3])
                x3n = sub(E, x[i1], t1)
                x1n = add(E, x[i1], t1)
                x[i4], x[i3], x[i1] = x4n, x3n, x1n
                if n4 != 1:
                    i1 += n8
                    i2 += n8
                    i3 += n8
                    i4 += n8
                    s = add(E, x[i3], x[i4])
                    d = sub(E, x[i3], x[i4])
                    t1 = ZERO if s == ZERO else E.tmp(f"(float)((double)({s}) * 0.70710678118654752440)")
                    t2 = ZERO if d == ZERO else E.tmp(f"(float)((double)({d}) * 0.70710678118654752440)")
                    x4n = sub(E, x[i2], t1)
                    x3n = sub(E, neg(x[i2]), t1)
                    x2n = sub(E, x[i1], t2)
                    x1n = add(E, x[i1], t2)
                    x[i4], x[i3], x[i2], x[i1] = x4n, x3n, x2n, x1n
            is_ = (id_ << 1) - n2
            id_ <<= 2
        for jj in range(1, n8):
            a = f32(jj * e)
            a3 = f32(3 * a)
            cc1, ss1 = f32(math.cos(a)), f32(math.sin(a))
            cc3, ss3 = f32(math.cos(a3)), f32(math.sin(a3))
            is_, id_ = 0, n2 << 1
            while is_ < N:
                for i in range(is_, N, id_):
                    i1 = i + jj
                    i2, i3, i4 = i1 + n4, i1 + 2 * n4, i1 + 3 * n4
                    i5 = i + n4 - jj
                    i6, i7, i8 = i5 + n4, i5 + 2 * n4, i5 + 3 * n4
                    t1 = add(E, mulc(E, x[i3], cc1), mulc(E, x[i7], ss1))
                    t2 = sub(E, mulc(E, x[i7], cc1), mulc(E, x[i3], ss1))
                    t3 = add(E, mulc(E, x[i4], cc3), mulc(E, x[i8], ss3))
                    t4 = sub(E, mulc(E, x[i8], cc3), mulc(E, x[i4], ss3))
                    t5 = add(E, t1, t3)
                    t6 = add(E, t2, t4)
                    t3 = sub(E, t1, t3)
                    t4 = sub(E, t2, t4)
                    n3 = sub(E, t6, x[i6])
                    n8_ = add(E, x[i6], t6)
                    n7 = sub(E, neg(x[i2]), t3)
                    n4_ = sub(E, x[i2], t3)
                    n6 = sub(E, x[i1], t5)
                    n1 = add(E, x[i1], t5)
                    n5 = sub(E, x[i5], t4)
                    n2_ = add(E, x[i5], t4)
                    x[i3], x[i8], x[i7], x[i4], x[i6], x[i1], x[i5], x[i2] = n3, n8_, n7, n4_, n6, n1, n5, n2_
                is_ = (id_ << 1) - n2
                id_ <<= 2
    print(f"/* generated by tools/gen_rfft_exact.py {N} {M} {NIN}: etsi/cpp/rfft.c:45-180 unrolled for one lane, {E.n} float statements; "
          f"inputs through the caller's macro CC_E(i), i = 0..{NIN - 1}, each fetched right before its first use; outputs X[0..{N - 1}] */")
    for l in E.lines:
        print(l)
    for i in range(N):
        v = x[i]
        print(f"X[{i}] = {'0.0f' if v == ZERO else v};")


if __name__ == "__main__":
    main()
