#!/usr/bin/env python3
"""tools/gen_rdft_codelet.py N NIN KOUT name > csrc/<name>.inc -- a straight-line real-input DFT codelet for one lane.

Emits the body of   X[k] = sum_{n < NIN} x[n] e^(-2 pi i n k / N),  k = 0 .. KOUT-1   (N a power of two, x[n] = 0 for n >= NIN)
as C statements over the floats x[0] .. x[NIN-1], assigning <name>_re[k], <name>_im[k] -- no loops, no memory, every twiddle a literal.  Used by
csrc/irm_kernel.hip (the IRM target needs bins 0..63 of a 512-point spectrum of 320 samples; a lane takes one of
the four polyphase components of a frame: N = 128, NIN = 80, KOUT = 64).

Method: radix-2 decimation in time on a hash-consed expression graph, using the Hermitian symmetry of a real input's
sub-transforms (only k = 0 .. n/2 of each is ever built; X[n/2 - k] = conj(E[k] - w^k O[k]) shares the twiddle product with
X[k] = E[k] + w^k O[k]); signs are carried beside the nodes (negation is an operand modifier on the device), constants fold,
x + 0 / x * 0 / x * 1 vanish -- so the zero padding and the trivial twiddles cost nothing -- dead code is dropped, and a
product with a single use is fused into its consumer as an FMA.  The float arithmetic order is this script's, not any
reference's: the IRM target's parity is unpinned with a 1e-4 tolerance, which is what makes the freedom legal.
"""
import math
import sys


class Graph:
    def __init__(self):
        self.nodes = []          # (op, a, b) with a, b = (sign, id) or for 'in': index, for 'const': value
        self.index = {}

    def _mk(self, key):
        if key not in self.index:
            self.index[key] = len(self.nodes)
            self.nodes.append(key)
        return self.index[key]

    # a value is None (exact zero) or (sign, id)
    def inp(self, i):
        return (1, self._mk(("in", i, None)))

    def const(self, v):
        if v == 0.0:
            return None
        return (1 if v > 0 else -1, self._mk(("const", abs(v), None)))

    def is_const(self, a):
        return a is not None and self.nodes[a[1]][0] == "const"

    def cval(self, a):
        return a[0] * self.nodes[a[1]][1]

    def neg(self, a):
        return None if a is None else (-a[0], a[1])

    def add(self, a, b):
        if a is None:
            return b
        if b is None:
            return a
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) + self.cval(b))
        if a[1] == b[1]:
            if a[0] != b[0]:
                return None
            return self.mul(self.const(2.0), a)
        # canonical form: first operand positive where possible
        if a[1] > b[1]:
            a, b = b, a
        if a[0] < 0:
            r = self.add(self.neg(a), self.neg(b))
            return self.neg(r)
        return (1, self._mk(("add", a, b)))

    def sub(self, a, b):
        return self.add(a, self.neg(b))

    def mul(self, a, b):
        if a is None or b is None:
            return None
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) * self.cval(b))
        if self.is_const(b):
            a, b = b, a
        if self.is_const(a) and self.nodes[a[1]][1] == 1.0:
            return (a[0] * b[0], b[1])
        s = a[0] * b[0]
        ia, ib = (a[1], b[1]) if a[1] <= b[1] else (b[1], a[1])
        return (s, self._mk(("mul", (1, ia), (1, ib))))


def rdft(g, xs):
    """xs: list of values (None = 0), len a power of two; returns [(re, im)] for k = 0 .. n/2"""
    n = len(xs)
    if n == 1:
        return [(xs[0], None)]
    if n == 2:
        return [(g.add(xs[0], xs[1]), None), (g.sub(xs[0], xs[1]), None)]
    E, O = rdft(g, xs[0::2]), rdft(g, xs[1::2])   # k = 0 .. n/4 each
    X = [None] * (n // 2 + 1)
    for k in range(n // 4 + 1):
        er, ei = E[k]
        orr, oi = O[k]
        c, s = math.cos(2 * math.pi * k / n), -math.sin(2 * math.pi * k / n)
        if k == 0:
            c, s = 1.0, 0.0
        if 4 * k == n:
            c, s = 0.0, -1.0
        wr, wi = g.const(c), g.const(s)
        tr = g.sub(g.mul(orr, wr), g.mul(oi, wi))
        ti = g.add(g.mul(orr, wi), g.mul(oi, wr))
        X[k] = (g.add(er, tr), g.add(ei, ti))
        X[n // 2 - k] = (g.sub(er, tr), g.neg(g.sub(ei, ti)))   # conj(E[k] - t)
    return X


def emit(g, outs, name):
    """outs: list of (label, value).  Dead-code elimination, use counts, FMA fusion, C emission."""
    uses = {}
    live = set()
    stack = [v[1] for _, v in outs if v is not None]
    while stack:
        i = stack.pop()
        uses[i] = uses.get(i, 0) + 1
        if i in live:
            continue
        live.add(i)
        op, a, b = g.nodes[i]
        if op in ("add", "mul"):
            stack += [a[1], b[1]]
    lines, nadd, nmul, nfma = [], 0, 0, 0

    def ref(v):
        s, i = v
        op, a, _ = g.nodes[i]
        t = f"x[{a}]" if op == "in" else (repr(float(a)) + "f" if op == "const" else f"t{i}")
        return t if s > 0 else f"-{t}"

    fused = set()
    for i in sorted(live):
        op, a, b = g.nodes[i]
        if op == "add":
            # a + b with b (or a) a single-use product: fma
            for p, q in ((b, a), (a, b)):
                pop, pa, pb = g.nodes[p[1]]
                if pop == "mul" and uses[p[1]] == 1:
                    fused.add(p[1])
                    m1 = ref((p[0] * pa[0], pa[1]))
                    lines.append((i, f"const float t{i} = __builtin_fmaf({m1}, {ref(pb)}, {ref(q)});"))
                    nfma += 1
                    break
            else:
                lines.append((i, f"const float t{i} = {ref(a)} + {ref(b)};".replace("+ -", "- ")))
                nadd += 1
        elif op == "mul":
            lines.append((i, f"const float t{i} = {ref(a)} * {ref(b)};"))
    body = [t for i, t in lines if i not in fused]
    nmul = sum(1 for i, _ in lines if g.nodes[i][0] == "mul" and i not in fused)
    print(f"/* generated by tools/gen_rdft_codelet.py {' '.join(sys.argv[1:])}: {nadd} additions, {nmul} multiplications, {nfma} fused "
          f"multiply-adds = {nadd + nmul + nfma} instructions; inputs x[0..{NIN - 1}], outputs {name}_re[k], {name}_im[k] */")
    for t in body:
        print(t)
    for label, v in outs:
        print(f"{label} = {ref(v) if v is not None else '0.0f'};")


if __name__ == "__main__":
    N, NIN, KOUT, NAME = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    g = Graph()
    xs = [g.inp(i) if i < NIN else None for i in range(N)]
    X = rdft(g, xs)
    outs = []
    for k in range(KOUT):
        outs.append((f"{NAME}_re[{k}]", X[k][0]))
        outs.append((f"{NAME}_im[{k}]", X[k][1]))
    emit(g, outs, NAME)
