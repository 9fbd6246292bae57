#!/usr/bin/env python3
"""tools/ns16k_swizzle_search.py -- the XOR swizzle of the 512-word work area of the 16 k-native variant's transform.

rfft (x, 512, 8) (aurora_etsi/rfft.cpp:46-181 with NS_FFT_ORDER 8 on NS_FFT_LENGTH 512) as the pipelined kernel's
transform wave runs it (csrc/ns16k_pipe_kernel.hip): lane l owns places 8l .. 8l+7 for the register-resident start
(length-2, n2 = 4, n2 = 8), then the levels n2 = 16 .. 256 go through LDS, one work item per lane and level -- a PAIR
(plain butterfly on i + k n4, pi/4 butterfly on i + n8 + k n4) or a TWIDDLED butterfly (i + j + k n4, i + n4 - j + k n4),
eight operands each -- and FFTtoPSD reads x[2b], x[2b+1], x[512-2b], x[511-2b] for b = l, l + 64.

ds_read_b32 / ds_write_b32 bank = word index mod 32, the two 32-lane halves of a wave are served separately, identical
addresses broadcast (MI355X_MICROARCH.md, LDS): the cost of one instruction is, per half, the largest number of DISTINCT
words on one bank.  With the natural layout the items of a level sit 16, 32, ... words apart and pile onto a few banks
(round 3's one-wave kernel: SQ_LDS_BANK_CONFLICT = 59 % of the LDS-active cycles).  Searched here: word i -> i ^ S[i >> 5],
S[0..15] in 0..31, and the item -> lane assignment (items dealt alternately to the two halves), by random restarts +
coordinate descent on the total number of bank passes.  Prints the table csrc/sea_tables.c carries (SEA16_SWZ).
"""
import random
import sys

N, M = 512, 8


def blocks(n, n2):
    out, is_, id_ = [], 0, n2 << 1
    while is_ < n:
        out += list(range(is_, n, id_))
        is_ = (id_ << 1) - n2
        id_ <<= 2
    return out


def level_items(n2):
    """list of 8-tuples of word indices, one per work item of level n2"""
    n4, n8 = n2 >> 2, n2 >> 3
    items = []
    for i in blocks(N, n2):
        for j in range(1, n8):
            items.append(tuple([i + j + k * n4 for k in range(4)] + [i + n4 - j + k * n4 for k in range(4)]))
    for i in blocks(N, n2):
        items.append(tuple([i + k * n4 for k in range(4)] + [i + n8 + k * n4 for k in range(4)]))
    return items


def lane_of(k):
    """item k of a level -> lane: dealt alternately to the two halves"""
    return (k & 1) * 32 + (k >> 1)


def instructions():
    """every LDS instruction of the transform wave as a list of 64 word indices (None: lane idle -> copies lane 0 / 32)"""
    ins = []
    for j in range(8):                      # the head's stores: place 8l + j
        ins.append([8 * l + j for l in range(64)])
    for n2 in (16, 32, 64, 128, 256):
        items = level_items(n2)
        assert len(items) <= 64, (n2, len(items))
        per_lane = [None] * 64
        for k, it in enumerate(items):
            per_lane[lane_of(k)] = it
        for op in range(8):                 # eight operand reads; the stores hit the same words
            ins.append([(per_lane[l][op] if per_lane[l] else None) for l in range(64)])
            ins.append(ins[-1])
    for half in (0, 1):                     # FFTtoPSD, bins b = l + 64 half
        for f in (lambda b: 2 * b, lambda b: 2 * b + 1, lambda b: (512 - 2 * b) % 512, lambda b: 511 - 2 * b):
            ins.append([f(l + 64 * half) for l in range(64)])
    return ins


def cost(S, ins):
    total = 0
    for words in ins:
        for h in (0, 1):
            seen = {}
            for l in range(32 * h, 32 * h + 32):
                w = words[l]
                if w is None:
                    continue
                a = w ^ S[w >> 5]
                seen.setdefault(a & 31, set()).add(a)
            total += max((len(v) for v in seen.values()), default=1)
    return total


def main():
    random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    ins = instructions()
    base = cost([0] * 16, ins)
    best, bestS = base, [0] * 16
    for restart in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        S = [0] + [random.randrange(32) for _ in range(15)]
        c = cost(S, ins)
        improved = True
        while improved:
            improved = False
            for b in range(1, 16):
                keep = S[b]
                for v in range(32):
                    S[b] = v
                    cv = cost(S, ins)
                    if cv < c:
                        c, keep, improved = cv, v, True
                S[b] = keep
        if c < best:
            best, bestS = c, list(S)
            print(f"restart {restart}: {c} bank passes (natural layout {base}, floor {2 * len(ins)})", bestS, flush=True)
    print("SEA16_SWZ =", bestS, "passes", best, "natural", base, "floor", 2 * len(ins))


if __name__ == "__main__":
    main()
