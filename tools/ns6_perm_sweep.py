#!/usr/bin/env python3
"""tools/ns6_perm_sweep.py [rounds] [steps] -- (GPU box) configs[1] step of the dense six-wave form for EVERY wave -> role map
(720 permutations of roles 0 FA, 1 FB, 2 B0, 3 N1, 4 G1, 5 S over waves 0..5), `rounds` passes over all of them in a fresh random
order each, `steps` launches per measurement; prints the maps sorted by their median step.  One process: the map is a launch
argument (sea_debug_ns6_perm)."""
import ctypes, itertools, os, random, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
batch = bench.build_shard(1024, 0, dev)
lib = ctypes.CDLL(sea.LIB_PATH)
lib.sea_ns_kernel_form(6)
out = torch.empty_like(batch.data)


def enc(w):  # three bits per wave, wave 0 lowest
    v = 0
    for k, r in enumerate(w):
        v |= r << (3 * k)
    return v


def octal(w):
    return "0" + "".join(str(w[k]) for k in (5, 4, 3, 2, 1, 0))


perms = list(itertools.permutations(range(6)))
times = {p: [] for p in perms}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    order = perms[:]
    random.Random(1234 + r).shuffle(order)
    t0 = time.time()
    for n, p in enumerate(order):
        lib.sea_debug_ns6_perm(enc(p))
        sea.ns_denoise_batch(batch, out=out)
        e0.record()
        for _ in range(steps):
            sea.ns_denoise_batch(batch, out=out)
        e1.record()
        torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / steps)
        if n % 120 == 0:
            print(f"round {r}: {n} / {len(order)} maps, {time.time() - t0:.0f} s", flush=True)
lib.sea_debug_ns6_perm(0)
res = sorted(((float(np.median(v)), float(np.min(v)), octal(p)) for p, v in times.items()))
names = ["FA", "FB", "B0", "N1", "G1", "S"]
for med, mn, o in res[:40] + res[-5:]:
    w = [int(c) for c in o[1:]][::-1]
    print(o, f"median {med:.3f} ms  min {mn:.3f}", " ".join(names[x] for x in w))
