#!/usr/bin/env python3
"""tools/ns6_residency.py [n_utt] -- (GPU box, SEA_MI355X_LIB = a -DSEA_NS6_TIMING variant of ns_pipe6_kernel.hip, SEA_NS_KERNEL=pipe6d)
the dense six-wave NoiseSup form on LPT shard 0 of the configs[4] corpus: start / end / CU of every workgroup -> workgroups resident
per CU over time, clocks per frame of a workgroup, how busy the CU slots are."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
batch = bench.build_shard(n, 0, torch.device("cuda", 0))
lib = ctypes.CDLL(sea.LIB_PATH)
for _ in range(2): sea.ns_denoise_batch(batch)
torch.cuda.synchronize()
m = min(n, 16384)
buf = (ctypes.c_uint * (4 * m))()
assert lib.sea_debug_ns6_wg(buf, m) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(m, 4).astype(np.int64)
t0 = a[:, 0].min()
start = (a[:, 0] - t0) & 0xffffffff
end = (a[:, 1] - t0) & 0xffffffff
order = batch.order.cpu().numpy()[:m]
frames = np.asarray(batch.host_lengths)[order] // 80
span = float(end.max())
print("workgroups", m, "frames", int(frames.sum()), "span ms", span * 1e-5, "M frames/s", frames.sum() / (span * 1e-8) / 1e6)
ns_per_frame = (end - start) * 10.0 / np.maximum(frames, 1)
q = np.argsort(start)
for lo, hi in ((0, 1024), (1024, 4096), (4096, 8192), (8192, m)):
    sel = q[lo:hi]
    if len(sel): print("workgroups by start", lo, hi, "ns per frame: median", round(float(np.median(ns_per_frame[sel])), 1), "frames median", int(np.median(frames[sel])))
key = (a[:, 3] & 0xF) * 1000 + ((a[:, 2] >> 13) & 0x7) * 100 + ((a[:, 2] >> 8) & 0xF)
mx, mean = [], []
for k in np.unique(key):
    s, e = start[key == k], end[key == k]
    ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
    cur = best = 0
    area = 0
    last = 0
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        best = max(best, cur)
    mx.append(best)
    mean.append(area / span)
print("CUs", len(mx), "resident workgroups per CU: max over time min/median/max over CUs", int(np.min(mx)), int(np.median(mx)), int(np.max(mx)),
      "| time-weighted mean over the launch", round(float(np.mean(mean)), 2))
gaps = []
for k in np.unique(key)[:64]:
    s, e = np.sort(start[key == k]), np.sort(end[key == k])
    # time from an end to the next start on that CU
    for t in e[:-4]:
        nxt = s[s >= t]
        if len(nxt): gaps.append(nxt[0] - t)
print("time from a workgroup's end to the next start on its CU (us): median", float(np.median(gaps)) * 1e-2, "p90", float(np.percentile(gaps, 90)) * 1e-2)
