#!/usr/bin/env python3
"""tools/ns_residency.py [n_utt] -- (GPU box, -DSEA_NS_TIMING build of ns_pipe_kernel.hip) how many workgroups of a four-wave NoiseSup
launch are resident on a CU at the same time: from every workgroup's start / end on the constant 100 MHz counter and its CU
(HW_ID), the maximum and the time-weighted mean overlap per CU.  SEA_NS_KERNEL selects the form (pipe / big)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = bench.build_shard(n, 0, torch.device("cuda", 0))
lib = ctypes.CDLL(sea.LIB_PATH)
for _ in range(2): sea.ns_denoise_batch(batch)
torch.cuda.synchronize()
m = min(n, 4096)
buf = (ctypes.c_uint * (4 * m))()
assert lib.sea_debug_ns_wg(buf, m) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(m, 4).astype(np.int64)
start = (a[:, 3] - a[:, 3].min()) & 0xffffffff
end = start + a[:, 0]
key = (a[:, 2] & 0xF) * 1000 + ((a[:, 1] >> 13) & 0x7) * 100 + ((a[:, 1] >> 8) & 0xF)
mx, mean = [], []
for k in np.unique(key):
    s, e = start[key == k], end[key == k]
    ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
    cur = best = 0
    area = 0
    last = ev[0][0]
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        best = max(best, cur)
    mx.append(best)
    mean.append(area / max(1, ev[-1][0] - ev[0][0]))
print(f"{os.environ.get('SEA_NS_KERNEL', 'auto')} n_utt {n} (first {m} workgroups): CUs {len(mx)}, resident workgroups per CU: max over time min/median/max over CUs",
      int(np.min(mx)), int(np.median(mx)), int(np.max(mx)), "| time-weighted mean", round(float(np.mean(mean)), 2))
gaps = []
for k in np.unique(key)[:64]:
    s, e = np.sort(start[key == k]), np.sort(end[key == k])
    for t in e[:-6]:
        nxt = s[s >= t]
        if len(nxt): gaps.append(nxt[0] - t)
if gaps: print("time from a workgroup's end to the next start on its CU (us): median", float(np.median(gaps)) * 1e-2, "p90", float(np.percentile(gaps, 90)) * 1e-2, "mean", float(np.mean(gaps)) * 1e-2)
print("workgroup life (us): median", float(np.median(end - start)) * 1e-2)
