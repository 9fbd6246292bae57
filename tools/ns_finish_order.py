#!/usr/bin/env python3
"""tools/ns_finish_order.py -- which workgroups of the configs[1] launch finish last, and the per-frame time of
each launch row (= priority level on its CU).  Needs the -DSEA_NS_TIMING variant (SEA_MI355X_LIB=ablate/libsea_<name>.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
dev = torch.device("cuda", 0)
batch = bench.build_shard(1024, 0, dev)
lib = ctypes.CDLL(sea.LIB_PATH)
for _ in range(3): sea.ns_denoise_batch(batch)
torch.cuda.synchronize()
n = 1024
buf = (ctypes.c_uint * (4 * n))()
assert lib.sea_debug_ns_wg(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 4)
order = batch.order.cpu().numpy()
frames = np.asarray(batch.host_lengths)[order] // 80
start = (a[:, 3] - a[:, 3].min()).astype(np.int64) & 0xffffffff
end = (start + a[:, 0]) * 10e-3   # us
span = a[:, 0] * 10.0 / np.maximum(frames, 1)
idx = np.argsort(-end)[:12]
print("last finishers: block, frames, ns/frame, end_us")
for b in idx: print(int(b), int(frames[b]), round(float(span[b])), round(float(end[b]), 1))
print("block 0..5:", [(int(frames[b]), round(float(span[b])), round(float(end[b]),1)) for b in range(6)])
for lo, hi in ((0,256),(256,512),(512,768),(768,1024)):
    print("row", lo//256, "median ns/frame", round(float(np.median(span[lo:hi]))), "max end_us", round(float(end[lo:hi].max()),1))
# spread of the finishing times over the CUs (workgroups b, b + 256, b + 512, b + 768 are launched onto one CU when the dispatcher
# deals the first 256 out one per CU): the latest finisher of each such group
grp = end.reshape(4, 256).max(axis=0)
print("per launch column: latest end_us  min", round(float(grp.min()), 1), "median", round(float(np.median(grp)), 1), "max", round(float(grp.max()), 1))
fr = frames.reshape(4, 256).sum(axis=0)
print("frames per launch column: min", int(fr.min()), "max", int(fr.max()))
hw = a[:, 1]
xcc = (a[:, 2] & 0xf).astype(np.int64)
for x in range(8):
    sel = xcc == x
    if sel.any():
        cols = np.unique(np.arange(n)[sel] % 256)
        print("xcc", x, "workgroups", int(sel.sum()), "latest end_us median over its columns", round(float(np.median(grp[cols])), 1), "max", round(float(grp[cols].max()), 1),
              "median ns/frame", round(float(np.median(span[sel]))))
