#!/usr/bin/env python3
"""tools/cc_phases.py -- (GPU box, SEA_MI355X_LIB=ablate/libsea_<v>.so built with
``tools/build_variant.sh <v> speech_enhancement_amd/csrc/cc_kernel.hip -DSEA_CC_TIMING``) shader clocks per 16-frame tile
workgroup 0 of compceps_kernel spends in each step (three waves share a SIMD: wall-clock shares, not lone costs)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import speech_enhancement_amd as sea
sea.load()
raw = ctypes.CDLL(os.environ["SEA_MI355X_LIB"])
from speech_enhancement_amd import corpus
utts = corpus.synth_corpus(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, max_len=96000)
batch = sea.PackedBatch.from_arrays(utts)
out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
c = sea.compceps_batch(batch, f32, first)
torch.cuda.synchronize()
z = np.zeros(8, np.uint64)
raw.sea_cc_timing(z.ctypes.data_as(ctypes.c_void_p), 1)
c = sea.compceps_batch(batch, f32, first)
torch.cuda.synchronize()
raw.sea_cc_timing(z.ctypes.data_as(ctypes.c_void_p), 0)
n = int(z[7])
names = ["stage", "energy+logE", "8 pairs: pre-emphasis, transform, power, mel", "23 logs", "DCT", "store"]
print("tiles of workgroup 0:", n, {k: int(z[i]) // max(n, 1) for i, k in enumerate(names)}, "sum", int(z[:6].sum()) // max(n, 1))
# residency: waves of the launch alive at the same time, per CU (HW_ID: CU, SE; XCC_ID) and per SIMD
nw = min(16384, int(os.environ.get("SEA_CC_NWAVES", "8192")))
buf = (ctypes.c_uint * (4 * nw))()
assert raw.sea_cc_waves(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(nw, 4).astype(np.int64)
a = a[a[:, 1] != 0]
t0 = a[:, 0].min()
s0 = (a[:, 0] - t0) & 0xffffffff
e0 = (a[:, 1] - t0) & 0xffffffff
print("waves", len(a), "launch span us", float(e0.max()) * 10e-3, "wave life us: median", float(np.median(e0 - s0)) * 10e-3, "max", float((e0 - s0).max()) * 10e-3,
      "sum of lives / span = mean waves alive", round(float((e0 - s0).sum()) / float(e0.max()), 1))
cu = (a[:, 3] & 0xF) * 1000 + ((a[:, 2] >> 13) & 0x7) * 100 + ((a[:, 2] >> 8) & 0xF)
simd = (a[:, 2] >> 4) & 0x3
for name, key in (("CU", cu), ("SIMD", cu * 10 + simd)):
    mx = []
    for k in np.unique(key):
        ev = sorted([(t, 1) for t in s0[key == k]] + [(t, -1) for t in e0[key == k]])
        cur = best = 0
        for t, d in ev:
            cur += d
            best = max(best, cur)
        mx.append(best)
    print(name + "s", len(mx), "waves alive at the same time per", name, ": min / median / max", int(np.min(mx)), int(np.median(mx)), int(np.max(mx)))
