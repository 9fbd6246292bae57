#!/usr/bin/env python3
"""tools/rs_finish_order.py -- end time of every workgroup of the fused resynthesis launch on the bench corpus, by launch row
(needs the -DSEA_RS_TIMING variant: SEA_MI355X_LIB=ablate/libsea_<name>.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
from speech_enhancement_amd import corpus
dev = torch.device("cuda", 0)
batch = bench.build_shard(1024, 0, dev)
masks = sea.MaskBatch.from_arrays([corpus.synth_mask(u, int(L)) for u, L in enumerate(batch.host_lengths)], dev)
scratch = torch.empty(sea.resynth_scratch_elems(batch), dtype=torch.float32, device=dev)
out = torch.zeros_like(batch.data)
lib = ctypes.CDLL(sea.LIB_PATH)
for _ in range(3): sea.resynth_batch(batch, masks, binary=False, out=out, scratch=scratch)
torch.cuda.synchronize()
n = 1024
buf = (ctypes.c_uint * (4 * n))()
assert lib.sea_debug_rs_wg(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 4).astype(np.int64)
t0 = a[:, 0].min()
start = ((a[:, 0] - t0) & 0xffffffff) * 10e-3
end = ((a[:, 1] - t0) & 0xffffffff) * 10e-3
order = batch.order.cpu().numpy()
L = np.asarray(batch.host_lengths)[order]
for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1024)):
    print("row", lo // 256, "median samples", int(np.median(L[lo:hi])), "median ns/sample", round(float(np.median((end - start)[lo:hi] * 1e3 / L[lo:hi])), 1),
          "end_us median", round(float(np.median(end[lo:hi])), 1), "max", round(float(end[lo:hi].max()), 1))
grp = end.reshape(4, 256).max(axis=0)
print("per launch column: latest end_us  min", round(float(grp.min()), 1), "median", round(float(np.median(grp)), 1), "max", round(float(grp.max()), 1))
# concurrently resident workgroups per CU (HW_ID: CU, SE; XCC_ID)
key = (a[:, 3] & 0xF) * 1000 + ((a[:, 2] >> 13) & 0x7) * 100 + ((a[:, 2] >> 8) & 0xF)
s0 = (a[:, 0] - t0) & 0xffffffff
e0 = (a[:, 1] - t0) & 0xffffffff
mx = []
for k in np.unique(key):
    ev = sorted([(t, 1) for t in s0[key == k]] + [(t, -1) for t in e0[key == k]])
    cur = best = 0
    for t, d in ev:
        cur += d
        best = max(best, cur)
    mx.append(best)
print("CUs", len(mx), "resident workgroups per CU at the same time: min / median / max over CUs", int(np.min(mx)), int(np.median(mx)), int(np.max(mx)))
