import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import speech_enhancement_amd as sea
import bench
dev = torch.device("cuda", 0)
lib = sea.load(); lib.sea_init(-1)
batch = bench.build_shard_ids(range(1024), dev)
host = batch.data.cpu().numpy()
ins = [np.ascontiguousarray(host[o:o + l]) for o, l in zip(batch.host_offsets, batch.host_lengths)]
outs = [np.zeros_like(x) for x in ins]
n = len(ins)
pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ins])
pout = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
lens = (ctypes.c_long * n)(*[x.size for x in ins])
for _ in range(3): assert lib.sea_denoise_utterances(pin, pout, lens, n) == 0
per = []
for _ in range(12):
    t = time.perf_counter(); lib.sea_denoise_utterances(pin, pout, lens, n); per.append(time.perf_counter() - t)
print("ms per call: median %.3f min %.3f" % (np.median(per) * 1e3, min(per) * 1e3), "slices", os.environ.get("SEA_HOST_SLICES"), "threads", lib.sea_host_threads())
