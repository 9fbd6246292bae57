#!/usr/bin/env python3
"""tools/ns16k_phases.py -- (GPU box, with SEA_MI355X_LIB=ablate/libsea_<v>.so built -DSEA16_TIMING) shader clocks per frame
workgroup 0 of ns16k_stream_kernel spends in each phase (the checkpoints cost ~100 clk each)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import speech_enhancement_amd as sea  # noqa: E402

lib = sea.load()
raw = ctypes.CDLL(os.environ["SEA_MI355X_LIB"])
B, nf = 1024, 200
fr = torch.randint(-6000, 6000, (B, nf, 160), device="cuda").float()
r = sea.ns16k_streams_push(fr)
torch.cuda.synchronize()
z = np.zeros(24, np.uint64)
raw.sea_ns16k_timing(z.ctypes.data_as(ctypes.c_void_p), 1)
r = sea.ns16k_streams_push(fr)
torch.cuda.synchronize()
raw.sea_ns16k_timing(z.ctypes.data_as(ctypes.c_void_p), 0)
names = ["window", "fft", "vad", "bins", "fd_var", "gamma", "fd_spec+gainfact", "idct", "fir"]
for st in (0, 1):
    print(f"stage {st}: " + "  ".join(f"{n} {int(z[st * 10 + k]) // nf}" for k, n in enumerate(names)))
print(f"gate {int(z[20]) // nf}  slide {int(z[21]) // nf}  dc+store {int(z[22]) // nf}  | total {int(z.sum()) // nf} clk per frame")
