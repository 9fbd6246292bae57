#!/usr/bin/env python3
"""tools/clock_under_load.py -- (GPU box, SEA_MI355X_LIB = a -DSEA_NS6_TIMING variant, SEA_NS_KERNEL=pipe6d) the shader clock a long
launch actually runs at: workgroup 0's frame loop counted on the shader clock (role timers: work + wait of one role over all beats) over
its span on the constant 100 MHz counter, for a short launch (1024 utterances, ~2 ms) and a long one (12 500, ~26 ms)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
lib = ctypes.CDLL(sea.LIB_PATH)
for n in (1024, 12500):
    batch = bench.build_shard(n, 0, torch.device("cuda", 0))
    for _ in range(3): sea.ns_denoise_batch(batch)
    torch.cuda.synchronize()
    t = (ctypes.c_ulonglong * 16)()
    assert lib.sea_debug_ns6_timing(t) == 0
    buf = (ctypes.c_uint * 4)()
    assert lib.sea_debug_ns6_wg(buf, 1) == 0
    wall_ticks = (buf[1] - buf[0]) & 0xffffffff
    clk = t[10] + t[11]  # role 5 (S): work + wait over all beats of workgroup 0
    print(n, "utterances: workgroup 0:", int(t[12]), "beats,", clk, "shader clocks in", wall_ticks * 10, "ns ->", round(clk / (wall_ticks * 10.0), 3), "GHz;",
          round(clk / max(1, int(t[12]))), "clk per beat")
