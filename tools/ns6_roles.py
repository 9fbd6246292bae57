#!/usr/bin/env python3
"""tools/ns6_roles.py [n_utt] -- (GPU box) shader clocks per frame workgroup 0's six roles of the six-wave NoiseSup forms spend working
/ waiting at the frame barrier, on the bench corpus (needs a -DSEA_NS6_TIMING build: SEA_MI355X_LIB=ablate/libsea_<name>.so)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import speech_enhancement_amd as sea
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = bench.build_shard(n, 0, torch.device("cuda", 0))
lib = ctypes.CDLL(sea.LIB_PATH)
for _ in range(3): sea.ns_denoise_batch(batch)
torch.cuda.synchronize()
t = (ctypes.c_ulonglong * 16)()
assert lib.sea_debug_ns6_timing(t) == 0
perm = os.environ.get("SEA_NS6_PERM", "default")
fr = max(int(t[12]), 1)
print(f"n_utt {n} map {perm}:", " ".join(f"{nm} {t[2 * r] // fr}+{t[2 * r + 1] // fr}" for r, nm in enumerate(["FA", "FB", "B0", "N1", "G1", "S"])), "clk per frame (work+wait), workgroup 0")
