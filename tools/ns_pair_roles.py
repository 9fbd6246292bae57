#!/usr/bin/env python3
"""tools/ns_pair_roles.py [n_utt] [frames] -- (GPU box) work / barrier-wait shader clocks per beat of the five waves of workgroup 0
of the two-utterances-per-workgroup NoiseSup form, from a -DSEA_P2_TIMING build:
    tools/build_variant.sh p2_t speech_enhancement_amd/csrc/ns_pipe2_kernel.hip -DSEA_P2_TIMING
    SEA_MI355X_LIB=ablate/libsea_p2_t.so python tools/ns_pair_roles.py 12288 800"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import speech_enhancement_amd as sea  # noqa: E402
from speech_enhancement_amd import corpus  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 800
    lib = sea.load()
    raw = ctypes.CDLL(sea._lib.LIB_PATH)
    base = [corpus.synth_utterance(300 + k, 80 * nf) for k in range(8)]
    batch = sea.PackedBatch.from_arrays([base[k % 8] for k in range(n)])
    out = torch.zeros_like(batch.data)
    prev = lib.sea_ns_kernel_form(5)
    sea.ns_denoise_batch(batch, out=out)
    torch.cuda.synchronize()
    raw.sea_debug_p2_ck(None, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    sea.ns_denoise_batch(batch, out=out)
    b.record()
    torch.cuda.synchronize()
    lib.sea_ns_kernel_form(prev)
    t = (ctypes.c_ulonglong * 16)()
    raw.sea_debug_p2_timing(t)
    beats = nf + 4
    roles = " ".join(f"{r} {t[2 * k] // beats}+{t[2 * k + 1] // beats}" for k, r in enumerate(("Fa", "Fb", "B0", "B1", "S")))
    ms = a.elapsed_time(b)
    ck = (ctypes.c_ulonglong * 8)()
    raw.sea_debug_p2_ck(ck, 0)
    if any(ck):
        print("checkpoints of the chosen wave, clk per beat:", [int(c) // beats for c in ck])
    print(f"pair form, {n} x {nf} frames: {ms:.3f} ms = {n * nf / ms / 1e3:.1f} M frames/s | clk per beat, work+wait: {roles}")


if __name__ == "__main__":
    main()
