#!/usr/bin/env python3
"""tools/rs_timing.py -- per-role work / barrier-wait cycles of the resynthesis kernels (needs the
-DSEA_RS_TIMING variant: SEA_MI355X_LIB=ablate/libsea_rs_timing.so).  Equal-length batch."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    L = 48000
    base, mask1 = corpus.synth_utterance(3, L), corpus.synth_mask(3, L)
    lib = ctypes.CDLL(sea.LIB_PATH)
    for n in (256, 1024):
        batch = sea.PackedBatch.from_arrays([base] * n, dev)
        masks = sea.MaskBatch.from_arrays([mask1] * n, dev)
        for _ in range(2):
            sea.resynth_batch(batch, masks)
            sea.subband_batch(batch)
        torch.cuda.synchronize()
        t = (ctypes.c_ulonglong * 32)()
        assert lib.sea_debug_rs_timing(t) == 0
        tiles = L / 16
        names = ["fwd R1", "fwd R2", "fwd R3", "bwd R1", "bwd R2", "bwd W", "bwd SUM"]
        print(json.dumps({"n_utt": n, **{nm: {"work_cyc_per_tile": round(t[2 * i] / tiles), "wait_cyc_per_tile": round(t[2 * i + 1] / tiles)}
                                         for i, nm in enumerate(names)},
                          **{nm: {"work_cyc_per_tile": round(t[16 + 2 * i] / tiles), "wait_cyc_per_tile": round(t[17 + 2 * i] / tiles)}
                             for i, nm in enumerate(["sb R1", "sb R2", "sb K", "sb HC", "sb W"])}}), flush=True)


if __name__ == "__main__":
    main()
