#!/usr/bin/env python3
"""tools/ns_timing.py -- per-role work / barrier-wait cycles per frame of ns_denoise_pipe_kernel
(needs the -DSEA_NS_TIMING variant: SEA_MI355X_LIB=ablate/libsea_ns_timing.so).  Equal-length batch."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    L = 48000
    base = corpus.synth_utterance(3, L)
    lib = ctypes.CDLL(sea.LIB_PATH)
    for n in (256, 512, 1024):
        batch = sea.PackedBatch.from_arrays([base] * n, dev)
        sea.ns_denoise_batch(batch)
        torch.cuda.synchronize()
        bk = (ctypes.c_ulonglong * 16)()
        assert lib.sea_debug_ns_back_ck(bk, 1) == 0
        sea.ns_denoise_batch(batch)
        torch.cuda.synchronize()
        assert lib.sea_debug_ns_back_ck(bk, 0) == 0
        t = (ctypes.c_ulonglong * 24)()
        assert lib.sea_debug_ns_timing(t) == 0
        fr = L // 80 + 4
        print(json.dumps({"n_utt": n, **{nm: {"work_cyc_per_frame": round(t[2 * i] / fr), "wait_cyc_per_frame": round(t[2 * i + 1] / fr)}
                                         for i, nm in enumerate(["F", "B0", "B1", "S"])},
                          "S_checkpoints_cyc_per_frame(prep,chains,energy,verify,store)": [round(t[8 + q] / fr) for q in range(5)],
                          "F_checkpoints_cyc_per_frame(input+records,transforms+psd)": [round(t[16] / fr), round(t[17] / fr)],
                          "shader_clock_MHz_during_the_launch": round(100.0 * t[18] / max(t[19], 1)),
                          "frame_period_ns": round(t[19] * 10.0 / fr),
                          "B0_checkpoints_cyc_per_frame(filter,stage,mel,sum+gainfact,idct+fir)": [round(bk[q] / fr) for q in range(5)],
                          "B1_checkpoints_cyc_per_frame(filter,stage,mel,sum+gainfact,idct)": [round(bk[8 + q] / fr) for q in range(5)]}), flush=True)


if __name__ == "__main__":
    main()
