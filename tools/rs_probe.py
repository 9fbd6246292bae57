#!/usr/bin/env python3
"""tools/rs_probe.py -- per-tile period of the resynthesis / subband kernels on EQUAL-length batches
(no tail imbalance): N utterances of L samples, N = 256 .. 2048 (1 .. 8 workgroups per CU)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_extra import timed


def main():
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    L = int(os.environ.get("RS_L", 48000))
    base = corpus.synth_utterance(3, L)
    mask1 = corpus.synth_mask(3, L)
    for n in (256, 512, 1024, 2048):
        batch = sea.PackedBatch.from_arrays([base] * n, dev)
        masks = sea.MaskBatch.from_arrays([mask1] * n, dev)
        scratch = torch.empty(sea.resynth_scratch_elems(batch), dtype=torch.float32, device=dev)
        out = torch.zeros_like(batch.data)
        _, ker = timed(lambda: sea.resynth_batch(batch, masks, out=out, scratch=scratch), 3)
        sb = torch.zeros(batch.total * 64, dtype=torch.int16, device=dev)
        _, ksb = timed(lambda: sea.subband_batch(batch, out=sb), 3)
        _, kns = timed(lambda: sea.ns_denoise_batch(batch), 3)
        tiles = L / 16
        print(json.dumps({"n_utt": n, "L": L, "resynth_ms": ker * 1e3, "resynth_ns_per_tile_both_passes": ker * 1e9 / tiles,
                          "subband_ms": ksb * 1e3, "subband_ns_per_tile": ksb * 1e9 / tiles,
                          "ns_ms": kns * 1e3, "ns_ns_per_frame": kns * 1e9 / (L // 80)}), flush=True)
        del scratch, sb, out, batch, masks


if __name__ == "__main__":
    main()
