#!/usr/bin/env python3
"""tools/ns_period.py -- frame period of the NoiseSup kernel forms on equal-length batches (run on the GPU box).
One process per form (the form override is read once): SEA_NS_KERNEL=pipe|pipe6|pipe8 python tools/ns_period.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    L = 64000
    base = corpus.synth_utterance(3, L)
    out = {"form": os.environ.get("SEA_NS_KERNEL", "auto"), "frames_per_utt": L // 80}
    for n in [int(x) for x in (sys.argv[1:] or ["128", "256", "512", "768", "1024"])]:
        batch = sea.PackedBatch.from_arrays([base] * n, dev)
        sea.ns_denoise_batch(batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            sea.ns_denoise_batch(batch)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[str(n)] = {"ms": round(ms, 3), "ns_per_frame_of_one_utt": round(ms * 1e6 / (L // 80)), "Mframes_s": round(n * (L // 80) / ms / 1e3, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
