#!/usr/bin/env python3
"""tools/sustained_vs_burst.py -- (GPU box) the dense six-wave NoiseSup form on 1024 EQUAL utterances (four per CU, in lock step) of
300 / 3000 / 12000 frames: ns per frame beat as a function of how long the launch lasts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import speech_enhancement_amd as sea
from speech_enhancement_amd import corpus
dev = torch.device("cuda", 0)
base = corpus.synth_utterance(5, 80 * 12000)
for nfr in (300, 3000, 12000):
    utts = [base[: 80 * nfr]] * 1024
    batch = sea.PackedBatch.from_arrays(utts)
    out = torch.zeros_like(batch.data)
    for _ in range(2): sea.ns_denoise_batch(batch, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): sea.ns_denoise_batch(batch, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(os.environ.get("SEA_NS_KERNEL", "auto"), nfr, "frames x 1024:", round(ms, 3), "ms,", round(ms * 1e6 / (nfr + 7), 1), "ns per beat,",
          round(1024 * nfr / ms / 1e3, 1), "M frames/s")
    del batch, out
