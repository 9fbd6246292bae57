#!/usr/bin/env python3
"""tools/ns16k_time.py [streams] [frames] -- HIP-event time of sea_ns16k_streams_push (the 16 k-native NoiseSup variant,
SURVEY 8(f) #4) on int16-valued random streams; prints frames/s and microseconds per frame per stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import speech_enhancement_amd as sea  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    lib = sea.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(16)
    fr = torch.randint(-6000, 6000, (B, nf, 160), device=dev, generator=gen).float()
    out = torch.zeros_like(fr)
    pr = torch.zeros((B, nf), dtype=torch.int32, device=dev)
    fl = torch.zeros((B, nf), dtype=torch.uint8, device=dev)
    ct = torch.zeros((B, nf), dtype=torch.int32, device=dev)
    w = torch.zeros((B, nf, 25), dtype=torch.float32, device=dev)
    st = torch.zeros((B, lib.sea_ns16k_state_floats()), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        assert lib.sea_ns16k_streams_push(fr.data_ptr(), out.data_ptr(), pr.data_ptr(), fl.data_ptr(), ct.data_ptr(), w.data_ptr(),
                                          st.data_ptr(), B, nf, 1, s) == 0
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    print(f"ns16k: {B} streams x {nf} frames: {ms:.3f} ms, {B * nf / ms / 1e3:.2f} M frames/s, {ms * 1e3 / nf:.2f} us per frame per stream")


if __name__ == "__main__":
    main()
