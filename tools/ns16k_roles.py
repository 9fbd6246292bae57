#!/usr/bin/env python3
"""tools/ns16k_roles.py [streams] [frames] -- (GPU box) work / barrier-wait shader clocks per frame of the four role waves of
stream 0 of the pipelined 16 k-native kernel, from a -DSEA16P_TIMING build:
    tools/build_variant.sh ns16p_t speech_enhancement_amd/csrc/ns16k_pipe_kernel.hip -DSEA16P_TIMING
    SEA_MI355X_LIB=ablate/libsea_ns16p_t.so python tools/ns16k_roles.py 1024 400"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import speech_enhancement_amd as sea  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    lib = sea.load()
    raw = ctypes.CDLL(sea._lib.LIB_PATH)
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
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    t = (ctypes.c_ulonglong * 16)()
    raw.sea_debug_ns16p_timing(t, 0)
    n = nf + 5
    roles = " ".join(f"{r} {t[2 * k] // n}+{t[2 * k + 1] // n}" for k, r in enumerate(("F", "B0", "B1", "S")))
    print(f"ns16k pipe, {B} x {nf}: {a.elapsed_time(b):.3f} ms | clk per beat, work+wait: {roles}")


if __name__ == "__main__":
    main()
