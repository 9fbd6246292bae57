#!/usr/bin/env python3
"""tools/bench_extra.py -- measurements of the other hot-path kernels (BASELINE configs[2], [3] and
the CompCeps front-end) on one MI355X.  bench.py stays the headline (NoiseSup, configs[1]); this
script prints one JSON line per workload with the same roofline convention.

    python tools/bench_extra.py [--utts 1024] [--steps 5] [--what resynth,ibm,ceps,rfft]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (build_shard)

HBM_PEAK_GBPS = 8000.0


def timed(fn, steps, warmup=1):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    return wall, float(np.mean([a.elapsed_time(b) for a, b in ev])) / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--what", default="resynth,ibm,subband,ceps,afe,rfft,host")
    args = ap.parse_args()
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    what = set(args.what.split(","))
    batch = bench.build_shard(args.utts, 0, dev)
    audio_s = float(np.sum(batch.host_lengths)) / 16000.0

    if what & {"resynth", "ibm"}:
        masks = sea.MaskBatch.from_arrays([corpus.synth_mask(u, int(L)) for u, L in enumerate(batch.host_lengths)], dev)
        hops = int(np.sum((np.asarray(batch.host_lengths) - 320) // 160 + 1))
        scratch = torch.empty(sea.resynth_scratch_elems(batch), dtype=torch.float32, device=dev)
        out = torch.zeros_like(batch.data)
        for name, binary in (("resynth", False), ("ibm", True)):
            if name not in what:
                continue
            wall, ker = timed(lambda: sea.resynth_batch(batch, masks, binary=binary, out=out, scratch=scratch), args.steps)
            alg = hops * 896
            print(json.dumps({
                "metric": f"resynth_64sub_{'IBM' if binary else 'ori'} hop-frames/sec (160-sample hop, 64 bands)",
                "value": hops / wall, "unit": "hop-frames/s", "ms_per_step": wall * 1e3, "rtf": wall / audio_s,
                "config": {"workload": f"BASELINE configs[{3 if binary else 2}]: {args.utts} utterances, 64-band gammatone "
                                       f"analysis/synthesis, {'ideal binary' if binary else 'ratio'} mask", "hop_frames": hops},
                "roofline": {"bound": "hbm", "kernels": "sea::resynth_fwd_kernel + sea::resynth_bwd_kernel",
                             "achieved": alg / ker / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": alg / ker / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": alg,
                             "intermediate_bytes_per_step": int(batch.total) * 64 * 4 * 2,
                             "intermediate_GBps": int(batch.total) * 64 * 4 * 2 / ker / 1e9, "avg_step_ms": ker * 1e3}}),
                  flush=True)

    if "subband" in what:
        out = torch.zeros(batch.total * 64, dtype=torch.int16, device=dev)
        wall, ker = timed(lambda: sea.subband_batch(batch, out=out), args.steps)
        samples = int(np.sum(batch.host_lengths))
        alg = samples * (2 + 128)
        print(json.dumps({
            "metric": "subbband() samples/sec (gammatone + hair cell -> 64 int16 streams)", "value": samples / wall,
            "unit": "samples/s", "ms_per_step": wall * 1e3, "rtf": wall / audio_s,
            "config": {"workload": f"SURVEY 8(f) #1: {args.utts} utterances, 64-channel analysis to 64 int16 streams"},
            "roofline": {"bound": "hbm", "kernel": "sea::subband_kernel", "achieved": alg / ker / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / ker / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_step": alg, "avg_step_ms": ker * 1e3}}), flush=True)
        del out

    if "irm" in what:   # SURVEY 8(f) #2 on two different subband blocks (as bench.py's also-line)
        sub = torch.zeros(batch.total * 64, dtype=torch.int16, device=dev)
        sub2 = torch.zeros(batch.total * 64, dtype=torch.int16, device=dev)
        keep = batch.data
        batch.data = torch.flip(keep, dims=[0]).contiguous()
        sea.subband_batch(batch, out=sub2)
        batch.data = keep
        sea.subband_batch(batch, out=sub)
        torch.cuda.synchronize()
        wall, ker = timed(lambda: sea.irm_target_batch(batch, sub, sub2), args.steps)
        hops = int(np.sum((np.asarray(batch.host_lengths) - 320) // 160 + 1))
        alg = hops * (2 * 128 * 160 + 256)
        print(json.dumps({
            "metric": "IRM target hop-frames/sec (x 64 channels)", "value": hops / wall, "unit": "hop-frames/s", "ms_per_step": wall * 1e3,
            "config": {"workload": f"SURVEY 8(f) #2: {args.utts} utterances, {hops} frames x 64 channels from two subband blocks"},
            "roofline": {"bound": "hbm", "kernel": "sea::irm_target_kernel", "achieved": alg / ker / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": alg / ker / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": alg,
                         "avg_step_ms": ker * 1e3}}), flush=True)
        del sub, sub2

    if "ceps" in what:
        out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
        torch.cuda.synchronize()
        res = {}

        def run():
            res["c"] = sea.compceps_batch(batch, f32, first)
        wall, ker = timed(run, args.steps)
        n = int(res["c"][2].sum().item())
        alg = n * (320 + 56)
        print(json.dumps({"metric": "CompCeps cepstral frames/sec (from the float NoiseSup stream)", "value": n / wall,
                          "unit": "frames/s", "ms_per_step": wall * 1e3,
                          "config": {"workload": f"{args.utts} utterances, {n} cepstral frames of 14 coefficients"},
                          "roofline": {"bound": "hbm", "kernel": "sea::compceps_kernel", "achieved": alg / ker / 1e9,
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / ker / 1e9 / HBM_PEAK_GBPS,
                                       "algorithmic_bytes_per_step": alg, "avg_step_ms": ker * 1e3}}), flush=True)

    if "afe" in what:
        # NoiseSup (with speech flags) -> WaveProc -> CompCeps -> PostProc -> VAD + flush
        lib = sea.load()
        n = batch.n_utt
        outb = torch.zeros_like(batch.data)
        f32 = torch.zeros(batch.total, dtype=torch.float32, device=dev)
        first = torch.full((n,), -1, dtype=torch.int32, device=dev)
        onset = torch.zeros(n, dtype=torch.int32, device=dev)
        flags = torch.zeros(batch.total // 8, dtype=torch.uint8, device=dev)
        nfr = np.asarray(batch.host_lengths) // 80
        ccum = np.concatenate(([0], np.cumsum(np.maximum(nfr - 6, 0)))).astype(np.int64)
        fcum = np.concatenate(([0], np.cumsum(nfr + 6))).astype(np.int64)
        tc, tf = int(ccum[-1]), int(fcum[-1])
        fcc = torch.zeros((tc, 14), dtype=torch.float32, device=dev)
        f15 = torch.zeros((tf, 15), dtype=torch.float32, device=dev)
        nfe = torch.zeros(n, dtype=torch.int32, device=dev)
        d_ccum, d_fcum = torch.from_numpy(ccum).to(dev), torch.from_numpy(fcum).to(dev)
        P = lambda t: t.data_ptr()
        st = torch.cuda.current_stream().cuda_stream

        def ns_fd():
            assert lib.sea_ns_denoise_batch_fd(P(batch.data), P(outb), P(f32), P(batch.offsets), P(batch.lengths),
                                               P(batch.order), P(first), P(flags), P(onset), n, st) == 0

        def feats():
            assert lib.sea_afe_features_batch(P(f32), P(flags), P(batch.offsets), P(batch.lengths), P(first), P(onset),
                                              P(d_ccum), tc, P(fcc), None, P(d_fcum), P(f15), P(nfe), None, n, st) == 0
        w1, k1 = timed(ns_fd, args.steps)
        w2, k2 = timed(feats, args.steps)
        emitted = int(nfe.sum().item())
        alg = batch.n_frames * 320 + emitted * 60
        print(json.dumps({"metric": "ETSI AFE feature frames/sec (NoiseSup + WaveProc + CompCeps + PostProc + VAD)",
                          "value": emitted / (w1 + w2), "unit": "feature frames/s", "ms_per_step": (w1 + w2) * 1e3,
                          "config": {"workload": f"SURVEY 8(f) #3: {n} utterances, {emitted} emitted feature frames of 15 floats",
                                     "ns_with_flags_ms": k1 * 1e3, "waveproc_compceps_postproc_vad_ms": k2 * 1e3},
                          "roofline": {"bound": "hbm", "kernels": "sea::ns_denoise_pipe_fd_kernel + sea::afe_ceps_kernel + sea::afe_vad_kernel",
                                       "achieved": alg / (k1 + k2) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": alg / (k1 + k2) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": alg,
                                       "avg_step_ms": (k1 + k2) * 1e3}}), flush=True)

    if "host" in what:
        # the host-buffer drop-in path: pack + hipMalloc + H2D + one launch + D2H (PCIe inclusive)
        import ctypes
        lib = sea.load()
        host = batch.data.cpu().numpy()
        ins = [np.ascontiguousarray(host[o:o + l]) for o, l in zip(batch.host_offsets, batch.host_lengths)]
        outs = [np.zeros_like(x) for x in ins]
        n = len(ins)
        pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ins])
        pout = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
        lens = (ctypes.c_long * n)(*[x.size for x in ins])
        lib.sea_denoise_utterances(pin, pout, lens, n)
        per = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ta = time.perf_counter()
            assert lib.sea_denoise_utterances(pin, pout, lens, n) == 0
            per.append((time.perf_counter() - ta) * 1e3)
        wall = (time.perf_counter() - t0) / args.steps
        # the reference's own calling pattern: one etsi_denoise(short*, short*, long) per utterance
        k = min(n, 128)
        fr1 = int(sum(x.size // 80 for x in ins[:k]))
        lib.etsi_denoise(ins[0].ctypes.data, outs[0].ctypes.data, ins[0].size)
        t1 = time.perf_counter()
        for x, y in zip(ins[:k], outs[:k]):
            assert lib.etsi_denoise(x.ctypes.data, y.ctypes.data, x.size) == 0
        per_call = (time.perf_counter() - t1) / k
        print(json.dumps({"metric": "etsi_denoise() drop-in, one call per utterance (PCIe inclusive)",
                          "value": fr1 / (per_call * k), "unit": "frames/s", "ms_per_step": per_call * 1e3,
                          "config": {"workload": f"{k} sequential calls, mean utterance {fr1 / k:.0f} frames; ms_per_step = per call"}}),
              flush=True)
        print(json.dumps({"metric": "NoiseSup frames/sec through the HOST-buffer entry point (PCIe inclusive)",
                          "value": batch.n_frames / wall, "unit": "frames/s", "ms_per_step": wall * 1e3,
                          "config": {"workload": f"sea_denoise_utterances on {n} host utterances: pack | H2D | launch | D2H | unpack "
                                                 f"pipeline over time slices, {lib.sea_host_threads()} packing threads",
                                     "ms_per_call_sorted": [round(v, 2) for v in sorted(per)]}}), flush=True)

    if "hostrs" in what:
        # resynth() through the host-buffer entry point (PCIe inclusive): int16 + mask rows in, int16 out
        import ctypes
        lib = sea.load()
        host = batch.data.cpu().numpy()
        ins = [np.ascontiguousarray(host[o:o + l]) for o, l in zip(batch.host_offsets, batch.host_lengths)]
        outs = [np.zeros_like(x) for x in ins]
        rng = np.random.default_rng(5)
        masks = [rng.random(((x.size - 320) // 160 + 1, 64), dtype=np.float32) for x in ins]
        n = len(ins)
        pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ins])
        pm = (ctypes.c_void_p * n)(*[m.ctypes.data for m in masks])
        pout = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
        lens = (ctypes.c_long * n)(*[x.size for x in ins])
        hops = int(sum(m.shape[0] for m in masks))
        for binary in (0, 1):
            assert lib.sea_resynth_utterances(pin, lens, pm, binary, pout, n) == 0, lib.sea_last_error()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                assert lib.sea_resynth_utterances(pin, lens, pm, binary, pout, n) == 0
            wall = (time.perf_counter() - t0) / args.steps
            print(json.dumps({"metric": f"resynth hop-frames/sec through the HOST-buffer entry point (PCIe inclusive), {'IBM' if binary else 'ratio mask'}",
                              "value": hops / wall, "unit": "hop-frames/s", "ms_per_step": wall * 1e3,
                              "config": {"workload": f"sea_resynth_utterances on {n} host utterances, {hops} mask rows"}}), flush=True)

    if "rfft" in what:
        n = 1 << 18
        x = torch.randn(n, 256, device=dev)
        wall, ker = timed(lambda: sea.rfft_batch(x), args.steps)
        alg = n * 2048
        print(json.dumps({"metric": "rfft256 frames/sec", "value": n / wall, "unit": "frames/s", "ms_per_step": wall * 1e3,
                          "config": {"workload": f"{n} frames of 256 floats"},
                          "roofline": {"bound": "hbm", "kernel": "sea::rfft256_kernel", "achieved": alg / ker / 1e9,
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / ker / 1e9 / HBM_PEAK_GBPS,
                                       "avg_step_ms": ker * 1e3}}), flush=True)


if __name__ == "__main__":
    main()
