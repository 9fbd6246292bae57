#!/usr/bin/env python3
"""tools/ns_mixed.py -- experiment: the configs[1] batch split by utterance length into TWO launches on two
streams, each with its own kernel form (run on the GPU box).  Prints ms per step for every (split, form pair)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FORMS = {"single": 1, "pipe": 2, "pipe6": 3, "big": 4}


def main():
    import torch
    import bench
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(sea.LIB_PATH)
    ids = list(range(1024))
    ids.sort(key=lambda u: -corpus.utterance_length(u))
    frames = sum(corpus.utterance_length(u) // 80 for u in ids)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = [a.split(":") for a in (sys.argv[1:] or ["0:pipe:pipe", "256:pipe6:pipe", "384:pipe6:pipe", "512:pipe6:pipe",
                                                    "512:pipe6:big", "512:pipe:big", "256:pipe6:big", "1024:pipe6:pipe", "1024:big:pipe"])]
    for n_long, fa, fb in cases:
        n_long = int(n_long)
        parts = []
        if n_long > 0:
            parts.append((bench.build_shard_ids(ids[:n_long], dev), FORMS[fa], s1))
        if n_long < 1024:
            parts.append((bench.build_shard_ids(ids[n_long:], dev), FORMS[fb], s2))
        outs = [torch.zeros_like(b.data) for b, _, _ in parts]

        def step():
            for (b, form, st), out in zip(parts, outs):
                lib.sea_ns_kernel_form(form)
                with torch.cuda.stream(st):
                    sea.ns_denoise_batch(b, out=out)
        torch.cuda.synchronize()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 10
        ms = []
        for _ in range(K):
            torch.cuda.synchronize()
            e0.record()
            s1.wait_event(e0); s2.wait_event(e0)
            step()
            cur = torch.cuda.current_stream()
            cur.wait_stream(s1); cur.wait_stream(s2)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        print(json.dumps({"long": n_long, "form_long": fa, "form_short": fb, "ms_median": round(ms[K // 2], 3), "ms_min": round(ms[0], 3),
                          "Mframes_s": round(frames / ms[K // 2] / 1e3, 1)}), flush=True)
        lib.sea_ns_kernel_form(0)


if __name__ == "__main__":
    main()
