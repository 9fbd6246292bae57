#!/usr/bin/env python3
"""tools/host_soak.py [calls] -- (GPU box) sea_denoise_utterances on the bench corpus over and over, from three host
threads at once (each with its own staging, streams and state; one shared packing pool), every output compared with the
first call's: a race in the time-slice pipeline (state hand-over between launches, staging reuse, non-temporal stores
becoming visible late) would show as a differing call."""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch  # noqa: F401
    import bench
    import speech_enhancement_amd as sea
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    lib = sea.load()
    dev = torch.device("cuda", 0)
    batch = bench.build_shard_ids(list(range(1024)), dev)
    host = batch.data.cpu().numpy()
    ins = [np.ascontiguousarray(host[o:o + l]) for o, l in zip(batch.host_offsets, batch.host_lengths)]
    n = len(ins)
    pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ins])
    lens = (ctypes.c_long * n)(*[x.size for x in ins])
    ref = [np.zeros_like(x) for x in ins]
    pref = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ref])
    assert lib.sea_denoise_utterances(pin, pref, lens, n) == 0
    bad = [0, 0, 0]
    t0 = time.time()

    def worker(k):
        assert lib.sea_init(0) == 0
        outs = [np.zeros_like(x) for x in ins]
        po = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
        for c in range(calls):
            for y in outs[:: 37]:
                y[:] = 0
            assert lib.sea_denoise_utterances(pin, po, lens, n) == 0
            bad[k] += int(not all(np.array_equal(a, b) for a, b in zip(outs, ref)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    res = {"host_threads": 3, "calls_per_thread": calls, "utterances_per_call": n, "differing_calls": bad, "seconds": round(time.time() - t0, 1)}
    print(json.dumps(res))
    sys.exit(1 if any(bad) else 0)


if __name__ == "__main__":
    main()
