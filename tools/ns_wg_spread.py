#!/usr/bin/env python3
"""tools/ns_wg_spread.py -- how long each workgroup of ns_denoise_pipe_kernel spends in its frame loop, by XCC and CU
(needs the -DSEA_NS_TIMING variant: SEA_MI355X_LIB=ablate/libsea_<name>.so).  Equal-length batch."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    dev = torch.device("cuda", 0)
    L = 64000
    base = corpus.synth_utterance(3, L)
    lib = ctypes.CDLL(sea.LIB_PATH)
    for n in (256, 512, 1024):
        batch = sea.PackedBatch.from_arrays([base] * n, dev)
        for _ in range(2):
            sea.ns_denoise_batch(batch)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint * (4 * n))()
        assert lib.sea_debug_ns_wg(buf, n) == 0
        a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 4)
        span = a[:, 0].astype(np.float64) * 10.0 / (L // 80)     # ns per frame
        start = (a[:, 3] - a[:, 3].min()).astype(np.float64) * 10e-3  # us after the first workgroup started
        xcc = a[:, 2] & 0xF
        cu = (a[:, 1] >> 8) & 0xF
        se = (a[:, 1] >> 13) & 0x7
        key = xcc * 1000 + se * 100 + cu
        per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
        out = {"n_utt": n, "ns_per_frame(min,median,p90,max)": [round(float(v)) for v in (span.min(), np.median(span), np.percentile(span, 90), span.max())],
               "start_us(median,max)": [round(float(np.median(start)), 1), round(float(start.max()), 1)],
               "workgroups_per_CU(min,max,distinct_CUs)": [int(per_cu.min()), int(per_cu.max()), int(per_cu.size)],
               "median_ns_per_frame_by_xcc": [round(float(np.median(span[xcc == x]))) for x in range(8) if np.any(xcc == x)]}
        # does the per-frame time follow the number of workgroups sharing the CU?
        cnt = per_cu[np.unique(key, return_inverse=True)[1]]
        out["median_ns_per_frame_by_workgroups_on_the_CU"] = {int(c): round(float(np.median(span[cnt == c]))) for c in sorted(set(cnt.tolist()))}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
