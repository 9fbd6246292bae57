#!/usr/bin/env python3
"""tools/order_sweep.py -- timing of launch-order policies for the utterance-per-workgroup kernels.
Workgroups b, b+256, b+512, b+768 of a 1024-workgroup launch share a CU (tools/hwid_probe.hip), so the
order array decides which utterances contend for one CU's SIMDs."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tools.bench_extra import timed


def policies(lengths, width=256):
    n = len(lengths)
    desc = np.argsort(-lengths, kind="stable").astype(np.int32)
    out = {"desc": desc, "identity": np.arange(n, dtype=np.int32)}
    rows = [desc[i:i + width] for i in range(0, n, width)]
    out["serpentine"] = np.concatenate([r[::-1] if k & 1 else r for k, r in enumerate(rows)]).astype(np.int32)
    if len(rows) == 4:
        # row 0 descending; rows 1..3 dealt so that every column's total is as equal as possible (greedy LPT
        # on column sums, one row at a time)
        cols = lengths[rows[0]].astype(np.float64).copy()
        res = [rows[0]]
        for r in rows[1:]:
            take = np.empty(width, dtype=np.int32)
            by_sum = np.argsort(cols, kind="stable")            # smallest column sum first
            take[by_sum] = r[:width]                             # gets the longest of this row
            cols += lengths[take]
            res.append(take)
        out["lpt_columns"] = np.concatenate(res).astype(np.int32)
    return out


def main():
    import torch
    import speech_enhancement_amd as sea
    dev = torch.device("cuda", 0)
    batch = bench.build_shard(1024, 0, dev)
    out = torch.zeros_like(batch.data)
    ref = None
    for name, order in policies(np.asarray(batch.host_lengths)).items():
        batch.order = torch.from_numpy(order).to(dev)
        wall, ker = timed(lambda: sea.ns_denoise_batch(batch, out=out), 10, warmup=2)
        same = True
        if ref is None:
            ref = out.clone()
        else:
            same = bool(torch.equal(ref, out))
        print(json.dumps({"order": name, "ns_ms": ker * 1e3, "same_output": same}), flush=True)


if __name__ == "__main__":
    main()
