#!/usr/bin/env python3
"""tools/ns_placement.py -- which SIMD each role wave of ns_denoise_pipe_kernel lands on, per CU (needs the
-DSEA_NS_TIMING variant: SEA_MI355X_LIB=ablate/libsea_ns_timing.so).  bench corpus (configs[1]) and an equal-length batch."""
import ctypes, json, os, sys
from collections import Counter, defaultdict
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import speech_enhancement_amd as sea
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(sea.LIB_PATH)
    n = 1024
    batch = bench.build_shard(n, 0, dev)
    for _ in range(2):
        sea.ns_denoise_batch(batch)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * (4 * n))()
    assert lib.sea_debug_ns_hw(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint32).reshape(n, 4)
    simd = (a >> 4) & 3
    xcc = (a[:, 0] >> 28) & 0xF
    key = xcc * 1000 + ((a[:, 0] >> 8) & 0xFF)   # HW_ID bits 8..15: CU, SH, SE
    distinct = Counter(len(set(simd[b].tolist())) for b in range(n))
    rows = Counter()
    per_cu = defaultdict(list)
    for b in range(n):
        per_cu[int(key[b])].append(b)
    # per CU: how many of its workgroups put role r on SIMD s; max multiplicity = imbalance
    worst = Counter()
    row_mix = Counter()
    for k, bs in per_cu.items():
        m = np.zeros((4, 4), dtype=int)
        for b in bs:
            for r in range(4):
                m[r, simd[b, r]] += 1
        worst[int(m.max())] += 1
        row_mix[tuple(sorted(b // 256 for b in bs))] += 1
    pat = Counter(tuple(simd[b].tolist()) for b in range(n))
    print(json.dumps({"distinct_SIMDs_per_workgroup": dict(distinct), "CUs": len(per_cu),
                      "workgroups_per_CU": dict(Counter(len(v) for v in per_cu.values())),
                      "max_same_role_waves_on_one_SIMD_per_CU": dict(worst),
                      "launch_rows_sharing_a_CU(top5)": [[list(k), v] for k, v in row_mix.most_common(5)],
                      "role_to_SIMD_patterns(top8)": [[list(k), v] for k, v in pat.most_common(8)]}), flush=True)
    # first 4 CUs in detail
    for k in sorted(per_cu)[:4]:
        print(k, [(b, simd[b].tolist()) for b in per_cu[k]], flush=True)


if __name__ == "__main__":
    main()
