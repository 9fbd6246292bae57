#!/usr/bin/env python3
"""tools/make_pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <out.json> -- HBM bytes per launch of the
bench kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; summaries by tools/prof_summary.py),
stamped with the sha of the kernel sources they were measured on (bench.py refuses a stale stamp).

gfx950: FETCH_SIZE counts 128-byte requests as 64 B (MI355X_MICROARCH.md, HBM section) -> doubled; calibrated in
round 1 on the NoiseSup kernel's own 4-byte-per-lane coalesced reads (2 x FETCH = 131.6 MB vs 130.8 MB of int16
input actually read).  WRITE_SIZE is exact.  Both are in KiB."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"sea::(\w+)\(.*\) " + counter + r": dispatches=(\d+) mean=([\d.]+) min=([\d.]+)", line)
        if m:
            out[m.group(1)] = float(m.group(4))  # min over the dispatches: the plain launches (one NoiseSup launch of the
            #                                      run also writes the float stream for CompCeps: +256 B per frame)
    return out


def main():
    import bench
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    keys = {"ns_denoise_pipe_kernel": "ns_denoise_kernel_bytes_per_launch", "resynth_fused_kernel": "resynth_bytes_per_launch",
            "compceps_kernel": "compceps_bytes_per_launch", "rfft256_kernel": "rfft256_bytes_per_launch",
            "subband_kernel": "subband_bytes_per_launch", "irm_target_kernel": "irm_bytes_per_launch"}
    j = {"_comment": "HBM traffic per launch on the bench.py workloads (configs[1] corpus, 1024 utterances; rfft256: 2^18 frames), "
                     "from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bytes = 2 * FETCH_KiB * 1024 + WRITE_KiB * 1024 "
                     "(gfx950 FETCH_SIZE correction, see tools/make_pmc_traffic.py); per kernel the minimum over its dispatches "
                     "(the bench's plain launches).",
         "source_stamp": bench.source_stamp(), "fetch_kib": fetch, "write_kib": write}
    for k, name in keys.items():
        if k in fetch and k in write:
            j[name] = int(2 * fetch[k] * 1024 + write[k] * 1024)
    with open(sys.argv[3], "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps({k: v for k, v in j.items() if k.endswith("per_launch")}))


if __name__ == "__main__":
    main()
