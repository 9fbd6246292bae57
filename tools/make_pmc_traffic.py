#!/usr/bin/env python3
"""tools/make_pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <out.json> [<sq1_summary.txt> <mix1_summary.txt>] -- HBM bytes per launch of the
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


# issue cost of one wave64 vector instruction by class, in clocks of its SIMD (tools/valu_probe.hip,
# profiles/r02_valu_probe.txt): f32 2.1, packed-f32 and f64 4.2, transcendental ~8.4
COST = {"f32": 2.1, "f64": 4.2, "trans": 8.4}


def parse_max(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"sea::(\w+)\(.*\) " + counter + r": dispatches=(\d+) mean=([\d.]+) min=([\d.]+) max=([\d.]+)", line)
        if m:
            out[m.group(1)] = float(m.group(5))
    return out


HEADLINE = "ns_denoise_pipe6_dense_kernel"  # the kernel form configs[1] runs on since round 4 (capi.hip::ns_pick_form)


def valu_issue(sq1, mix1, k=HEADLINE):
    """(vector instructions, priced issue clocks) per launch of the headline kernel from the SQ passes: every
    instruction at the f32 cost, f64 ones at the f64 cost, transcendentals at theirs.  (Packed-f32 instructions have
    no counter of their own and are priced as plain f32: the figure is a LOWER bound of the issue time.)"""
    total = parse(sq1, "SQ_INSTS_VALU").get(k)  # minimum over the dispatches = the plain launches, as for the traffic
    if not total:
        return None, None
    f64 = sum(parse(mix1, c).get(k, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
    tr32 = parse(mix1, "SQ_INSTS_VALU_TRANS_F32").get(k, 0.0)
    tr64 = parse(mix1, "SQ_INSTS_VALU_TRANS_F64").get(k, 0.0)
    clk = (total - f64 - tr32 - tr64) * COST["f32"] + f64 * COST["f64"] + (tr32 + tr64) * COST["trans"]
    return int(total), int(clk)


def main():
    import bench
    big = None
    if "--big" in sys.argv:   # --big <fetch> <write> <sq1> <mix1>: the configs[4] form's own passes
        i = sys.argv.index("--big")
        big = sys.argv[i + 1:i + 5]
        del sys.argv[i:]
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    keys = {HEADLINE: "ns_denoise_kernel_bytes_per_launch", "resynth_fused_kernel": "resynth_bytes_per_launch",
            "compceps_kernel": "compceps_bytes_per_launch", "rfft256_kernel": "rfft256_bytes_per_launch",
            "subband_kernel": "subband_bytes_per_launch", "irm_target_kernel": "irm_bytes_per_launch",
            "ns16k_pipe_kernel": "ns16k_bytes_per_launch"}
    j = {"_comment": "HBM traffic per launch on the bench.py workloads (configs[1] corpus, 1024 utterances; rfft256: 2^18 frames), "
                     "from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bytes = 2 * FETCH_KiB * 1024 + WRITE_KiB * 1024 "
                     "(gfx950 FETCH_SIZE correction, see tools/make_pmc_traffic.py); per kernel the minimum over its dispatches "
                     "(the bench's plain launches).",
         "source_stamp": bench.source_stamp(), "fetch_kib": fetch, "write_kib": write}
    for k, name in keys.items():
        if k in fetch and k in write:
            j[name] = int(2 * fetch[k] * 1024 + write[k] * 1024)
    # rfft256 runs at two sizes in the bench: the minimum over its dispatches is the cache-sized one, the maximum the 2 GiB one
    fmax, wmax = parse_max(sys.argv[1], "FETCH_SIZE"), parse_max(sys.argv[2], "WRITE_SIZE")
    if "rfft256_kernel" in fmax and "rfft256_kernel" in wmax:
        j["rfft256_2g_bytes_per_launch"] = int(2 * fmax["rfft256_kernel"] * 1024 + wmax["rfft256_kernel"] * 1024)
    # the feature chain is three launches
    afe = ("ns_denoise_pipe_fd_kernel", "afe_ceps_kernel", "afe_vad_kernel")
    if all(k in fetch and k in write for k in afe):
        j["afe_bytes_per_launch"] = int(sum(2 * fetch[k] * 1024 + write[k] * 1024 for k in afe))
    if len(sys.argv) > 5:  # SQ passes: <sq1 summary> <mix1 summary>
        insts, clk = valu_issue(sys.argv[4], sys.argv[5])
        if insts:
            j["ns_valu_insts_per_launch"] = insts
            j["ns_valu_issue_clk_per_launch"] = clk
    if big and all(os.path.exists(b) for b in big):
        kb = "ns_denoise_pipe_big_kernel"
        bf, bw = parse(big[0], "FETCH_SIZE"), parse(big[1], "WRITE_SIZE")
        if kb in bf and kb in bw:
            j["ns_big_bytes_per_launch"] = int(2 * bf[kb] * 1024 + bw[kb] * 1024)
            j["fetch_kib"][kb], j["write_kib"][kb] = bf[kb], bw[kb]
        insts, clk = valu_issue(big[2], big[3], kb)
        if insts:
            j["ns_big_valu_insts_per_launch"] = insts
            j["ns_big_valu_issue_clk_per_launch"] = clk
    with open(sys.argv[3], "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps({k: v for k, v in j.items() if k.endswith("per_launch")}))


if __name__ == "__main__":
    main()
