#!/usr/bin/env python3
"""tools/kernel_resources.py [file.hip ...] -- register / LDS / scratch use of every gfx950 kernel, from the
compiler's own -Rpass-analysis=kernel-resource-usage remarks (cross-compiles, no GPU needed).

    python tools/kernel_resources.py                      # every .hip under speech_enhancement_amd/csrc
    python tools/kernel_resources.py ns_pipe_kernel.hip -DSEA_NS_MIN_WAVES=4
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech_enhancement_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-fno-gpu-flush-denormals-to-zero",
         "-fno-slp-vectorize", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
        "LDS Size [bytes/block]"]


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith(".hip")]
    extra = [a for a in args if not a.endswith(".hip")]
    if not files:
        files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'kernel':58s} " + " ".join(f"{k.split(' ')[0][:10]:>10s}" for k in KEYS))
    for f in files:
        path = f if os.path.exists(f) else os.path.join(CSRC, f)
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [path, "-o", "/dev/null"], capture_output=True, text=True)
        cur, vals = None, {}
        for line in r.stderr.splitlines():
            m = re.search(r"remark: .*?:\d+:\d+: (.*?) \[-Rpass-analysis", line) or re.search(r"remark: (.*?) \[-Rpass-analysis", line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                if cur:
                    emit(cur, vals)
                cur, vals = t.split(":", 1)[1].strip(), {}
            else:
                for k in KEYS:
                    if t.startswith(k + ":"):
                        vals[k] = t.split(":")[-1].strip()
        if cur:
            emit(cur, vals)
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(r.returncode)


def emit(name, vals):
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(.*", "", name)
    print(f"{name[:58]:58s} " + " ".join(f"{vals.get(k, '-'):>10s}" for k in KEYS))


if __name__ == "__main__":
    main()
