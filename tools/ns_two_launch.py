#!/usr/bin/env python3
"""tools/ns_two_launch.py -- (GPU box) experiment: configs[1] as TWO concurrent launches on two streams -- the n_top longest
utterances in one kernel form, the rest in another -- against the one launch of the four-wave form.  Outputs compared
exactly; times by HIP events around both launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import speech_enhancement_amd as sea  # noqa: E402


def main():
    lib = sea.load()
    dev = torch.device("cuda", 0)
    batch = bench.build_shard_ids(list(range(1024)), dev)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    lens = np.asarray(batch.host_lengths)
    by_len = np.argsort(-lens, kind="stable")
    ref = torch.zeros_like(batch.data)
    lib.sea_ns_kernel_form(0)
    sea.ns_denoise_batch(batch, out=ref)
    torch.cuda.synchronize()

    def timed(fn, reps=7):
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]

    out = torch.zeros_like(batch.data)
    print(f"one launch, four-wave form: {timed(lambda: sea.ns_denoise_batch(batch, out=out)):.3f} ms")
    side = torch.cuda.Stream()
    ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()
    for n_top in (n_cu, n_cu // 2, 2 * n_cu):
        for form_top, form_rest in ((3, 4), (3, 2), (2, 4), (2, 2)):
            top, rest = by_len[:n_top], by_len[n_top:]
            sets = []
            for idx in (top, rest):
                # launch order within a set: longest first, every other row of n_cu reversed (as engine.launch_order)
                order = torch.arange(len(idx), dtype=torch.int32, device=dev)
                sets.append((torch.from_numpy(batch.host_offsets[idx].astype(np.int64)).to(dev),
                             torch.from_numpy(lens[idx].astype(np.int64)).to(dev), order, len(idx)))
            out.zero_()

            def run():
                main_s = torch.cuda.current_stream()
                ev0.record(main_s)
                side.wait_event(ev0)
                o, l, od, n = sets[0]
                lib.sea_ns_kernel_form(form_top)
                assert lib.sea_ns_denoise_batch(batch.data.data_ptr(), out.data_ptr(), None, o.data_ptr(), l.data_ptr(), od.data_ptr(), None, n, main_s.cuda_stream) == 0
                o, l, od, n = sets[1]
                lib.sea_ns_kernel_form(form_rest)
                assert lib.sea_ns_denoise_batch(batch.data.data_ptr(), out.data_ptr(), None, o.data_ptr(), l.data_ptr(), od.data_ptr(), None, n, side.cuda_stream) == 0
                ev1.record(side)
                main_s.wait_event(ev1)
            ms = timed(run)
            ok = torch.equal(out, ref)
            print(f"top {n_top} in form {form_top} | rest {len(rest)} in form {form_rest}: {ms:.3f} ms  identical={ok}")
    lib.sea_ns_kernel_form(0)


if __name__ == "__main__":
    main()
