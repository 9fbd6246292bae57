#!/usr/bin/env python3
"""tools/soak.py [iterations] -- determinism soak on the GPU box: every hot-path kernel run repeatedly on the bench
corpus (configs[1]), every output compared bit for bit with the first run's (which the -m gpu tests compare with the
oracle).  A race in the role pipelines' LDS hand-overs would show here as a differing iteration.  Also alternates
the NoiseSup kernel forms and runs two launches concurrently on two streams to vary the timing."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import speech_enhancement_amd as sea
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(sea.LIB_PATH)
    ids = list(range(1024))
    batch = bench.build_shard_ids(ids, dev)
    masks = bench.build_masks(batch, ids, dev)
    res = {}
    t0 = time.time()

    # NoiseSup: all forms, int16 + float stream + first_out
    ref = None
    bad = 0
    for k in range(it):
        form = (6, 2, 4, 3, 6)[k % 5]        # dense six-wave (the configs[1] form), pipe, big, pipe6, dense
        lib.sea_ns_kernel_form(form)
        out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
        if ref is None:
            ref = (out.clone(), f32.clone(), first.clone())
        else:
            bad += int(not (torch.equal(out, ref[0]) and torch.equal(f32.view(torch.int32), ref[1].view(torch.int32)) and torch.equal(first, ref[2])))
    lib.sea_ns_kernel_form(0)
    res["noisesup_iterations"] = it
    res["noisesup_differing"] = bad

    # two concurrent launches on two streams (different occupancy / timing), same expected result
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    o1, o2 = torch.zeros_like(batch.data), torch.zeros_like(batch.data)
    for k in range(max(it // 4, 5)):
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            sea.ns_denoise_batch(batch, out=o1)
        with torch.cuda.stream(s2):
            sea.ns_denoise_batch(batch, out=o2)
        torch.cuda.synchronize()
        bad += int(not (torch.equal(o1, ref[0]) and torch.equal(o2, ref[0])))
    res["noisesup_concurrent_pairs_differing"] = bad
    print(json.dumps(res), flush=True)

    # CompCeps and the AFE chain
    cref = None
    bad = 0
    for k in range(max(it // 4, 5)):
        ceps, cum, n_ceps = sea.compceps_batch(batch, ref[1], ref[2])
        if cref is None:
            cref = (ceps.clone(), n_ceps.clone())
        else:
            bad += int(not (torch.equal(ceps.view(torch.int32), cref[0].view(torch.int32)) and torch.equal(n_ceps, cref[1])))
    res["compceps_differing"] = bad
    import numpy as np
    aref = None
    bad = 0
    for k in range(max(it // 10, 3)):
        r = sea.afe_features_batch(batch)
        cat = np.concatenate([f.ravel() for f in r["feats"]]).view(np.int32)
        if aref is None:
            aref = cat
        else:
            bad += int(not np.array_equal(cat, aref))
    res["afe_chain_differing"] = bad
    print(json.dumps(res), flush=True)

    # resynthesis (ratio mask, binary mask), subbband, IRM target
    for name, binary in (("resynth_ratio", False), ("resynth_ibm", True)):
        rref = None
        bad = 0
        for k in range(max(it // 10, 3)):
            out, scratch = sea.resynth_batch(batch, masks, binary=binary)
            del scratch
            if rref is None:
                rref = out.clone()
            else:
                bad += int(not torch.equal(out, rref))
        res[name + "_differing"] = bad
        del rref
    sref = None
    bad = 0
    for k in range(max(it // 10, 3)):
        sub = sea.subband_batch(batch)
        if sref is None:
            sref = sub.clone()
        else:
            bad += int(not torch.equal(sub, sref))
    res["subband_differing"] = bad
    iref = None
    bad = 0
    for k in range(3):
        irm = sea.irm_target_batch(batch, sref, sref).data
        if iref is None:
            iref = irm.clone()
        else:
            bad += int(not torch.equal(irm.view(torch.int32), iref.view(torch.int32)))
    res["irm_differing"] = bad
    # the 16 k-native NoiseSup variant: 2048 streams (two waves per SIMD) and 1024 (one), chunked differently each time
    gen = torch.Generator(device=dev)
    gen.manual_seed(16)
    fr = torch.randint(-6000, 6000, (2048, 120, 160), device=dev, generator=gen).float()
    fr[::7, 40:44] = 0.0                      # gated frames in every seventh stream
    nref = None
    bad = 0
    for k in range(max(it // 10, 3)):
        cut = (120, 60, 1, 37)[k % 4]
        r = sea.ns16k_streams_push(fr[:, :cut].contiguous())
        parts = [r]
        if cut < 120:
            parts.append(sea.ns16k_streams_push(fr[:, cut:].contiguous(), state=r["state"]))
        cat = [torch.cat([p[key] for p in parts], dim=1) for key in ("out", "produced", "flags", "counter", "wiener")]
        if nref is None:
            nref = [c.clone() for c in cat]
        else:
            bad += int(not all(torch.equal(a.view(torch.uint8) if a.dtype == torch.uint8 else a.view(torch.int32), b.view(torch.uint8) if b.dtype == torch.uint8 else b.view(torch.int32))
                               for a, b in zip(cat, nref)))
    res["ns16k_differing"] = bad
    res["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(res), flush=True)
    sys.exit(1 if any(v for k, v in res.items() if k.endswith("differing")) else 0)


if __name__ == "__main__":
    main()
