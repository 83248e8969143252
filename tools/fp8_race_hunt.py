"""Two runs of two MX-fp8 training steps from the same state: are the losses bit-identical?  (a stream race of the
fp8 mode's overlapped schedule; the environment selects the schedule and the XMC_FP8_DEBUG bits under test.)
    XMC_FP8_OVERLAP=1 XMC_PREFETCH_G=1 XMC_OVERLAP_BWD=0 XMC_OVERLAP_PREP=0 python tools/fp8_race_hunt.py [--config c3 --batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--keep-alive", action="store_true", help="no tensor allocated during a step is freed before its end: "
                    "separates a use-after-free through the caching allocator from a missing stream dependency")
    ap.add_argument("--sync-after", default="", help="comma list of HipOps methods followed by a device synchronisation")
    ap.add_argument("--fp8", type=int, default=1)
    ap.add_argument("--trace-inputs", action="store_true", help="with --trace-ops: also checksum every input of a conv call")
    ap.add_argument("--trace-ops", action="store_true", help="checksum every HipOps result on its own stream (no host "
                    "synchronisation) and report the first call whose bytes differ between run 1 and run 2")
    a = ap.parse_args()
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_c1_config() if a.config == "c1" else coco_xmc.get_c3_config()
    cfg.batch_size = a.batch
    cfg.conv_fp8 = bool(a.fp8)
    cfg.pretrained_image_contrastive = False

    from xmcgan_image_generation_amd.ops import HipOps
    for name in [m for m in a.sync_after.split(",") if m]:
        orig = getattr(HipOps, name)

        def wrapped(self, *args, __orig=orig, **kw):
            out = __orig(self, *args, **kw)
            torch.cuda.synchronize()
            return out
        setattr(HipOps, name, wrapped)
    sums = []
    if a.trace_ops:
        import inspect

        def cks(t):
            if not (torch.is_tensor(t) and t.is_cuda and t.numel()):
                return None
            if t.dtype == torch.uint8 and t.dim() == 3 and t.shape[-1] == 80:
                t = t[..., :66]                                   # MX packets: the 14 pad bytes are never written
            v = t.detach().contiguous().view(-1)
            v = v.view(torch.int16) if v.element_size() == 2 else (v.view(torch.int32) if v.element_size() == 4 else v.view(torch.uint8))
            return v.to(torch.int64).sum()
        for name, orig in inspect.getmembers(HipOps, predicate=inspect.isfunction):
            if isinstance(inspect.getattr_static(HipOps, name), (staticmethod, classmethod)):
                continue
            if name.startswith("_") or name in ("side", "join_side", "empty", "zeros", "begin_pool", "record_event", "wait_event",
                                                "join_wgrad", "zeros_act", "wl_fused_ok"):
                continue

            def traced(self, *args, __orig=orig, __name=name, **kw):
                ins = []
                if __name == "conv" and a.trace_inputs:
                    w = args[1]
                    cand = [("x", args[0]), ("mask", kw.get("mask")), ("res", kw.get("res")), ("alpha_dev", kw.get("alpha_dev")),
                            ("w.data", getattr(w, "data", None) if hasattr(w, "mx8") else w)]
                    m8 = getattr(w, "mx8", None)
                    if m8 is not None:
                        cand += [("w.mx8.q", m8[0]), ("w.mx8.s", m8[1])]
                    xm = getattr(args[0], "mx8", None)
                    if xm is not None:
                        cand.append(("x.mx8", xm[0]))
                    ins = [(k, cks(v)) for k, v in cand if v is not None]
                out = __orig(self, *args, **kw)
                outs = out if isinstance(out, (tuple, list)) else (out,)
                cs = [cks(o) for o in outs]
                if __name == "conv_wgrad":
                    cs = [cks(args[2]), cks(args[3] if len(args) > 3 else kw.get("db"))]
                if __name.startswith("adam"):
                    cs = [cks(args[0])]
                shapes = [tuple(o.shape) if torch.is_tensor(o) else None for o in outs]
                if __name == "conv":
                    w = args[1]
                    shapes = [f"x {tuple(args[0].shape)} -> {shapes[0]}", "w " + (f"packed mx8={'yes' if getattr(w, 'mx8', None) is not None else 'no'} "
                              f"data={'yes' if getattr(w, 'data', None) is not None else 'NONE'} phase={'yes' if getattr(w, 'phase', None) is not None else 'no'}"
                              if hasattr(w, "mx8") else f"plain {tuple(w.shape)}"),
                              {k: (v if not torch.is_tensor(v) else "T") for k, v in kw.items() if v is not None and v is not False},
                              "phase" if getattr(self, "last_conv_phase", False) else "3x3/1x1", "x.mx8" if getattr(args[0], "mx8", None) is not None else ""]
                if ins:
                    sums.append((__name + ":inputs", [k for k, _ in ins], [c for _, c in ins], torch.cuda.current_stream().cuda_stream))
                sums.append((__name, shapes, [c for c in cs if c is not None], torch.cuda.current_stream().cuda_stream))
                return out
            setattr(HipOps, name, traced)
    keep = []
    if a.keep_alive:
        for fn in ("empty", "empty_like", "zeros", "zeros_like"):
            o = getattr(torch, fn)

            def alloc(*x, __o=o, **k):
                t = __o(*x, **k)
                if t.is_cuda:
                    keep.append(t)
                return t
            setattr(torch, fn, alloc)

    def run():
        del keep[:]
        del sums[:]
        torch.manual_seed(0)
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=a.batch).items()}
        outs = []
        for i in range(2):
            state, m = train_utils.train_step(i, state, tb, xmc_gan, gen, disc, cfg, {})
            outs.append({k: float(v) for k, v in m.items()})
        torch.cuda.synchronize()
        if a.trace_ops:
            outs.append({"trace": [(n, sh, [int(c) for c in cs], st) for n, sh, cs, st in sums]})
        return outs
    runs = [run() for _ in range(a.runs)]
    if a.trace_ops:
        traces = [r.pop()["trace"] for r in runs]
        ref = traces[0]
        streams = sorted({t[3] for t in ref})
        for k, tr in enumerate(traces[1:], 2):
            assert len(tr) == len(ref)
            first = [i for i, (x, y) in enumerate(zip(ref, tr)) if x[2] != y[2]]
            print(f"run {k} vs run 1: {len(first)} of {len(ref)} calls differ" + (f"; first at call {first[0]}" if first else ""))
            real = [i for i in first if ref[i][0] not in ("wprep_run", "sn_bank_power_iter_fused", "wprep_create")]
            if real:
                i0 = real[0]
                for i in range(max(0, i0 - 14), min(len(ref), i0 + 6)):
                    n, sh, cs, st = ref[i]
                    extra = ""
                    if n.endswith(":inputs") and i in first:
                        extra = "   differing inputs: " + ", ".join(k for k, x, y in zip(sh, cs, tr[i][2]) if x != y)
                    print(f"    call {i:5d} {'DIFF' if i in first else '    '} {n:24s} stream {streams.index(st)} shapes {sh}{extra}")
    same = all(r == runs[0] for r in runs[1:])
    tag = " ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("XMC_"))
    tag += (" keep-alive" if a.keep_alive else "") + (f" sync-after={a.sync_after}" if a.sync_after else "") + ("" if a.fp8 else " bf16")
    print(f"[{tag}] identical over {a.runs} runs: {same}", flush=True)
    if not same:
        for i in range(2):
            for k in runs[0][i]:
                vals = [r[i][k] for r in runs]
                if len(set(vals)) > 1:
                    print(f"   step {i} {k:10s} " + "  ".join(f"{v:.8g}" for v in vals))


if __name__ == "__main__":
    main()
