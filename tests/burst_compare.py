"""Run one burst through the HIP pipeline (C-ABI mfsr_burst_*) and through the CPU oracle pipeline, keeping every
frame's flow field and robustness mask, and compare the two with the flip-set classification of tests/flipset.py.

Used by the -m gpu pipeline tests and by bench.py's cpu_baseline leg (the oracle is the checker there)."""
from __future__ import annotations

import numpy as np


def run_hip(cfg, frames, device="cuda:0", per_frame=True):
    """frames: list of [H, W] int16/uint16 torch tensors (any device).  Returns numpy results."""
    import torch
    from multi_frame_super_resolution_amd.pipeline import BurstPipeline, view_as_tensor
    dev = torch.device(device)
    pipe = BurstPipeline(cfg, dev)
    dframes = [f.to(dev) for f in frames]
    flows, masks = [], []
    pipe.begin_burst()
    ref = cfg.reference
    pipe.set_reference(dframes[ref])
    for k in range(len(dframes)):
        pipe.add_frame(dframes[k], k == ref)
        if per_frame or k == len(dframes) - 1:
            flow_t, mask_t, _, _ = pipe.debug_views()
            flows.append(view_as_tensor(flow_t, 2, dev).cpu().numpy())
            masks.append(view_as_tensor(mask_t, 4, dev).cpu().numpy())
    out, out16 = pipe.finish()
    torch.cuda.synchronize()
    _, _, kp_t, trk_t = pipe.debug_views()
    res = dict(out=out.cpu().numpy(), out16=out16.cpu().numpy().view(np.uint16), img_out=pipe.img_out.cpu().numpy(),
               tw=pipe.total_weights.cpu().numpy(), flows=flows, masks=masks, flow=flows[-1], mask=masks[-1],
               kparam=view_as_tensor(kp_t, 4, dev).cpu().numpy(), tracking=view_as_tensor(trk_t, 1, dev).cpu().numpy()[..., 0])
    pipe.close()
    return res


def run_oracle(cfg, frames):
    """frames: list of [H, W] 16-bit torch tensors or numpy arrays."""
    from oracle.pipeline import OraclePipeline
    op = OraclePipeline(cfg)
    nf = [(f.cpu().numpy() if hasattr(f, "cpu") else f).view(np.uint16) for f in frames]
    img_out = np.zeros((op.hrH, op.hrW, 3), np.float32)
    tw = np.zeros_like(img_out)
    op.set_reference(nf[cfg.reference])
    flows, masks = [], []
    for k, f in enumerate(nf):
        op.add_frame(f, k == cfg.reference, img_out, tw)
        flows.append(op.flow)
        masks.append(op.mask)
    out, q = op.finish(img_out, tw)
    return dict(out=out, out16=q, img_out=img_out, tw=tw, flows=flows, masks=masks, flow=flows[-1], mask=masks[-1],
                kparam=op.kparam4, tracking=op.ref_pyr[0])


def psnr(a, b, peak=1.0):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 200.0 if mse == 0 else float(10 * np.log10(peak * peak / mse))


def classify(cfg, h, o):
    """Flip-set report of a HIP result against an oracle result (both from run_* with per-frame products)."""
    from tests.flipset import FlipSet
    fs = FlipSet(cfg)
    assert len(h["flows"]) == len(o["flows"])
    for k in range(len(h["flows"])):
        if k == cfg.reference:
            continue   # identity flow, certainty 1 on both sides
        fs.add_frame(h["flows"][k], o["flows"][k], h["masks"][k], o["masks"][k])
    fs.add_weights(h["tw"], o["tw"])
    rep = fs.report(h["out"], o["out"], h["out16"], o["out16"])
    rep["psnr_db_vs_oracle"] = round(psnr(h["out"], o["out"]), 2)
    return rep


def assert_parity(rep, what, max_flip_fraction=2e-2):
    """The +-1 LSB contract: outside the flip set NO 8-bit sample is off by more than 1 LSB (the CLI's output depth,
    multi_frame_sr.cpp:207); the flip set is small; inside it the error is bounded by what one mis-rounded tap can do."""
    print(f"[{what}] PSNR vs oracle {rep['psnr_db_vs_oracle']:.1f} dB; flip set {rep['flip_fraction']:.2e} of the pixels "
          f"(max flow diff {rep['max_flow_diff_px']:.1e} px); >1 LSB 8-bit: outside {rep['n_gt1_8bit_outside']} "
          f"(max {rep['max8_outside']}), inside {rep['n_gt1_8bit_inside']} (max {rep['max8_inside']}); "
          f"16-bit: max outside {rep['max16_outside']}, frac >1 outside {rep['frac_gt1_16bit_outside']:.2e}, "
          f"max inside {rep['max16_inside']}; causes/frame {rep['flips_by_cause_per_frame']}")
    assert rep["psnr_db_vs_oracle"] >= 70.0
    assert rep["n_gt1_8bit_outside"] == 0, "a sample outside the flip set is off by more than 1 LSB"
    assert rep["max8_outside"] <= 1
    assert rep["flip_fraction"] <= max_flip_fraction
    assert rep["frac_gt1_8bit"] <= 1e-3
