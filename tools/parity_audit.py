#!/usr/bin/env python3
"""Where do the HIP and the oracle pipeline differ OUTSIDE the flip set, and why?  (diagnostic behind tests/burst_compare.py)

  * accumulators: relative differences of imgOut / totalWeights, and for the u16 samples that differ by more than one LSB16
    the total weight of their channel;
  * flow: where |flow_hip - flow_oracle| is large, the conditioning of the Lucas-Kanade window there (the reference only
    tests the LARGER singular value against lkMinDet, opticalFlow.cu:255 -- the smaller one may be ~0).

usage: python tools/parity_audit.py WxH N scale [mono]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from multi_frame_super_resolution_amd.pipeline import default_config
    from multi_frame_super_resolution_amd.synth import make_burst
    from tests.burst_compare import classify, flow_conditioning, run_hip, run_oracle
    from tests.flipset import FlipSet

    W, H = (int(v) for v in sys.argv[1].split("x"))
    N, s = int(sys.argv[2]), int(sys.argv[3])
    mono = len(sys.argv) > 4 and sys.argv[4] == "mono"
    frames, _, _ = make_burst(W, H, N, scale=s, mono=mono, seed=1234 + 2, max_shift=4.0)
    cfg = default_config(W, H, N, s, mono)
    h = run_hip(cfg, frames)
    o = run_oracle(cfg, frames)
    for eps in (4e-6, 1e-6, 2.5e-7, 0.0):
        fs = FlipSet(cfg, tie_eps=eps)
        for k in range(N):
            if k != cfg.reference:
                fs.add_frame(h["flows"][k], o["flows"][k], h["masks"][k], o["masks"][k])
        fs.add_weights(h["tw"], o["tw"])
        rep = fs.report(h["out"], o["out"], h["out16"], o["out16"])
        print(f"tie_eps {eps:.1e}: flip {rep['flip_fraction']:.3e}  8-bit >1 outside {rep['n_gt1_8bit_outside']} max8 out {rep['max8_outside']} "
              f"max16 out {rep['max16_outside']} frac16>1 out {rep['frac_gt1_16bit_outside']:.2e}  causes {rep['flips_by_cause_per_frame']}")
    rep = classify(cfg, h, o)
    print({k: v for k, v in rep.items() if k not in ("flips_by_cause_per_frame", "_flips")})
    E = rep["_flips"]
    out = ~E
    for name in ("img_out", "tw"):
        a, b = h[name].astype(np.float64), o[name].astype(np.float64)
        rel = np.abs(a - b) / (np.abs(b) + 1e-30)
        ab = np.abs(a - b)
        r = rel[out]
        print(f"{name}: outside flip set: rel diff p50 {np.percentile(r, 50):.2e} p99 {np.percentile(r, 99):.2e} p99.99 {np.percentile(r, 99.99):.2e} "
              f"max {r.max():.2e}; abs max {ab[out].max():.2e}; frac rel>3e-5 & abs>1e-6: {np.mean((r > 3e-5) & (ab[out] > 1e-6)):.2e}")
    d16 = np.abs(h["out16"].astype(np.int64) - o["out16"].astype(np.int64))
    bad = (d16 > 1) & out[..., None]
    tw = o["tw"]
    print(f"u16 samples >1 LSB16 outside the flip set: {int(bad.sum())} of {bad.size} ({bad.mean():.2e}), max {int(d16[out].max())}")
    if bad.any():
        w = tw[bad]
        print("  total weight of those samples: min %.3e p10 %.3e p50 %.3e p90 %.3e max %.3e (weightThreshold %.3e)" % (
            w.min(), np.percentile(w, 10), np.percentile(w, 50), np.percentile(w, 90), w.max(), cfg.weightThreshold))
        for tau in (1e-3, 1e-2, 0.03, 0.1, 0.3, 1.0, 3.0):
            sel = out[..., None] & (tw >= tau)
            print(f"  tau {tau:g}: samples with tw >= tau: {sel.mean():.4f} of all; of them >1 LSB16: {int((d16[sel] > 1).sum())}, max {int(d16[sel].max())}")
        relI = (np.abs(h["img_out"] - o["img_out"]) / (np.abs(o["img_out"]) + 1e-30))[bad]
        relW = (np.abs(h["tw"] - o["tw"]) / (np.abs(o["tw"]) + 1e-30))[bad]
        print("  rel diff of their accumulators: imgOut p50 %.2e max %.2e; tw p50 %.2e max %.2e" % (
            np.percentile(relI, 50), relI.max(), np.percentile(relW, 50), relW.max()))
        vo = o["out"][bad]
        print("  their output value (oracle): p10 %.4f p50 %.4f p90 %.4f" % (np.percentile(vo, 10), np.percentile(vo, 50), np.percentile(vo, 90)))
    # conditioning of the tap exponents: px^2 kx + 2 px py kz + py^2 ky is a sum of terms of magnitude up to
    # kappa = 4|kx| + 4|ky| + 8|kz| that may cancel; every fp32 rounding of a term is an ABSOLUTE error eps*kappa of the
    # exponent = a RELATIVE error of the weight
    from tests.burst_compare import exponent_conditioning
    kap = exponent_conditioning(o["kparam"], h["img_out"].shape[0], h["img_out"].shape[1])
    eps = 2.0 ** -24
    for name in ("img_out", "tw"):
        a, b = h[name].astype(np.float64), o[name].astype(np.float64)
        rel = np.abs(a - b) / (np.abs(b) + 1e-30)
        sig = (np.abs(a - b) > 1e-6) & out[..., None]
        ratio = (rel / (eps * np.maximum(kap[..., None], 1.0)))[sig]
        print(f"{name}: rel diff / (eps*kappa) over the samples with abs diff > 1e-6 outside the set: p50 {np.percentile(ratio, 50):.2f} p99 {np.percentile(ratio, 99):.2f} "
              f"p99.99 {np.percentile(ratio, 99.99):.2f} max {ratio.max():.2f}")
    # the kernel parameters themselves differ (E3's eigen-decomposition is ill-conditioned where the tensor is nearly
    # isotropic): a difference dk changes every exponent by up to 0.5 * (4|dkx| + 4|dky| + 8|dkz|)
    dkap = 0.5 * exponent_conditioning(h["kparam"].astype(np.float64) - o["kparam"].astype(np.float64), h["img_out"].shape[0], h["img_out"].shape[1])
    print("exponent difference bound from the kernel-parameter differences: p50 %.2e p99 %.2e p99.99 %.2e max %.2e" % tuple(np.percentile(dkap, [50, 99, 99.99, 100])))
    for name in ("img_out", "tw"):
        a, b = h[name].astype(np.float64), o[name].astype(np.float64)
        rel = np.abs(a - b) / (np.abs(b) + 1e-30)
        sig = (np.abs(a - b) > 1e-6) & out[..., None]
        bound = 3e-5 + dkap[..., None] + 8 * eps * kap[..., None]
        ratio = (rel / bound)[sig]
        print(f"{name}: rel diff / (3e-5 + dkappa + 8 eps kappa): p50 {np.percentile(ratio, 50):.3f} p99 {np.percentile(ratio, 99):.3f} p99.99 {np.percentile(ratio, 99.99):.3f} max {ratio.max():.3f}; "
              f"n > 1: {int((ratio > 1).sum())} of {ratio.size}")
    print("kappa: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(kap, [50, 90, 99, 100])))
    for K0 in (30, 100, 300, 1000, 3000):
        for tau in (0.01, 0.1):
            sel = out[..., None] & (tw >= tau) & (kap[..., None] <= K0)
            if sel.any():
                print(f"  kappa <= {K0}, tw >= {tau}: {sel.mean():.4f} of the samples; >1 LSB16: {int((d16[sel] > 1).sum())}, max {int(d16[sel].max())}")
    # flow
    for k in range(N):
        if k == cfg.reference:
            continue
        d = np.abs(h["flows"][k] - o["flows"][k]).max(-1)
        s1, s2 = flow_conditioning(o["tracking"], cfg.lkHalfWindow)
        big = d > 1e-4
        print(f"frame {k}: max |dflow| {d.max():.2e}; px with |d|>1e-4: {int(big.sum())} ({big.mean():.2e})")
        if big.any():
            c = s2 / np.maximum(s1, 1e-30)
            print("   there: sigma2/sigma1 p50 %.2e p90 %.2e max %.2e ; sigma2 p50 %.2e max %.2e | elsewhere sigma2/sigma1 p1 %.2e p50 %.2e ; lkMinDet %.2e" % (
                np.percentile(c[big], 50), np.percentile(c[big], 90), c[big].max(), np.percentile(s2[big], 50), s2[big].max(),
                np.percentile(c[~big], 1), np.percentile(c[~big], 50), cfg.lkMinDet))
            for thr in (1e-2, 3e-2, 1e-1):
                print(f"   |d|>1e-4 with sigma2/sigma1 > {thr:g}: {int((big & (c > thr)).sum())}; max |d| where ratio > thr: {d[c > thr].max():.2e}")


if __name__ == "__main__":
    main()
