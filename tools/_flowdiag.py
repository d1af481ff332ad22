import sys, numpy as np
sys.path.insert(0, "/root/repo")
from multi_frame_super_resolution_amd.pipeline import default_config
from multi_frame_super_resolution_amd.synth import make_burst
from tests.burst_compare import run_hip, run_oracle, flow_conditioning
W,H,N,s,mono = 1920,1080,3,2,True
frames,shifts,_ = make_burst(W,H,N,scale=s,mono=mono,seed=1236,max_shift=4.0)
cfg = default_config(W,H,N,s,mono)
h = run_hip(cfg, frames); o = run_oracle(cfg, frames)
s1, s2 = flow_conditioning(o["tracking"], cfg.lkHalfWindow)
print("lkHalfWindow", cfg.lkHalfWindow, "iters", cfg.lkIterations, "shifts", shifts)
for k in range(1,N):
    fh, fo = h["flows"][k], o["flows"][k]
    d = np.abs(fh-fo).max(-1)
    med = np.median(fo.reshape(-1,2),0)
    dev = np.abs(fo-med).max(-1)      # how far the oracle's flow is from the global shift (pure translation burst)
    yy, xx = np.mgrid[0:d.shape[0], 0:d.shape[1]]
    border = np.minimum(np.minimum(yy, d.shape[0]-1-yy), np.minimum(xx, d.shape[1]-1-xx))
    big = d > 1e-4
    print(f"frame {k}: n big {big.sum()}, border dist there: p50 {np.percentile(border[big],50)} p90 {np.percentile(border[big],90)}; "
          f"flow deviation from the median there: p50 {np.percentile(dev[big],50):.3f} p90 {np.percentile(dev[big],90):.3f}; elsewhere p50 {np.percentile(dev[~big],50):.4f} p99 {np.percentile(dev[~big],99):.4f}")
    idx = np.argsort(d.ravel())[::-1][:12]
    for i in idx:
        y, x = divmod(i, d.shape[1])
        print(f"   ({y},{x}) d {d[y,x]:.2e} hip {fh[y,x]} orc {fo[y,x]} s1 {s1[y,x]:.3e} s2 {s2[y,x]:.3e} border {border[y,x]} dev {dev[y,x]:.3f}")
    for lo, hi in ((0,0.01),(0.01,0.05),(0.05,0.2),(0.2,1),(1,100)):
        m = (dev>=lo)&(dev<hi)
        if m.any(): print(f"   dev in [{lo},{hi}): n {m.sum()} max d {d[m].max():.2e} p99 d {np.percentile(d[m],99):.2e}")

from tests.burst_compare import flow_difference_report
for k in range(1,N):
    for thr in (1e-4, 2e-4, 3e-4):
        print(k, thr, flow_difference_report(h["flows"][k], o["flows"][k], o["tracking"], cfg.lkHalfWindow, thr))
