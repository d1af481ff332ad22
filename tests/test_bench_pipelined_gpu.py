"""bench.py's pipelined step loop (two burst contexts, exchange/finish on a side stream) must give
the same u16 frame as the plain loop: same inputs, same kernels, only the stream schedule differs."""
import json
import os
import subprocess
import sys

import pytest
import torch

from multi_frame_super_resolution_amd import distributed as mdist
from multi_frame_super_resolution_amd import synth
from multi_frame_super_resolution_amd.pipeline import BurstPipeline, default_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_contexts_side_stream_same_result():
    dev = torch.device("cuda:0")
    cfg = default_config(512, 384, 4, scale=2)
    frames, _, _ = synth.make_burst(512, 384, 4, seed=5, device=dev)
    ref_pipe = BurstPipeline(cfg, dev)
    expect = mdist.process_burst(ref_pipe, frames, n_frames=4).clone()
    pipes = [BurstPipeline(cfg, dev), BurstPipeline(cfg, dev)]
    side = torch.cuda.Stream(device=dev)
    ev_acc = [torch.cuda.Event(), torch.cuda.Event()]
    ev_done = [torch.cuda.Event(), torch.cuda.Event()]
    outs = []
    main = torch.cuda.current_stream()
    for i in range(6):
        j = i % 2
        if i >= 2:
            main.wait_event(ev_done[j])
        mdist.accumulate_local(pipes[j], frames, 0, 1, 4)
        ev_acc[j].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_acc[j])
            _, o = pipes[j].finish(want_float=False, want_u16=True)
            outs.append(o.clone())
            ev_done[j].record(side)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, expect)
    for p in pipes + [ref_pipe]:
        p.close()


@pytest.mark.gpu
def test_bench_force_pipelined_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "1080p5_gray_x2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--force-pipelined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["roofline"]["launches_timed"] > 0
