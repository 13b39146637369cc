"""Where a wave of k_fill3 spends its life: shader-clock spans summed in the DBG build (SITATOR_DEBUG_STOP = 10 / 11 / 12),
averaged per wave: scratch/phase_times.py [config] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SITATOR_FILL_AUTOTUNE"] = "0"
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
host = synth.config_host(cfg)
ctx, *_ = _setup(host, synth.CONFIG_MOBILE[cfg], F, seed=2)
for _ in range(30):
    ctx.fill(check_for_zeros=False)
out = {}
for mode in (11, 10, 12):
    os.environ["SITATOR_DEBUG_STOP"] = str(mode)
    for _ in range(3):
        rc, nz, err = ctx.fill(check_for_zeros=False)
    out[mode] = ctx.info()["census"]
    print("mode", mode, "fill ms %.4f" % ctx.timers()["fill"], out[mode])
os.environ.pop("SITATOR_DEBUG_STOP")
nw = out[11][3]
names = {10: ["start -> barrier A passed", "A -> barrier B passed (phase 1b)", "B -> window set-up done", "D0 passes"],
         11: ["D1 + E passes", "T + end of window", "whole wave", None],
         12: ["start -> frame requested, at barrier A", "wait at barrier A", "phase 1b work", "wait at barrier B"]}
print("waves %d; shader-clock cycles per wave (s_memtime, 100 MHz ticks x clock ratio = cycles as the SQ counts them):" % nw)
for mode in (10, 11, 12):
    for n, v in zip(names[mode], out[mode]):
        if n:
            print("  %-44s %9.0f" % (n, v / nw))
