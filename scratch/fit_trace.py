"""Step trace of the speculative fit (SITATOR_FF_TRACE): scratch/fit_trace.py [config] [frames]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tr = "/tmp/ff_trace.txt"
for _f in (tr, tr + ".found"):
    if os.path.exists(_f):
        os.remove(_f)
os.environ["SITATOR_FF_TRACE"] = tr
os.environ["SITATOR_PIPELINE"] = "0"
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg])
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False); st = la.run(sn, frames)
rows = [tuple(int(x) for x in l.split()) for l in open(tr)]
print("steps", len(rows), "sites", st.site_network.n_sites, la._ctx.info()["fit_batches"], la._ctx.info()["fit_rewalks"])
kinds = collections.Counter()
import json
out = []
for i, (pos, nb, fn, fb, lognf, kd) in enumerate(rows):
    nnew = fb >> 32
    fb = fb & 0xffffffff
    fb = fb - (1 << 32) if fb >= (1 << 31) else fb
    nf, logn = lognf >> 32, lognf & 0xffffffff
    out.append((pos, nb, nnew, nf))
    K, dec, vdec = kd & 0xffffff, (kd >> 24) & 0xffffff, (kd >> 48) & 0xffffff
    sx = lambda v: v - (1 << 24) if v >= (1 << 23) else v
    dec, vdec = sx(dec), sx(vdec)
    cut = min(nb, fn if fn >= 0 else nb, fb if fb >= 0 else nb)
    kind = "clean" if cut == nb else ("unhandled-new/break" if (fn >= 0 and cut == fn) else "bad")
    if kind == "bad":
        d = "spec %s -> true %s" % ("NEW" if dec == -1 else ("tent" if dec >= K else "old"), "NEW" if vdec == -1 else ("tent" if vdec >= K else ("old" if vdec >= 0 else "break")))
        kinds[d] += 1
    kinds[kind] += 1
    if i < 40 or (kind == "bad" and kinds["bad"] < 40):
        print(i, "pos", pos, "nb", nb, "first_new", fn, "first_bad", fb, "nnew", nnew, "nfound", nf, "log", logn, "K", K, "dec", dec, "vdec", vdec, kind)
print(kinds)
if os.path.exists(tr + ".found"):
    fr = [tuple(int(x) for x in l.split()) for l in open(tr + ".found")]
    print("k_fs_found phases (cycles): step, nnew, nf | set-up, bidding, staging, scoring, kernel, rounds")
    for i, (r_, o_) in enumerate(zip(fr, out)):
        if r_[4] > 100000: print(i, o_[2], o_[3], "|", *r_)
json.dump(out, open(os.environ.get("FF_STEPS_JSON", "/tmp/ff_steps.json"), "w"))
