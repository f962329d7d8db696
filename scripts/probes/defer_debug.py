#!/usr/bin/env python
"""which gradient slots differ between deferred folds (ParamStore.defer_wgrad) and per-consumer writes on the toy MemVLA fixture"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_memvla_gpu import build, T  # noqa: E402
from tests.helpers import rel_err  # noqa: E402
import dexbotic_amd.functional as Fn  # noqa: E402
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def run(defer, no_bg):
    Fn._NO_DEFER_BGRAD = no_bg
    g, cfg, m = build(gd, "float32", True)
    m.train()
    st = m.store
    st.defer_wgrad = defer
    st.set_expected(m.unused_parameter_names())
    st.begin_step()
    out = m(input_ids=T(g["input_ids"]), attention_mask=T(g["attention_mask"]), images=T(g["images"]), actions=T(g["actions"]),
            indexes=[list(map(int, r)) for r in g["indexes"]], noise=T(g["noise"]), timesteps=T(g["timesteps"]),
            drop_ids=T(g["drop_u"]) < 0.1)
    out.loss.backward()
    print("stash left:", list(st._wg_stash), list(st._bg_stash))
    st.flush_wgrads()
    torch.cuda.synchronize()
    return {n: st.g(n).float().cpu().numpy().copy() for n in st.slots if st.grad_written[n]}, g


base, g = run(False, False)
for label, (defer, no_bg) in {"dW deferred only": (True, True), "dW + bias / LN deferred": (True, False)}.items():
    got, _ = run(defer, no_bg)
    d = sorted(((rel_err(got[n], base[n]), n) for n in base if np.abs(base[n]).max() > 0), reverse=True)
    print(label, "worst:")
    for e, n in d[:8]:
        ref = g["grad/" + n] if "grad/" + n in g.files else None
        wn = n.replace(".bias", ".weight")
        print(f"   {e:.3e}  {n}   max|diff| {np.abs(got[n] - base[n]).max():.3e}  max|db| {np.abs(base[n]).max():.3e}  max|dW| {np.abs(base[wn]).max():.3e}  |db|2 {np.linalg.norm(base[n]):.3e}")
