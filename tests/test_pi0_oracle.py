"""Pin the CPU restatement of the pi0 policy (oracle/pi0_oracle.py) against golden vectors produced by the live
reference Pi0ForCausalLM (oracle/gen_golden_pi0.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import pi0_oracle as P
from oracle.weights import make_weights, weights_crc


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def load(golden_dir):
    g = np.load(os.path.join(golden_dir, "pi0_t1.npz"), allow_pickle=False)
    cfg = P.Pi0OracleConfig()
    w = make_weights(P.pi0_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    return g, cfg, w


def test_pi0_forward_loss_and_grads_match_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    out = P.pi0_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]), t(g["image_masks"]),
                        t(g["states"]), t(g["actions"]), t(g["noise"]), t(g["time"]))
    assert rel(out["v_t"].detach().numpy(), g["v_t"]) < 2e-5
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    out["loss"].backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), g[key]) < 5e-5, key
        elif key.startswith("gradN/"):
            assert sd[key[6:]].grad is not None, key
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])


def test_pi0_inference_matches_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    t = torch.from_numpy
    with torch.no_grad():
        acts = P.pi0_inference_action(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["states"]), t(g["images"]),
                                      t(g["image_masks"]), t(g["init_noise"]), 10)
    assert rel(acts.numpy(), g["infer_actions"]) < 2e-5
