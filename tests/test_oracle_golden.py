"""Pin the CPU oracle against the golden vectors produced by the live reference
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cogact_oracle as O
from oracle.weights import cogact_shapes, make_weights, weights_crc

CFGS = {
    "t1": O.OracleConfig(),
    "t2": O.OracleConfig(vocab_size=640, hidden_size=512, intermediate_size=768, num_hidden_layers=3,
                         num_attention_heads=4, num_key_value_heads=2, v_hidden=192, v_inter=384,
                         v_layers=4, v_heads=3, dit_hidden=192, dit_depth=3, dit_heads=3),
}


def load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"cogact_{tag}.npz"), allow_pickle=False)
    cfg = CFGS[tag]
    w = make_weights(cogact_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"]), "RandomState weights are not reproducible here"
    return g, cfg, w


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_forward_matches_reference(golden_dir, tag):
    g, cfg, w = load(golden_dir, tag)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    t = torch.from_numpy
    drop = t(g["drop_u"]) < 0.1                                # dit.py:85-87
    out = O.cogact_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]),
                           t(g["actions"]), t(g["noise"]), t(g["timesteps"]), drop)
    feats = out["image_features"]
    # projector output in the reference is per image ([B*V, N_v, d]); ours is [B, V*N_v, d]
    assert rel(feats.reshape(-1, feats.shape[-1]).numpy(),
               g["proj_out"].reshape(-1, feats.shape[-1])) < 1e-5
    assert np.array_equal(out["attention_mask"].numpy(), g["new_attention_mask"])
    assert rel(out["inputs_embeds"].numpy(), g["inputs_embeds"]) < 1e-5
    assert rel(out["logits"].numpy(), g["logits"]) < 2e-5
    assert rel(out["x_t"].numpy(), g["x_t"]) < 1e-6
    assert rel(out["eps_hat"].numpy(), g["eps_hat"]) < 2e-5
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


@pytest.mark.parametrize("tag", ["t1"])
def test_vit_matches_reference(golden_dir, tag):
    g, cfg, w = load(golden_dir, tag)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    imgs = torch.from_numpy(g["images"])
    v = O.clip_vision_features(sd, cfg, imgs.reshape(-1, *imgs.shape[-3:]))
    assert rel(v.numpy(), g["vit_out"]) < 1e-5


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_backward_and_adamw_match_reference(golden_dir, tag):
    g, cfg, w = load(golden_dir, tag)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    drop = t(g["drop_u"]) < 0.1
    out = O.cogact_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]),
                           t(g["actions"]), t(g["noise"]), t(g["timesteps"]), drop)
    out["loss"].backward()
    no_grad = set(g["no_grad_params"].tolist())
    mine_no_grad = {k for k, p in sd.items() if p.grad is None}
    assert mine_no_grad == no_grad
    for key in g.files:
        if key.startswith("grad/"):
            n = key[5:]
            assert rel(sd[n].grad.numpy(), g[key]) < 2e-4, n
        elif key.startswith("gradS/"):
            n = key[6:]
            assert rel(sd[n].grad.reshape(-1)[::97].numpy(), g[key]) < 2e-4, n
            assert abs(float(sd[n].grad.double().norm()) - float(g["gradN/" + n])) < 2e-4 * float(g["gradN/" + n])
    names = [k for k, p in sd.items() if p.grad is not None]
    from oracle.gen_golden import no_decay_name
    params = [sd[k].detach().clone() for k in names]
    grads = [sd[k].grad.clone() for k in names]
    dec = [i for i, k in enumerate(names) if not no_decay_name(k)]
    nod = [i for i, k in enumerate(names) if no_decay_name(k)]
    # clip is global over ALL grads; the two groups only differ in weight decay
    total = float(torch.sqrt(sum((x.double() ** 2).sum() for x in grads)))
    assert abs(total - float(g["grad_norm"])) < 2e-4 * total
    coef = min(1.0, 1.0 / (total + 1e-6))
    grads = [x * coef for x in grads]
    for idx, wd in ((dec, 0.01), (nod, 0.0)):
        ps = [params[i] for i in idx]
        O.adamw_step(ps, [grads[i] for i in idx], [torch.zeros_like(p) for p in ps],
                     [torch.zeros_like(p) for p in ps], step=1, lr=1e-3, weight_decay=wd,
                     max_grad_norm=None)
    new = dict(zip(names, params))
    for key in g.files:
        if key.startswith("param1/"):
            n = key[7:]
            big = np.abs(sd[n].grad.numpy()) > 1e-5        # first Adam step is sign-like: skip ~0 grads
            assert np.abs(new[n].numpy() - g[key])[big].max() < 2e-6, n


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_inference_matches_reference(golden_dir, tag):
    g, cfg, w = load(golden_dir, tag)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    t = torch.from_numpy
    norms = {"min": g["norm_min"].tolist(), "max": g["norm_max"].tolist()}
    acts, samples, traj, cog = O.cogact_inference_action(sd, cfg, t(g["infer_ids"]), t(g["images"][:1]),
                                                         t(g["init_noise"]), norms)
    ref_traj = g["ddim_traj"]
    for i in range(len(traj)):
        assert rel(traj[i].numpy(), ref_traj[i]) < 1e-4, i
    assert rel(acts, g["infer_actions"]) < 1e-4


def test_diffusion_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "diffusion_tables.npz"))
    tr = O.training_tables(100)
    assert np.array_equal(tr.betas, g["betas"])
    assert np.array_equal(tr.alphas_cumprod, g["alphas_cumprod"])
    assert np.array_equal(tr.sqrt_alphas_cumprod, g["sqrt_alphas_cumprod"])
    assert np.array_equal(tr.sqrt_one_minus_alphas_cumprod, g["sqrt_one_minus_alphas_cumprod"])
    for n in (1, 2, 5, 10, 20, 25, 50):
        dd = O.ddim_tables(100, n)
        assert dd.timestep_map == g[f"ddim{n}/timestep_map"].tolist()
        assert np.array_equal(dd.alphas_cumprod, g[f"ddim{n}/alphas_cumprod"])
        assert np.array_equal(dd.alphas_cumprod_prev, g[f"ddim{n}/alphas_cumprod_prev"])
        assert np.array_equal(dd.sqrt_recip_alphas_cumprod, g[f"ddim{n}/sqrt_recip_alphas_cumprod"])
        assert np.array_equal(dd.sqrt_recipm1_alphas_cumprod, g[f"ddim{n}/sqrt_recipm1_alphas_cumprod"])
    # SURVEY App. A constants
    assert abs(tr.betas[0] - 6.3128159834e-04) < 1e-13
    assert abs(tr.alphas_cumprod[99] - 2.4285722794e-07) < 1e-16


def test_action_integer_rows_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "action_bins.npz"))
    V = int(g["vocab"])
    normed = O.norm_action(g["action"], g["mn"], g["mx"])
    assert np.array_equal(normed, g["normed"])
    bins = O.action2bin(g["normed_all"], V)
    assert np.array_equal(bins, g["bins"])
    assert O.bin2string(bins) == g["strings"].tolist()
    den = O.denorm(g["normed_all"], {"min": g["mn"].tolist(), "max": g["mx"].tolist()})
    assert np.array_equal(den, g["denorm"])
    dec = np.concatenate([O.discrete_action_to_continuous(s, V) for s in g["strings"].tolist()])
    assert np.array_equal(dec, g["decoded"])


def test_lm_loss_grads_and_greedy_ids_match_reference(golden_dir):
    """row A10: lm_head + HF causal-LM cross-entropy (loss, logits, gradients) and the greedy continuation
    (token ids bit-exact) against the reference's DexboticForCausalLM"""
    g = np.load(os.path.join(golden_dir, "lm_t1.npz"), allow_pickle=False)
    cfg = CFGS["t1"]
    w = {k: v for k, v in make_weights(cogact_shapes(cfg), int(g["seed"])).items() if ".action_head." not in k}
    assert weights_crc(w) == int(g["weights_crc"])
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    out = O.lm_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]), t(g["labels"]))
    assert rel(out["logits"].detach().numpy(), g["logits"]) < 2e-5
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    out["loss"].backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), g[key]) < 5e-5, key
    assert rel(sd["model.llm.embed_tokens.weight"].grad[t(g["embed_rows"])].numpy(), g["grad_embed_rows"]) < 5e-5
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])
    with torch.no_grad():
        sd0 = {k: v.detach() for k, v in sd.items()}
        ids, rows = O.greedy_decode(sd0, cfg, t(g["decode_prompt"]), t(g["images"][:1]), len(g["decode_new_ids"]))
    assert np.array_equal(ids, g["decode_new_ids"])
    assert rel(rows.numpy(), g["decode_logits"]) < 2e-5


def test_hybrid_cogact_losses_and_grads_match_reference(golden_dir):
    """HybridCogACT co-training step (text CE + has_action-weighted diffusion loss) against the reference"""
    g = np.load(os.path.join(golden_dir, "hybrid_t1.npz"), allow_pickle=False)
    cfg = CFGS["t1"]
    w = make_weights(cogact_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    out = O.hybrid_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["labels"]), t(g["images"]),
                           t(g["actions"]), t(g["has_action"]), t(g["has_text"]), t(g["noise"]), t(g["timesteps"]),
                           t(g["drop_u"]) < 0.1)
    for k in ("loss", "text_loss", "action_loss"):
        assert abs(float(out[k]) - float(g[k])) < 1e-5 * abs(float(g[k])), k
    out["loss"].backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), g[key]) < 5e-5, key
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/dexbotic"), reason="needs the reference checkout (build container only)")
def test_committed_fixtures_regenerate_bit_identically_from_the_live_reference(golden_dir, tmp_path):
    """The parity chain's first link, kept permanent: the committed generator, run NOW on the reference's own classes, writes
    cogact_t1.npz / action_bins.npz / diffusion_tables.npz whose every array equals the committed fixture bit for bit (integer keys
    array_equal, float keys same bytes).  A fixture edited by hand, a generator that drifted, or a reference change shows here."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, '/root/reference')\n"
        "import oracle.gen_golden as G\n"
        f"G.GOLD = {str(tmp_path)!r}\n"
        "torch.manual_seed(0); torch.set_num_threads(8)\n"
        "G.install_timm_shim()\n"
        "from oracle.cogact_oracle import OracleConfig\n"
        "G.gen_diffusion_tables(); G.gen_action_bins()\n"
        "G.gen_cogact('t1', OracleConfig(), seed=1234, B=3, L=12, lengths=[12, 9, 11], views=1)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in ("cogact_t1.npz", "action_bins.npz", "diffusion_tables.npz"):
        new = np.load(os.path.join(str(tmp_path), name), allow_pickle=False)
        old = np.load(os.path.join(golden_dir, name), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            a, b = old[k], new[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            if a.dtype.kind in "iub" or a.dtype.kind in "US":
                assert np.array_equal(a, b), (name, k)
            else:
                assert a.tobytes() == b.tobytes(), (name, k)
