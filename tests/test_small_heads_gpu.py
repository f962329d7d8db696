"""The remaining factory branches of the plugin surface on the GPU: `LinearModel` regression head (action_models.py:14-45),
`mlpNx_gelu` with N > 2 and `linearNx` projectors (mm_projector/builder.py:51-79), and `AutoModel.from_pretrained` resolving a
checkpoint directory to the native class (the registration pattern of dexbotic/model/dm0/__init__.py:12-16).  Each against the
plain torch modules the reference builds, forward and gradients, fp32."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _store_with(build):
    from dexbotic_amd.engine import ParamStore, attach_parameters, building
    st = ParamStore(DEV, torch.float32)
    with building(st):
        mod = build()
    st.finalize(train=True)
    root = nn.Module()
    attach_parameters(root, st)
    return st, mod


def _load(st, ref: nn.Module, prefix: str):
    for k, v in ref.state_dict().items():
        st.w32(prefix + k).copy_(v.to(DEV))


def test_linear_model_head_matches_torch():
    from dexbotic_amd.model.cogact.action_model.builder import build_action_model
    cfg = types.SimpleNamespace(action_model_type="Linear", hidden_size=192, action_dim=7, chunk_size=1)
    st, head = _store_with(lambda: build_action_model(cfg))
    torch.manual_seed(0)
    ref = nn.Sequential(nn.Linear(192, 768), nn.ReLU(), nn.Linear(768, 768), nn.ReLU(), nn.Linear(768, 7)).double()
    _load(st, ref.float(), "model.action_head.linear.")
    ref = ref.double()
    z = torch.randn(6, 1, 192, device=DEV, requires_grad=True)
    x = torch.rand(6, 1, 7, device=DEV) * 2 - 1
    st.begin_step()
    loss = head.loss(x, z)
    loss.backward()
    zr = z.detach().cpu().double().requires_grad_(True)
    want = torch.abs(ref(zr) - x.cpu().double()).mean()
    want.backward()
    assert abs(loss.item() - want.item()) < 1e-5 * abs(want.item())
    assert torch.allclose(z.grad.cpu().double(), zr.grad, rtol=1e-4, atol=1e-7)
    for i in (0, 2, 4):
        g = st.g(f"model.action_head.linear.{i}.weight").cpu().double()
        assert torch.allclose(g, ref[i].weight.grad, rtol=1e-4, atol=1e-7), i
    assert torch.allclose(head(z.detach()).cpu().double(), ref(zr.detach()), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("ptype", ["mlp3x_gelu", "linear2x", "linear"])
def test_projector_types_match_torch(ptype):
    from dexbotic_amd.model.modules.mm_projector.builder import build_vision_projector
    mm, hid = 64, 128
    cfg = types.SimpleNamespace(mm_projector_type=ptype, mm_hidden_size=mm, hidden_size=hid, projector_bias=True)
    st, proj = _store_with(lambda: build_vision_projector(cfg))
    torch.manual_seed(1)
    if ptype == "mlp3x_gelu":
        ref = nn.Sequential(nn.Linear(mm, hid), nn.GELU(), nn.Linear(hid, hid), nn.GELU(), nn.Linear(hid, hid))
        in_dim = mm
    elif ptype == "linear2x":
        ref, in_dim = nn.Linear(2 * mm, hid, bias=True), 2 * mm
    else:
        ref, in_dim = nn.Linear(mm, hid), mm
    _load(st, ref, "model.mm_projector.")
    ref = ref.double()
    x = torch.randn(2, 70, in_dim, device=DEV, requires_grad=True)
    st.begin_step()
    y = proj(x)
    probe = torch.randn_like(y)
    (y * probe).sum().backward()
    xr = x.detach().cpu().double().requires_grad_(True)
    yr = ref(xr)
    (yr * probe.cpu().double()).sum().backward()
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(x.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-5)
    names = [n for n in st.slots if n.endswith("weight")]
    for n in names:
        mod = ref if isinstance(ref, nn.Linear) else ref[int(n.split(".")[-2])]
        assert torch.allclose(st.g(n).cpu().double(), mod.weight.grad, rtol=1e-4, atol=1e-5), n


def test_automodel_resolves_a_checkpoint_directory_to_the_native_class(golden_dir, tmp_path):
    from transformers import AutoModel
    from tests.helpers import build_product, load_golden
    from dexbotic_amd.model.cogact.cogact_arch import CogACTForCausalLM
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "float32", DEV, train=False)
    m.save_pretrained(str(tmp_path))
    m2 = AutoModel.from_pretrained(str(tmp_path))
    assert type(m2) is CogACTForCausalLM
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert sd.keys() == sd2.keys() and all(torch.equal(sd[k].cpu(), sd2[k].cpu()) for k in sd)
