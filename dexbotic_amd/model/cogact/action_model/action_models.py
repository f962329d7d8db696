"""Diffusion action model (mirror of dexbotic/model/cogact/action_model/action_models.py:63-135)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .... import functional as Fn
from ....engine import ParamStore
from .diffusion import create_diffusion
from .dit import DiT


def DiT_S(**kw):
    return DiT(depth=6, hidden_size=384, num_heads=4, **kw)


def DiT_B(**kw):
    return DiT(depth=12, hidden_size=768, num_heads=12, **kw)


def DiT_L(**kw):
    return DiT(depth=24, hidden_size=1024, num_heads=16, **kw)


class LinearModel(nn.Module):
    """the regression head of action_models.py:14-45: Linear(token, 768) -> ReLU -> Linear(768, 768) -> ReLU -> Linear(768, 7),
    L1 loss against the scaled action.  Parameter names ``linear.{0,2,4}.{weight,bias}`` like the reference's nn.Sequential.
    The three products run on the fp32 masters (the head sits inside the reference's autocast(float32) region)."""

    def __init__(self, store: ParamStore, prefix: str, token_size: int, model_type: str, in_channels: int,
                 future_action_window_size: int, past_action_window_size: int, action_scale: float = 1.0):
        super().__init__()
        from ....engine import Fp32View
        self.store, self.p, self.action_scale = Fp32View(store), prefix, action_scale
        store.new_bucket()
        for i, (o, k) in zip((0, 2, 4), ((768, token_size), (768, 768), (7, 768))):
            store.register([(f"{prefix}linear.{i}.weight", (o, k)), (f"{prefix}linear.{i}.bias", (o,))])

    def _mlp(self, z: torch.Tensor) -> torch.Tensor:
        from .... import _lib as L
        st, p = self.store, self.p
        anchor = st.params[p + "linear.4.weight"]
        h = z.float().contiguous()
        for i, act in ((0, L.ACT_RELU), (2, L.ACT_RELU), (4, L.ACT_NONE)):
            h = Fn.LinearFn.apply(h, anchor, st, f"{p}linear.{i}.weight", f"{p}linear.{i}.bias", act, None)
        return h

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._mlp(x) / self.action_scale

    def loss(self, x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        out = self._mlp(z)
        assert out.shape == x.shape
        # (a [N, 1, 7] tensor: the L1 mean is torch glue, like the reference line :43)
        return torch.abs(out - x.float() * self.action_scale).mean()


# model-size registry (action_models.py:60); tests may register tiny sizes the same way
DiT_models = {"DiT-S": DiT_S, "DiT-B": DiT_B, "DiT-L": DiT_L}


class ActionModel(nn.Module):
    def __init__(self, store: ParamStore, prefix: str, token_size: int, model_type: str, in_channels: int,
                 future_action_window_size: int, past_action_window_size: int, diffusion_steps: int = 100,
                 noise_schedule: str = "squaredcos_cap_v2", use_per_attn: bool = False,
                 per_token_size: Optional[int] = None):
        super().__init__()
        self.in_channels = in_channels
        self.noise_schedule = noise_schedule
        self.diffusion_steps = diffusion_steps
        self.diffusion = create_diffusion(timestep_respacing="", noise_schedule=noise_schedule,
                                          diffusion_steps=diffusion_steps, sigma_small=True, learn_sigma=False)
        self.ddim_diffusion = None
        self.past_action_window_size = past_action_window_size
        self.future_action_window_size = future_action_window_size
        self.net = DiT_models[model_type](store=store, prefix=prefix + "net.", token_size=token_size,
                                          in_channels=in_channels, class_dropout_prob=0.1, learn_sigma=False,
                                          future_action_window_size=future_action_window_size,
                                          past_action_window_size=past_action_window_size,
                                          use_per_attn=use_per_attn, per_token_size=per_token_size)
        self.model_type = model_type

    def loss(self, x: torch.Tensor, z: torch.Tensor, reduction: str = "mean", *, noise: Optional[torch.Tensor] = None,
             timestep: Optional[torch.Tensor] = None, drop_ids: Optional[torch.Tensor] = None,
             sample_weight: Optional[torch.Tensor] = None, per_token: Optional[torch.Tensor] = None, per_repeat: int = 1):
        """x (N,T,A) ground-truth chunk, z (N,1,token) condition.  The three random draws of the reference
        (action_models.py:106-109 noise/timestep, dit.py:85-87 CFG drop) can be injected for parity tests;
        otherwise they come from torch's device RNG exactly where the reference draws them."""
        # reduction="mean": CogACT (cogact_arch.py:134).  HybridCogACT asks for "none" and then takes the
        # has_action-weighted mean of the per-sample means (hybrid_cogact_arch.py:165-173): pass that weight as
        # `sample_weight` [N] and the weighted scalar comes back (the [N,T,A] tensor itself is never needed).
        assert reduction == "mean" or sample_weight is not None, "reduction='none' needs sample_weight"
        x = x.float()
        if noise is None:
            noise = torch.randn_like(x)
        if timestep is None:
            timestep = torch.randint(0, self.diffusion.num_timesteps, (x.size(0),), device=x.device)
        if drop_ids is None and self.training and self.net.class_dropout_prob > 0:
            drop_ids = torch.rand(x.shape[0], device=x.device) < self.net.class_dropout_prob
        x_t = self.diffusion.q_sample(x, timestep, noise)
        # per_token: MemVLA; per_repeat R: x / z are R stacked repeats of the per_token.shape[0] distinct samples (DiT.forward)
        noise_pred = self.net(x_t, timestep, z, drop_ids=drop_ids, per_token=per_token, per_repeat=per_repeat)
        assert noise_pred.shape == noise.shape == x.shape
        if sample_weight is not None:
            return Fn.MseLossRowsFn.apply(noise_pred, noise.float(), sample_weight)
        return Fn.MseLossFn.apply(noise_pred, noise.float())

    def create_ddim(self, ddim_step: int = 10):
        self.ddim_diffusion = create_diffusion(timestep_respacing="ddim" + str(ddim_step),
                                               noise_schedule=self.noise_schedule,
                                               diffusion_steps=self.diffusion_steps, sigma_small=True,
                                               learn_sigma=False)
        return self.ddim_diffusion
