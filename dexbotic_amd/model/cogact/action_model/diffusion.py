"""Gaussian-diffusion schedule + DDIM sampler for the CogACT action expert.

Mirrors the subset of dexbotic/model/cogact/action_model/diffusion.py the CogACT path uses:
``create_diffusion`` (:1114-1150), ``space_timesteps`` (:992-1051), the float64 tables of
``GaussianDiffusion.__init__`` (:242-292) as re-derived by ``SpacedDiffusion`` (:1054-1071),
``q_sample`` (:308-326) and ``ddim_sample_loop`` with eta=0, EPSILON mean / FIXED_SMALL variance,
clip_denoised=False (:351-441,626-673,714-794).  Tables are numpy float64 on the host exactly like
the reference (``_extract_into_tensor`` :975-987 gathers then ``.float()``); the per-element arithmetic
runs in libdexbotic_amd kernels (dxa_qsample, dxa_ddim_step).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Set

import numpy as np
import torch

from .... import kernels as K


def squaredcos_cap_v2_betas(num_steps: int, max_beta: float = 0.999) -> np.ndarray:
    """beta_i = min(1 - abar((i+1)/T) / abar(i/T), max_beta), abar(t) = cos^2((t+0.008)/1.008 * pi/2)."""
    def abar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    out = np.empty(num_steps, dtype=np.float64)
    for i in range(num_steps):
        out[i] = min(1.0 - abar((i + 1) / num_steps) / abar(i / num_steps), max_beta)
    return out


def linear_betas(num_steps: int) -> np.ndarray:
    scale = 1000 / num_steps
    return np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> Set[int]:
    """Which original timesteps a respaced process keeps ("ddimN" = fixed integer stride with exactly N
    steps; "ddim1" is the reference's hard-coded {50}); otherwise per-section even spacing."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            if want == 1:
                return {50}
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


class SpacedDiffusion:
    """Schedule tables of a (possibly respaced) diffusion process + the two operations CogACT needs."""

    def __init__(self, use_timesteps: Sequence[int], betas: np.ndarray):
        base_ac = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64), axis=0)
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        self.timestep_map: List[int] = []
        new_betas, last = [], 1.0
        for i, ac in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        self.betas = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = len(self.betas)
        self.alphas_cumprod = np.cumprod(1.0 - self.betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self._dev_tables = {}

    def _table(self, name: str, device) -> torch.Tensor:
        key = (name, str(device))
        if key not in self._dev_tables:
            self._dev_tables[key] = torch.from_numpy(getattr(self, name)).to(device)      # float64 on device
        return self._dev_tables[key]

    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps; coefficient gather = indexing (plumbing), FMA = kernel."""
        a = self._table("sqrt_alphas_cumprod", x_start.device)[t].float().contiguous()
        s = self._table("sqrt_one_minus_alphas_cumprod", x_start.device)[t].float().contiguous()
        return K.qsample(x_start.float().contiguous(), noise.float().contiguous(), a, s)

    @torch.no_grad()
    def ddim_sample_loop(self, model: Callable, shape, noise: torch.Tensor, clip_denoised: bool = False,
                         model_kwargs: Optional[dict] = None, eta: float = 0.0, device=None, progress=False,
                         return_trajectory: bool = False):
        """`model(x, t, **model_kwargs)` returns the RAW network output for the batch x (for CFG: batch
        [cond; uncond], handled here through ``cfg_scale`` in model_kwargs exactly like
        DiT.forward_with_cfg, dit.py:294-311)."""
        if clip_denoised or eta != 0.0:
            raise NotImplementedError("the CogACT path samples with clip_denoised=False, eta=0 (cogact_arch.py:185-192)")
        model_kwargs = dict(model_kwargs or {})
        cfg_scale = model_kwargs.pop("cfg_scale", None)
        use_cfg = cfg_scale is not None
        x = noise.float().contiguous().clone()
        nb = x.shape[0] // 2 if use_cfg else x.shape[0]
        f32 = lambda a, i: float(np.float32(a[i]))
        traj = []
        for i in reversed(range(self.num_timesteps)):
            t = torch.full((x.shape[0],), self.timestep_map[i], device=x.device, dtype=torch.long)
            out = model(x, t, **model_kwargs)
            K.ddim_step(x, out.float().contiguous(), nb, use_cfg, float(cfg_scale or 0.0),
                        f32(self.sqrt_recip_alphas_cumprod, i), f32(self.sqrt_recipm1_alphas_cumprod, i),
                        f32(self.alphas_cumprod_prev, i))
            if return_trajectory:
                traj.append(x.clone())
        return (x, traj) if return_trajectory else x


def create_diffusion(timestep_respacing, noise_schedule: str = "linear", use_kl: bool = False,
                     sigma_small: bool = False, predict_xstart: bool = False, learn_sigma: bool = True,
                     rescale_learned_sigmas: bool = False, diffusion_steps: int = 1000) -> SpacedDiffusion:
    if predict_xstart or learn_sigma or use_kl or rescale_learned_sigmas:
        raise NotImplementedError("CogACT uses EPSILON prediction with fixed variance (action_models.py:78-83)")
    if noise_schedule == "squaredcos_cap_v2":
        betas = squaredcos_cap_v2_betas(diffusion_steps)
    elif noise_schedule == "linear":
        betas = linear_betas(diffusion_steps)
    else:
        raise NotImplementedError(f"unknown beta schedule: {noise_schedule}")
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(space_timesteps(diffusion_steps, timestep_respacing), betas)
