"""DB-CogACT policy: host-side mirror of dexbotic/model/cogact/cogact_arch.py on libdexbotic_amd kernels.

``CogActConfig`` (:13-17), ``CogActModel`` (:20-45), ``CogACTForCausalLM.forward`` (:56-147: VLM prefill ->
cognition feature of the last un-padded token -> 4x repeated diffusion loss in fp32) and
``.inference_action`` (:149-198: prefill -> CFG 1.5 -> 10-step DDIM -> de-normalised [T][A] list).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from ... import functional as Fn
from ... import graphs
from ...engine import building
from ..dexbotic_arch import (ActionOutputForCausalLM, CausalLMOutputDexbotic, DexboticConfig, DexboticForCausalLM,
                             register_with_hf,
                             DexboticVLMModel)
from .action_model.builder import build_action_model


class CogActConfig(DexboticConfig):
    model_type = "dexbotic_cogact"

    def __init__(self, action_model_type: Optional[str] = None, action_dim: Optional[int] = None,
                 chunk_size: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        self.action_model_type = action_model_type
        self.action_dim = action_dim
        self.chunk_size = chunk_size


register_with_hf(CogActConfig)


class CogActModel(DexboticVLMModel):
    def __init__(self, config: CogActConfig, store):
        super().__init__(config, store)
        self.action_head = None
        if config.action_model_type is not None:
            self.action_head = self._build_action_head_module(config)

    def _build_action_head_module(self, config: CogActConfig):
        if getattr(self, "action_head", None) is not None:
            return self.action_head
        with building(self.store):
            self.action_head = build_action_model(config)
        return self.action_head

    @property
    def action_head_module(self) -> nn.Module:
        return self.action_head

    @property
    def action_head_prefix(self) -> str:
        return "action_head"


class CogACTForCausalLM(DexboticForCausalLM, ActionOutputForCausalLM):
    config_class = CogActConfig
    # every sample of a batch is processed independently of the others and of the call order: the micro-batches of a gradient
    # accumulation group may run as one batch (trainer.NativeTrainer coalesce_micro_batches)
    coalescible_micro_batches = True
    gradient_side_stream = True      # trainer.NativeTrainer: the fp32 head's dW products and the bias column sums beside the dX chain

    def _real_init(self, config: CogActConfig):
        self.model = CogActModel(config, self.store)
        self.store.new_bucket()
        self.store.register([("lm_head.weight", (config.vocab_size, config.hidden_size))])

    def unused_parameter_names(self) -> List[str]:
        """parameters that never get a gradient in CogACT training (lm_head, the CLIP layer after
        hidden_states[-2], post_layernorm, history_embedder): skipped by the DP reducer."""
        names = ["lm_head.weight"] + self.model.mm_vision_tower.unused_parameter_names()
        names += [n for n in self.store.slots if ".history_embedder." in n]
        return names

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, cache_position=None, actions=None, states=None, repeated_diffusion_steps: int = 4,
                **kwargs) -> CausalLMOutputDexbotic:
        """kwargs ``noise`` [R*B,T,A], ``timesteps`` [R*B], ``drop_ids`` [R*B] inject the random draws of the
        action loss (parity tests); by default they are drawn like the reference does."""
        (_, position_ids, attention_mask, past_key_values, inputs_embeds, labels, cache_position
         ) = self.model._prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                               labels, cache_position, images)
        last_hidden_state = self.model.run_llm(inputs_embeds, attention_mask)            # [B,S,d], post final norm
        B, S, d = last_hidden_state.shape
        loss = None
        if attention_mask is not None and actions is not None:
            plan = self.model._last_plan
            idx = plan.dev(last_hidden_state.device)["last_flat"]
            cognition = Fn.GatherRowsFn.apply(last_hidden_state.reshape(B * S, d), idx)      # [B,d] fp32
            A, T = self.config.action_dim, self.config.chunk_size
            acts = actions.reshape(actions.size(0), -1, A).float()[:, :T, :]
            R = repeated_diffusion_steps
            acts_rep = acts.repeat(R, 1, 1)
            cog_rep = cognition.repeat(R, 1).unsqueeze(1)                                   # [R*B,1,d]
            loss = self.model.action_head_module.loss(acts_rep, cog_rep, noise=kwargs.get("noise"),
                                                      timestep=kwargs.get("timesteps"), drop_ids=kwargs.get("drop_ids"))
        return CausalLMOutputDexbotic(loss=loss, logits=last_hidden_state, past_key_values=None,
                                      hidden_states=(last_hidden_state,), attentions=None)

    __call__ = nn.Module.__call__

    # ------------------------------------------------------------------------------------------ inference
    def _sample_actions(self, images: torch.Tensor, plan_t: torch.Tensor, B: int, S: int, noise: torch.Tensor,
                        cfg_scale: float, num_ddim_steps: int, return_trajectory: bool = False):
        """Device-only part of inference_action (cogact_arch.py:151-204): vision tower -> projector -> splice ->
        decoder -> cognition token -> DDIM loop over the DiT head.  No host synchronisation and no host->device
        copies, so the whole thing can be captured in one HIP graph."""
        image_features = self.model._extract_vision_features(images)
        embeds = Fn.SpliceFn.apply(image_features, self.store.params[self.model.llm.embed_name], self.store,
                                   self.model.llm.embed_name, plan_t).view(B, S, -1)
        hidden = self.model.llm(embeds, None, None)
        cognition = hidden[:, -1, :].float().unsqueeze(1)                                    # [B,1,d]
        head = self.model.action_head
        if cfg_scale > 1.0:
            noise = torch.cat([noise, noise], 0)
            unc = self.store.w32("model.action_head.net.z_embedder.uncondition")          # [1,d]
            z = torch.cat([cognition, unc.unsqueeze(0).expand(B, 1, -1)], 0)
            model_kwargs = dict(z=z, cfg_scale=cfg_scale)
            sample_fn = head.net.forward_with_cfg
        else:
            model_kwargs = dict(z=cognition)
            sample_fn = head.net.forward
        N_rows = noise.shape[0]
        if not return_trajectory and hasattr(head.net, "fused_sampler_ok") and \
                head.net.fused_sampler_ok(N_rows, self.config.chunk_size + 1):
            # all DDIM steps in ONE persistent launch (embedders, blocks, final layer, guidance and update inside)
            return head.net.ddim_sample_fused(noise[:B], model_kwargs["z"], head.ddim_diffusion,
                                              cfg_scale if cfg_scale > 1.0 else None), None
        res = head.ddim_diffusion.ddim_sample_loop(sample_fn, noise.shape, noise, clip_denoised=False,
                                                   model_kwargs=model_kwargs, eta=0.0, device=cognition.device,
                                                   return_trajectory=return_trajectory)
        samples, traj = (res if isinstance(res, tuple) else (res, None))
        return samples[:B], traj

    @torch.no_grad()
    def inference_action(self, input_ids, image_tensor, inference_args={}, **kwargs):
        """Reference contract (cogact_arch.py:151-204).  The device work of a given request shape is captured once into a
        HIP graph (second call with that shape; the first runs eagerly) and replayed afterwards — bit-identical to the eager
        launches (tests/test_parity_gpu.py).  The request is GPU-bound (436 launches, 19.3 ms of kernels at the BASELINE shape),
        the replay removes the launch gaps: p50 19.76 -> 19.46 ms (profiles/r04_infer_eager_vs_graph.txt; rounds 1-3 measured it
        level or slower and kept it off).  ``inference_args["use_graph"]=False`` or env DXA_INFER_GRAPH=0: eager launches.  At most
        ``MAX_INFER_GRAPHS`` request shapes keep a graph (least recently used dropped): instruction lengths vary in serving."""
        cfg_scale = inference_args.get("cfg_scale", 1.5)
        num_ddim_steps = inference_args.get("num_ddim_steps", 10)
        action_norms = inference_args.get("action_norms")
        return_traj = bool(kwargs.get("return_trajectory"))
        head = self.model.action_head
        if head.ddim_diffusion is None or head.ddim_diffusion.num_timesteps != num_ddim_steps:
            head.create_ddim(ddim_step=num_ddim_steps)
        dev = self.store.device
        images = image_tensor.to(device=dev, dtype=self.store.compute_dtype)
        n_img_tokens = self.model.num_image_tokens(images)
        plan = self.model._plans.get(input_ids, None, None, n_img_tokens,
                                     getattr(self.config, "tokenizer_model_max_length", None),
                                     getattr(self.config, "tokenizer_padding_side", "right"))
        B, S = plan.plan.shape
        noise = kwargs.get("noise")
        if noise is None:
            noise = torch.randn(B, self.config.chunk_size, self.config.action_dim, device=dev, dtype=torch.float32)
        noise = noise.to(device=dev, dtype=torch.float32)
        use_graph = inference_args.get("use_graph", os.environ.get("DXA_INFER_GRAPH", "1") != "0")
        from ... import kernels as K
        head.net.used_fused = False
        graph_stream = None
        if dev.type == "cuda" and use_graph and not return_traj:
            if hasattr(head.net, "refresh_packed"):
                head.net.refresh_packed()          # derived weight copies a captured graph reads (bf16 sampler): up to date before the replay
            samples = self._graph_sample(images, plan.plan.reshape(-1), B, S, noise, float(cfg_scale), int(num_ddim_steps))
            traj = None
            ent = self._infer_graphs[(tuple(images.shape), B, S, float(cfg_scale), int(num_ddim_steps))]
            graph_stream = ent["stream"]
            # a replay runs no Python of DiT.forward: whether the captured work contains the persistent kernel was recorded
            # when the entry was built, so its abort word is checked after EVERY replay of such an entry
            head.net.used_fused = bool(ent.get("fused"))
        else:
            plan_t = plan.dev(dev)["plan"]
            samples, traj = self._sample_actions(images, plan_t, B, S, noise, float(cfg_scale), int(num_ddim_steps),
                                                 return_traj)
        host = samples[0].cpu().numpy()
        if dev.type == "cuda" and head.net.used_fused and K.dit_blocks_timed_out(graph_stream):
            # the persistent DiT kernel needs all its workgroups resident at once; something else held CUs for seconds
            # (another process on this GPU): this request is redone on the block-by-block kernels, which this process
            # keeps using from now on
            head.net.allow_fused = False
            self.__dict__.pop("_infer_graphs", None)      # every captured graph may hold the persistent kernel: re-capture
            samples, traj = self._sample_actions(images, plan.dev(dev)["plan"], B, S, noise, float(cfg_scale),
                                                 int(num_ddim_steps), return_traj)
            host = samples[0].cpu().numpy()
        actions = self._denorm(host, action_norms).tolist()
        if traj is not None:
            return actions, samples, traj
        return actions

    MAX_INFER_GRAPHS = 8

    def _graph_sample(self, images, plan_np, B, S, noise, cfg_scale, num_ddim_steps):
        key = (tuple(images.shape), B, S, cfg_scale, num_ddim_steps)
        cache = self.__dict__.setdefault("_infer_graphs", {})
        ent = cache.pop(key, None)
        if ent is not None:
            cache[key] = ent                                   # most recently used last
        if ent is None:
            while len(cache) >= self.MAX_INFER_GRAPHS:
                cache.pop(next(iter(cache)))                   # least recently used: its graph and static buffers are freed
            # first request of this shape: eager on a private stream (this also lets every lazily created
            # resource — split-K scratch of that stream, device tables, kernel attributes — come into being)
            # ONE private stream for every request shape of this model: the library keeps per-stream scratch (64 MiB of split-K
            # partials, the DiT sync block) for the life of the process, so a stream per shape would grow without bound
            shared = self.__dict__.get("_infer_stream")
            if shared is None:
                shared = self.__dict__["_infer_stream"] = torch.cuda.Stream(device=images.device)
            ent = {"stream": shared, "graph": None,
                   "images": images.clone(), "noise": noise.clone(),
                   "plan_host": torch.from_numpy(plan_np.copy()).pin_memory(),
                   "plan": torch.from_numpy(plan_np).to(images.device)}
            cache[key] = ent
            ent["stream"].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ent["stream"]):
                out, _ = self._sample_actions(ent["images"], ent["plan"], B, S, ent["noise"], cfg_scale, num_ddim_steps)
            ent["fused"] = bool(self.model.action_head.net.used_fused)
            torch.cuda.current_stream().wait_stream(ent["stream"])
            return out
        ent["plan_host"].copy_(torch.from_numpy(plan_np))
        ent["stream"].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ent["stream"]):
            ent["images"].copy_(images, non_blocking=True)
            ent["noise"].copy_(noise, non_blocking=True)
            ent["plan"].copy_(ent["plan_host"], non_blocking=True)
            if ent["graph"] is None:
                g = torch.cuda.CUDAGraph()
                with graphs.capture(g, ent["stream"]):
                    ent["out"], _ = self._sample_actions(ent["images"], ent["plan"], B, S, ent["noise"], cfg_scale,
                                                         num_ddim_steps)
                ent["graph"] = g
                ent["fused"] = bool(self.model.action_head.net.used_fused)
            ent["graph"].replay()
        torch.cuda.current_stream().wait_stream(ent["stream"])
        return ent["out"]


from ..dexbotic_arch import register_model_with_hf  # noqa: E402

register_model_with_hf(CogACTForCausalLM)
