"""Stable-Diffusion SDS guidance behind the interface stylize.py uses: counterpart of models/diffusion.py:23-312.

The UNet / VAE / CLIP text encoder are NOT part of the MI355X hot path: they stay PyTorch-ROCm modules (`diffusers` / `transformers`
when installed, or any objects with the same call surface).  What lives here is the arithmetic around them, which is what the render
path exchanges data with -- an image in, d(loss)/d(image) out:

    mannual_backward (models/diffusion.py:92-149):
        rgb [1,3,h,w] -> bilinear resize to 512x512 (align_corners=False; the zero-padded copy the reference builds first is
        overwritten, quirk C.6) -> t ~ U{20..980} -> latents = vae.encode(2 rgb - 1).latent_dist.sample() * 0.18215 (WITH grad) ->
        noise ~ N(0,1), latents_noisy = sqrt(abar_t) latents + sqrt(1 - abar_t) noise -> UNet on [noisy, noisy] with [uncond, text]
        embeddings (no grad) -> classifier-free guidance with scale 100 -> grad = (1 - abar_t) (eps_hat - noise), clamped to [-1, 1] ->
        latents.backward(grad): the gradient reaches pred_rgb.grad through the VAE encoder only.

`StableDiffusion(device, version)` loads the pretrained components exactly like the reference when `diffusers` and `transformers`
are importable; `StableDiffusion(device, components=dict(vae=..., unet=..., tokenizer=..., text_encoder=...))` takes them
ready-made (tests use tiny seeded stand-ins, tests/common_sd.py).  The noise schedule (PNDMScheduler(beta_start=0.00085,
beta_end=0.012, "scaled_linear", 1000) in the reference) is restated here: only `alphas_cumprod` and `add_noise` are used by the SDS path.
`SDSGuidance` adapts it to the `guidance(rgb, text=...) -> grad` callable of avatarcraft_amd.stylize.sds_step."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ScaledLinearSchedule:
    """betas = linspace(sqrt(b0), sqrt(b1), N)^2, alphas_cumprod = cumprod(1 - betas): the only parts of diffusers' PNDMScheduler
    (schedulers/scheduling_pndm.py, "scaled_linear") that StableDiffusion.mannual_backward touches"""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        t = timesteps.to(original_samples.device)
        a = ac[t] ** 0.5
        s = (1 - ac[t]) ** 0.5
        while a.dim() < original_samples.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise


class StableDiffusion(nn.Module):
    """same constructor arguments, attributes and methods as the reference's class (models/diffusion.py:23)"""

    def __init__(self, device, version="1.5", components=None, latent_size=512, unet_autocast=None):
        """unet_autocast (not in the reference, which runs everything in fp32): a torch dtype (torch.bfloat16) under which the no-grad UNet forward
        runs (torch.autocast); the VAE encoder -- the only differentiable stage -- stays fp32.  Opt-in: the noise prediction moves by bf16 round-off."""
        super().__init__()
        self.unet_autocast = unet_autocast
        self.sd_version = version
        self.device = device
        self.num_train_timesteps = 1000
        self.min_step = int(self.num_train_timesteps * 0.02)
        self.max_step = int(self.num_train_timesteps * 0.98)
        self.use_depth = False
        self.image_size = latent_size                       # the reference hard-codes 512 (SD's training resolution)
        if version == "1.5":
            self.model_key = "runwayml/stable-diffusion-v1-5"
        elif version == "2.0":
            self.use_depth = True
            self.model_key = "stabilityai/stable-diffusion-2-depth"
        else:
            raise ValueError(f"sd_version {version!r}: the reference knows '1.5' and '2.0' (models/diffusion.py:45-49)")
        if components is None:
            try:
                from transformers import CLIPTextModel, CLIPTokenizer
                from diffusers import AutoencoderKL, UNet2DConditionModel
            except Exception as e:       # not installed in this image: the guidance is then an injected callable (SyntheticGuidance for measurement)
                raise RuntimeError("StableDiffusion needs `diffusers` and `transformers` (and the pretrained weights) -- pass components=dict(vae=, "
                                   "unet=, tokenizer=, text_encoder=) or install them; " + repr(e)) from e
            components = dict(vae=AutoencoderKL.from_pretrained(self.model_key, subfolder="vae"),
                              tokenizer=CLIPTokenizer.from_pretrained(self.model_key, subfolder="tokenizer"),
                              text_encoder=CLIPTextModel.from_pretrained(self.model_key, subfolder="text_encoder"),
                              unet=UNet2DConditionModel.from_pretrained(self.model_key, subfolder="unet"))
        self.vae = components["vae"].to(device)
        self.tokenizer = components["tokenizer"]
        self.text_encoder = components["text_encoder"].to(device)
        self.unet = components["unet"].to(device)
        self.scheduler = components.get("scheduler") or ScaledLinearSchedule(0.00085, 0.012, self.num_train_timesteps)
        self.alphas = self.scheduler.alphas_cumprod.to(device)
        for m in (self.vae, self.text_encoder, self.unet):           # frozen: only d/d(image) is wanted (the reference never steps them either)
            for p in m.parameters():
                p.requires_grad_(False)

    # ------------------------------------------------------------------ :72-89
    def get_text_embeds(self, prompt):
        if not isinstance(prompt, list):
            prompt = [prompt]
        text_input = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt")
        with torch.no_grad():
            text_embeddings = self.text_encoder(text_input.input_ids.to(self.device))[0]
        uncond_input = self.tokenizer([""] * len(prompt), padding="max_length", max_length=self.tokenizer.model_max_length, return_tensors="pt")
        with torch.no_grad():
            uncond_embeddings = self.text_encoder(uncond_input.input_ids.to(self.device))[0]
        return torch.cat([uncond_embeddings, text_embeddings])

    # ------------------------------------------------------------------ :304-312
    def encode_imgs(self, imgs):
        imgs = 2 * imgs - 1
        posterior = self.vae.encode(imgs).latent_dist
        return posterior.sample() * 0.18215

    def decode_latents(self, latents):
        latents = 1 / 0.18215 * latents
        with torch.no_grad():
            imgs = self.vae.decode(latents).sample
        return (imgs / 2 + 0.5).clamp(0, 1)

    # ------------------------------------------------------------------ :92-149
    def mannual_backward(self, text_embeddings, pred_rgb, guidance_scale=100, pred_depth=None):
        """back-propagates the SDS gradient into pred_rgb.grad (pred_rgb [1,3,h,w], requires grad); returns nothing, like the reference"""
        S = self.image_size
        pred_rgb_512 = F.interpolate(pred_rgb, (S, S), mode="bilinear", align_corners=False)
        if self.use_depth and pred_depth is not None:
            pred_depth = F.interpolate(pred_depth, size=(S // 8, S // 8), mode="bicubic", align_corners=False)
            pred_depth = 2.0 * (pred_depth - pred_depth.min()) / (pred_depth.max() - pred_depth.min()) - 1.0
            pred_depth = torch.cat([pred_depth] * 2)
        t = torch.randint(self.min_step, self.max_step + 1, [1], dtype=torch.long, device=self.device)
        self._mark("start")
        if self.channels_last and pred_rgb_512.is_cuda:
            pred_rgb_512 = pred_rgb_512.contiguous(memory_format=torch.channels_last)
        latents = self.encode_imgs(pred_rgb_512)                    # WITH grad: the only differentiable stage
        self._mark("vae_encode_fwd")
        with torch.no_grad():
            noise = torch.randn_like(latents)
            latents_noisy = self.scheduler.add_noise(latents, noise, t)
            latent_model_input = torch.cat([latents_noisy] * 2)
            if self.use_depth and pred_depth is not None:
                latent_model_input = torch.cat([latent_model_input, pred_depth], dim=1)
            if self.unet_autocast is not None and latent_model_input.is_cuda:
                with torch.autocast("cuda", dtype=self.unet_autocast):
                    noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=text_embeddings).sample.float()
            else:
                noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=text_embeddings).sample
        self._mark("unet_fwd")
        noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
        noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
        w = 1 - self.alphas[t]
        grad = (w * (noise_pred - noise)).clamp(-1, 1)
        latents.backward(gradient=grad, retain_graph=True)
        self._mark("vae_bwd")

    # measurement / tuning hooks (not in the reference).  phase_marks: a list that receives (name, torch.cuda.Event) at the phase boundaries of
    # mannual_backward (bench.py: vae_encode_fwd | unet_fwd | vae_bwd).  channels_last: see tune().
    phase_marks = None
    channels_last = False

    def _mark(self, name):
        if self.phase_marks is not None and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True); ev.record(); self.phase_marks.append((name, ev))

    def tune(self, channels_last="unet", miopen_find=True):
        """PyTorch-level settings for the two networks that keep the reference's precision (fp32 everywhere, no autocast: SURVEY 0.5): NHWC memory format for
        the convolution stacks of the VAE encoder and the UNet (MIOpen's fp32 kernels for these shapes are NHWC-native: no layout transposes around every
        convolution), MIOpen's find mode (torch.backends.cudnn.benchmark: the fastest solver per shape, picked once), SDPA attention where the modules
        allow it (the sd_arch stand-ins call F.scaled_dot_product_attention; diffusers' modules select it through their attention processor)."""
        if miopen_find:
            torch.backends.cudnn.benchmark = True
        # channels_last: "unet" (default) | "vae" | "both" / True | None.  Measured on the SD-1.5-sized stand-in (tools/guidance_tune_probe.py,
        # profiles/r06_experiments.txt section 8): NHWC helps the UNet's 64 x 64 .. 8 x 8 stages and HURTS the VAE encoder's 512 x 512 ones
        which = {True: "both", False: None}.get(channels_last, channels_last)
        self.channels_last = which in ("vae", "both")                # (the image handed to the VAE encoder follows the encoder's format)
        for name, m in (("vae", self.vae), ("unet", self.unet)):
            if isinstance(m, torch.nn.Module) and which in (name, "both"):
                m.to(memory_format=torch.channels_last)
        for m in (self.vae, self.unet):
            f = getattr(m, "set_attn_processor", None)
            if callable(f):
                try:
                    from diffusers.models.attention_processor import AttnProcessor2_0
                    f(AttnProcessor2_0())
                except Exception:                                # noqa: BLE001 -- an older diffusers: its default stays
                    pass
        return self

    def calc_grad(self, text_embeddings, pred_rgb, guidance_scale=100):
        """:150-206: the same, returning pred_rgb.grad"""
        self.mannual_backward(text_embeddings, pred_rgb, guidance_scale)
        return pred_rgb.grad.detach().clone()

    def train_step(self, text_embeddings, pred_rgb, guidance_scale=100):
        """:211-258"""
        self.mannual_backward(text_embeddings, pred_rgb, guidance_scale)
        return 0


class SDSGuidance:
    """`guidance(rgb [1,3,h,w], text=...) -> d loss / d rgb` for stylize.sds_step / stylize_epochs, over a StableDiffusion instance:
    what Trainer.train does between render_val and the patch loop (stylize.py:118-137).  Text embeddings are cached per prompt."""

    def __init__(self, sd, tgt_text="", guidance_scale=100.0):
        self.sd, self.tgt_text, self.scale = sd, tgt_text, guidance_scale
        self._emb = {}

    def __call__(self, rgb, text=None):
        text = self.tgt_text if text is None else text
        if text not in self._emb:
            self._emb[text] = self.sd.get_text_embeds([text])
        img = rgb.detach().clone().requires_grad_(True)             # "rgb_pred_global.requires_grad = True" (stylize.py:118)
        with torch.enable_grad():
            self.sd.mannual_backward(self._emb[text], img, self.scale)
        return img.grad.clone().detach()


def real_sd_probe(version="1.5"):
    """(available, reason): can StableDiffusion(device, version) load the REAL pretrained networks on this machine without a network?  Needs
    `diffusers`, `transformers` and the four sub-models of the checkpoint in the local Hugging Face cache (models/diffusion.py:45-69 downloads them
    with from_pretrained; this image has no egress).  bench.py --real-sd and tests/test_gpu_stylize.py record the outcome either way."""
    key = {"1.5": "runwayml/stable-diffusion-v1-5", "2.0": "stabilityai/stable-diffusion-2-depth"}.get(version)
    if key is None:
        return False, f"unknown sd_version {version!r}"
    missing = []
    for mod in ("diffusers", "transformers"):
        try:
            __import__(mod)
        except Exception as e:                                   # noqa: BLE001
            missing.append(f"{mod} not importable ({type(e).__name__})")
    if missing:
        return False, "; ".join(missing)
    try:
        from huggingface_hub import try_to_load_from_cache
        need = {"vae": "config.json", "unet": "config.json", "text_encoder": "config.json", "tokenizer": "vocab.json"}
        absent = [sub for sub, fn in need.items() if not isinstance(try_to_load_from_cache(key, f"{sub}/{fn}"), str)]
        if absent:
            return False, f"{key} not in the local Hugging Face cache (missing: {', '.join(absent)}); no network"
    except Exception as e:                                       # noqa: BLE001
        return False, f"huggingface_hub cache lookup failed ({type(e).__name__}: {e})"
    return True, key
