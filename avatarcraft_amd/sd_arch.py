"""Stable-Diffusion 1.5 ARCHITECTURE stand-in for timing the real guidance when `diffusers` and the pretrained weights are absent.

The reference's guidance (models/diffusion.py:28-69,92-149) loads `runwayml/stable-diffusion-v1-5` through `diffusers` -- neither the library nor
the weights exist offline (bench.py's `real_sd` field says so).  What one SDS step costs on the MI355X does not depend on the VALUES of the weights,
only on the networks' shapes, so this module restates the two networks the step runs from their PUBLISHED configurations (the checkpoint's
`unet/config.json` and `vae/config.json`), in plain PyTorch, randomly initialised:

  * `VAEEncoder`  = AutoencoderKL.encode: conv_in 3 -> 128, four DownEncoderBlock2D (128, 256, 512, 512; 2 ResNet blocks each, stride-2 convolution
    between levels), mid block (ResNet, single-head attention, ResNet), GroupNorm / SiLU / conv_out -> 8, quant_conv; 34.2 M parameters.  Runs WITH
    grad at 512 x 512 (the only differentiable stage of mannual_backward).
  * `UNet`        = UNet2DConditionModel: block_out_channels (320, 640, 1280, 1280), 2 layers per block, CrossAttnDownBlock2D x 3 + DownBlock2D,
    mid block with cross attention, UpBlock2D + CrossAttnUpBlock2D x 3, 8 heads, cross-attention width 768, GEGLU feed-forward; 859.5 M parameters.
    Runs without grad on a batch of two 4 x 64 x 64 latents (unconditional + text) with [2, 77, 768] text embeddings.

`components()` returns the dict guidance.StableDiffusion(components=...) takes; the text encoder is a stub that returns random [n, 77, 768]
embeddings (CLIP runs once per prompt and is cached by SDSGuidance: not part of a step's cost).  NOTHING here produces meaningful images -- it is a
clock, and bench.py labels it as one ("sd_arch_standin")."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Res(nn.Module):
    def __init__(self, cin, cout, temb=None, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout) if temb else None
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _Attn(nn.Module):
    def __init__(self, dim, heads, ctx=None, bias=False):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=bias)
        self.to_k = nn.Linear(ctx or dim, dim, bias=bias)
        self.to_v = nn.Linear(ctx or dim, dim, bias=bias)
        self.to_out = nn.Linear(dim, dim)

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        b, n, c = x.shape
        sp = lambda t: t.view(b, -1, self.heads, c // self.heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(self.to_q(x)), sp(self.to_k(ctx)), sp(self.to_v(ctx)))
        return self.to_out(o.transpose(1, 2).reshape(b, n, c))


class _Transformer(nn.Module):
    """Transformer2DModel with one BasicTransformerBlock (self-attention, cross-attention, GEGLU feed-forward)"""

    def __init__(self, dim, heads, ctx):
        super().__init__()
        self.norm = nn.GroupNorm(32, dim, eps=1e-6)
        self.proj_in = nn.Conv2d(dim, dim, 1)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn1 = _Attn(dim, heads)
        self.attn2 = _Attn(dim, heads, ctx)
        self.ff_in = nn.Linear(dim, 8 * dim)
        self.ff_out = nn.Linear(4 * dim, dim)
        self.proj_out = nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        t = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(b, h * w, c)
        t = t + self.attn1(self.norm1(t))
        t = t + self.attn2(self.norm2(t), ctx)
        a, g = self.ff_in(self.norm3(t)).chunk(2, dim=-1)
        t = t + self.ff_out(a * F.gelu(g))
        return x + self.proj_out(t.reshape(b, h, w, c).permute(0, 3, 1, 2))


class _Out:
    def __init__(self, sample):
        self.sample = sample


class UNet(nn.Module):
    def __init__(self, ch=(320, 640, 1280, 1280), layers=2, heads=8, ctx=768, in_ch=4, out_ch=4):
        super().__init__()
        temb = 4 * ch[0]
        self.ch0 = ch[0]
        self.time_embedding = nn.Sequential(nn.Linear(ch[0], temb), nn.SiLU(), nn.Linear(temb, temb))
        self.conv_in = nn.Conv2d(in_ch, ch[0], 3, padding=1)
        self.down = nn.ModuleList()
        skips = [ch[0]]
        c = ch[0]
        for i, co in enumerate(ch):
            blk = nn.ModuleDict(dict(res=nn.ModuleList(), attn=nn.ModuleList()))
            for _ in range(layers):
                blk["res"].append(_Res(c, co, temb)); c = co
                if i < len(ch) - 1:
                    blk["attn"].append(_Transformer(co, heads, ctx))
                skips.append(c)
            if i < len(ch) - 1:
                blk["down"] = nn.Conv2d(c, c, 3, stride=2, padding=1)
                skips.append(c)
            self.down.append(blk)
        self.mid = nn.ModuleList([_Res(c, c, temb), _Transformer(c, heads, ctx), _Res(c, c, temb)])
        self.up = nn.ModuleList()
        for i, co in enumerate(reversed(ch)):
            blk = nn.ModuleDict(dict(res=nn.ModuleList(), attn=nn.ModuleList()))
            for _ in range(layers + 1):
                blk["res"].append(_Res(c + skips.pop(), co, temb)); c = co
                if i > 0:
                    blk["attn"].append(_Transformer(co, heads, ctx))
            if i < len(ch) - 1:
                blk["up"] = nn.Conv2d(c, c, 3, padding=1)
            self.up.append(blk)
        self.norm_out = nn.GroupNorm(32, c)
        self.conv_out = nn.Conv2d(c, out_ch, 3, padding=1)

    def forward(self, x, t, encoder_hidden_states=None):
        half = self.ch0 // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=x.device) / half)
        ang = t.to(x.device).float().reshape(-1, 1).expand(x.shape[0], 1) * freqs[None]
        temb = self.time_embedding(torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(x.dtype))
        ctx = encoder_hidden_states
        h = self.conv_in(x)
        hs = [h]
        for blk in self.down:
            for j, res in enumerate(blk["res"]):
                h = res(h, temb)
                if len(blk["attn"]):
                    h = blk["attn"][j](h, ctx)
                hs.append(h)
            if "down" in blk:
                h = blk["down"](h); hs.append(h)
        h = self.mid[2](self.mid[1](self.mid[0](h, temb), ctx), temb)
        for blk in self.up:
            for j, res in enumerate(blk["res"]):
                h = res(torch.cat([h, hs.pop()], dim=1), temb)
                if len(blk["attn"]):
                    h = blk["attn"][j](h, ctx)
            if "up" in blk:
                h = blk["up"](F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return _Out(self.conv_out(F.silu(self.norm_out(h))))


class _Posterior:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))

    def sample(self):
        return self.mean + self.std * torch.randn_like(self.mean)


class _Enc:
    def __init__(self, moments):
        self.latent_dist = _Posterior(moments)


class VAEEncoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), layers=2, latent=4):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        self.down = nn.ModuleList()
        c = ch[0]
        for i, co in enumerate(ch):
            blk = nn.ModuleDict(dict(res=nn.ModuleList()))
            for _ in range(layers):
                blk["res"].append(_Res(c, co, None, eps=1e-6)); c = co
            if i < len(ch) - 1:
                blk["down"] = nn.Conv2d(c, c, 3, stride=2, padding=0)
            self.down.append(blk)
        self.mid_res1, self.mid_res2 = _Res(c, c, None, eps=1e-6), _Res(c, c, None, eps=1e-6)
        self.mid_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.mid_attn = _Attn(c, 1, bias=True)
        self.norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)
        self.quant_conv = nn.Conv2d(2 * latent, 2 * latent, 1)

    def encode(self, x):
        h = self.conv_in(x)
        for blk in self.down:
            for res in blk["res"]:
                h = res(h)
            if "down" in blk:
                h = blk["down"](F.pad(h, (0, 1, 0, 1)))
        h = self.mid_res1(h)
        b, c, hh, ww = h.shape
        t = self.mid_norm(h).permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = h + self.mid_attn(t).reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        h = self.mid_res2(h)
        return _Enc(self.quant_conv(self.conv_out(F.silu(self.norm_out(h)))))


class _Tok:
    model_max_length = 77

    class _Ids:
        def __init__(self, n):
            self.input_ids = torch.zeros(n, 77, dtype=torch.long)

    def __call__(self, prompt, **kw):
        return self._Ids(len(prompt))


class _TextStub(nn.Module):
    """returns seeded random [n, 77, 768] embeddings: CLIP runs once per prompt (cached by SDSGuidance), not per step"""

    def __init__(self):
        super().__init__()
        self.anchor = nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, ids):
        g = torch.Generator().manual_seed(int(ids.shape[0]) + 7)
        return (torch.randn(ids.shape[0], 77, 768, generator=g).to(self.anchor.device),)


def components(seed=0, device=None):
    """device: build (and randomly initialise) the 0.9 G parameters there directly -- on the CPU the initialisation alone takes ~20 s"""
    torch.manual_seed(seed)
    if device is None:
        return dict(vae=VAEEncoder(), unet=UNet(), tokenizer=_Tok(), text_encoder=_TextStub())
    with torch.device(device):
        return dict(vae=VAEEncoder(), unet=UNet(), tokenizer=_Tok(), text_encoder=_TextStub())


def parameter_counts():
    with torch.device("meta"):
        u, v = UNet(), VAEEncoder()
    return sum(p.numel() for p in u.parameters()), sum(p.numel() for p in v.parameters())
