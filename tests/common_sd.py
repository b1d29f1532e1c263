"""Tiny seeded stand-ins with the call surface of the `diffusers` / `transformers` objects models/diffusion.py uses (neither library
nor any weights are available offline).  tests/golden/make_golden.py installs them as stub `diffusers` / `transformers` modules, imports the
REFERENCE's models/diffusion.py over them and records its SDS gradient; the tests hand the same objects to avatarcraft_amd.guidance."""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Dist:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def sample(self):
        return self.mean + self.std * torch.randn_like(self.mean)      # consumes the global RNG stream, like DiagonalGaussianDistribution.sample


class TinyVAE(nn.Module):
    """encode: 3 -> 4 channels at 1/8 resolution (two strided convolutions), a posterior with a small fixed std"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(101)
        self.c1 = nn.Conv2d(3, 8, 4, stride=4); self.c2 = nn.Conv2d(8, 4, 2, stride=2)
        self.d1 = nn.ConvTranspose2d(4, 3, 8, stride=8)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.2

    def encode(self, x):
        h = self.c2(torch.tanh(self.c1(x)))
        return types.SimpleNamespace(latent_dist=_Dist(h, 0.05))

    def decode(self, z):
        return types.SimpleNamespace(sample=self.d1(z))

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()


class TinyUNet(nn.Module):
    """eps(x, t, text): two 3x3 convolutions modulated by the timestep and the mean text embedding"""
    in_channels = 4

    def __init__(self, dim=16):
        super().__init__()
        g = torch.Generator().manual_seed(202)
        self.c1 = nn.Conv2d(4, 12, 3, padding=1); self.c2 = nn.Conv2d(12, 4, 3, padding=1)
        self.te = nn.Linear(dim, 12); self.tt = nn.Linear(1, 12)
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.3

    def forward(self, x, t, encoder_hidden_states=None):
        e = self.te(encoder_hidden_states.mean(1)) + self.tt(t.reshape(-1, 1).float().expand(x.shape[0], 1) / 1000.0)
        h = torch.tanh(self.c1(x) + e[:, :, None, None])
        return types.SimpleNamespace(sample=self.c2(h))

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()


class TinyTokenizer:
    model_max_length = 77

    def __call__(self, prompts, padding=None, max_length=77, truncation=False, return_tensors="pt"):
        ids = torch.zeros(len(prompts), max_length, dtype=torch.long)
        for i, p in enumerate(prompts):
            for j, ch in enumerate(p[:max_length - 1]):
                ids[i, j + 1] = 1 + (ord(ch) % 63)
        return types.SimpleNamespace(input_ids=ids)

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()


class TinyTextEncoder(nn.Module):
    def __init__(self, dim=16):
        super().__init__()
        g = torch.Generator().manual_seed(303)
        self.emb = nn.Embedding(64, dim)
        self.emb.weight.data = torch.randn(64, dim, generator=g)

    def forward(self, ids):
        return (self.emb(ids),)

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()


def components():
    return dict(vae=TinyVAE(), unet=TinyUNet(), tokenizer=TinyTokenizer(), text_encoder=TinyTextEncoder())


class StubPNDMScheduler:
    """what models/diffusion.py needs of diffusers.PNDMScheduler: the constructor keywords, `alphas_cumprod`, `add_noise` -- the published
    "scaled_linear" schedule (betas = linspace(sqrt(beta_start), sqrt(beta_end), N)**2) written out for the stub module"""

    def __init__(self, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", num_train_timesteps=1000):
        assert beta_schedule == "scaled_linear"
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)

    def add_noise(self, original_samples, noise, timesteps):
        sqrt_alpha_prod = self.alphas_cumprod[timesteps] ** 0.5
        sqrt_one_minus_alpha_prod = (1 - self.alphas_cumprod[timesteps]) ** 0.5
        while len(sqrt_alpha_prod.shape) < len(original_samples.shape):
            sqrt_alpha_prod = sqrt_alpha_prod.unsqueeze(-1); sqrt_one_minus_alpha_prod = sqrt_one_minus_alpha_prod.unsqueeze(-1)
        return sqrt_alpha_prod * original_samples + sqrt_one_minus_alpha_prod * noise
