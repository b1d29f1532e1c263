"""NeRF frequency (sin/cos) positional encoding -- counterpart of the reference's
encoder/freq_encoder.py:10-54 (pure PyTorch there as well; config-1 plumbing, not a kernel)."""
import torch


class FreqEncoder:
    def __init__(self, input_dims, num_freqs, max_freq_log2, include_input=True, log_sampling=True,
                 periodic_fns=(torch.sin, torch.cos)):
        self.input_dims, self.include_input, self.periodic_fns = input_dims, include_input, tuple(periodic_fns)
        if log_sampling:
            self.freq_bands = 2. ** torch.linspace(0., max_freq_log2, num_freqs)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** max_freq_log2, num_freqs)
        self.out_dim = input_dims * (int(include_input) + len(self.periodic_fns) * num_freqs)

    def embed(self, inputs):
        parts = [inputs] if self.include_input else []
        for freq in self.freq_bands:
            for fn in self.periodic_fns:
                parts.append(fn(inputs * freq))
        return torch.cat(parts, -1)


def get_freq_embedder(multires, input_dims=3):
    enc = FreqEncoder(input_dims, multires, multires - 1)
    return (lambda x, eo=enc: eo.embed(x)), enc.out_dim
