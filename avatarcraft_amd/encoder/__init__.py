"""Drop-in for the reference's `encoder` package (encoder/__init__.py:4-32)."""
from . import freq_encoder
from .hashencoder import HashEncoder
from .shencoder import SHEncoder


def get_encoder(encoder_type: str, encoder_configs: dict):
    """Construct the encoder and return (module_or_fn, output_dim); same types / config keys as the reference."""
    if encoder_type == "frequency":
        return freq_encoder.get_freq_embedder(encoder_configs["freq_multires"], encoder_configs["in_dim"])
    if encoder_type in ("hash", "hashgrid"):
        enc = HashEncoder(encoder_configs["in_dim"], encoder_configs["hash_num_levels"], encoder_configs["hash_level_dim"],
                          encoder_configs["hash_per_level_scale"], encoder_configs["hash_base_resolution"],
                          encoder_configs["hash_log2_hashmap_size"], encoder_configs["hash_desired_resolution"])
        return enc, enc.output_dim
    if encoder_type in ("sh", "sphere_harmonics"):
        enc = SHEncoder(encoder_configs["in_dim"])
        return enc, enc.output_dim
    raise NotImplementedError("Encoder type {} not implemented".format(encoder_type))
