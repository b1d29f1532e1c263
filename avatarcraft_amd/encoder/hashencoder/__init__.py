from .hashgrid import HashEncoder, hash_encode
