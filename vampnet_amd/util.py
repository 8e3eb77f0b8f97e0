"""The reference's `vampnet.util` helpers that callers of the hot path use (vampnet/util.py:6-46): layout glue, no arithmetic.
(`parallelize` — a tqdm map used by dataset scripts — is not part of the path.)"""
import torch


def scalar_to_batch_tensor(x, batch_size):
    """util.py:6-7"""
    return torch.tensor(x).repeat(batch_size)


def codebook_flatten(tokens: torch.Tensor):
    """(batch, codebook, time) -> (batch, time * codebook), time-major like the classifier's output (util.py:35-39)."""
    b, c, t = tokens.shape
    return tokens.permute(0, 2, 1).reshape(b, t * c)


def codebook_unflatten(flat_tokens: torch.Tensor, n_c: int = None):
    """(batch, time * codebook) -> (batch, codebook, time) (util.py:41-46)."""
    b, n = flat_tokens.shape
    return flat_tokens.reshape(b, n // n_c, n_c).permute(0, 2, 1)
