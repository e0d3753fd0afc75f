"""Host-side helpers that keep the reference's config / masking conventions."""
import torch


class Munch(dict):
    """Attribute-style dict (the reference uses the `munch` package, models.py:24,672)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def recursive_munch(d):
    """utils.py:63-69 of the reference: nested dict/list -> Munch."""
    if isinstance(d, dict):
        return Munch((k, recursive_munch(v)) for k, v in d.items())
    if isinstance(d, list):
        return [recursive_munch(v) for v in d]
    return d


def length_to_mask(lengths):
    """utils.py:42-46: True where position >= length."""
    mask = torch.arange(int(lengths.max()), device=lengths.device).unsqueeze(0).expand(lengths.shape[0], -1)
    return torch.gt(mask.type_as(lengths) + 1, lengths.unsqueeze(1))
