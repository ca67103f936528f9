"""Host-side batching glue (reference utils/__init__.py:118-217): zero-padded collation and token-bucket batching."""
import sys

import torch


def collate_1d(values, pad_idx=0, max_len=None):
    size = max(v.size(0) for v in values) if max_len is None else max_len
    res = values[0].new_full((len(values), size), pad_idx)
    for i, v in enumerate(values):
        res[i, :len(v)] = v
    return res


def collate_2d(values, pad_idx=0, max_len=None):
    size = max(v.size(0) for v in values) if max_len is None else max_len
    res = values[0].new_full((len(values), size, values[0].shape[1]), pad_idx)
    for i, v in enumerate(values):
        res[i, :len(v)] = v
    return res


def batch_by_size(indices, num_tokens_fn, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
    """Greedy buckets in the given order: a batch closes when (len+1)*longest > max_tokens or len == max_sentences;
    closed batches are trimmed to a multiple of `required_batch_size_multiple` (reference utils/__init__.py:163-217)."""
    max_tokens = sys.maxsize if max_tokens is None else max_tokens
    max_sentences = sys.maxsize if max_sentences is None else max_sentences
    mult = required_batch_size_multiple
    batches, batch, lens = [], [], []
    longest = 0
    for idx in indices:
        n = num_tokens_fn(idx)
        lens.append(n)
        longest = max(longest, n)
        assert longest <= max_tokens, f"sentence at index {idx} of size {longest} exceeds max_tokens limit of {max_tokens}!"
        full = len(batch) > 0 and (len(batch) == max_sentences or (len(batch) + 1) * longest > max_tokens)
        if full:
            keep = max(mult * (len(batch) // mult), len(batch) % mult)
            batches.append(batch[:keep])
            batch, lens = batch[keep:], lens[keep:]
            longest = max(lens) if lens else 0
        batch.append(idx)
    if batch:
        batches.append(batch)
    return batches
