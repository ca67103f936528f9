"""IndexedDataset (reference utils/indexed_datasets.py:7-54, byte-compatible) under forked DataLoader workers: a descriptor that
was opened -- and read from -- BEFORE the fork is shared by the parent and all workers; item reads must not depend on its offset."""
import numpy as np
import torch

from neuralsvb_amd.utils.indexed_datasets import IndexedDataset, IndexedDatasetBuilder


class _Items(torch.utils.data.Dataset):
    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        return self.ds[i]


def test_reads_are_positional_across_forked_workers(tmp_path):
    rng = np.random.RandomState(0)
    items = [{"id": i, "x": rng.randn(int(rng.randint(10, 4000))).astype(np.float32)} for i in range(96)]
    b = IndexedDatasetBuilder(str(tmp_path / "train"))
    for it in items:
        b.add_item(it)
    b.finalize()
    ds = IndexedDataset(str(tmp_path / "train"), num_cache=0)
    assert ds[5]["id"] == 5                                  # the file is open (and its offset moved) before any fork
    loader = torch.utils.data.DataLoader(_Items(ds), batch_size=None, shuffle=False, num_workers=3, collate_fn=None)
    for epoch in range(3):
        for i, got in enumerate(loader):
            mine = ds[(7 * i + epoch) % len(ds)]             # the parent keeps reading through the same descriptor meanwhile
            assert mine["id"] == (7 * i + epoch) % len(ds)
            assert int(got["id"]) == i
            assert np.array_equal(np.asarray(got["x"]), items[i]["x"])
