"""On-disk dataset of the reference, byte-compatible (utils/indexed_datasets.py:7-54):
`<prefix>.data` = concatenated pickles, `<prefix>.idx` = np.save({'offsets': [...]})."""
import pickle

import numpy as np


class IndexedDataset:
    def __init__(self, path, num_cache=1):
        self.path = path
        self.offsets = np.load(f"{path}.idx", allow_pickle=True).item()["offsets"]
        self._fh = open(f"{path}.data", "rb")
        self._cache = {}
        self._cache_cap = max(0, num_cache)

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        if i < 0 or i >= len(self):
            raise IndexError("index out of range")
        if i in self._cache:
            return self._cache[i]
        self._fh.seek(self.offsets[i])
        item = pickle.loads(self._fh.read(self.offsets[i + 1] - self.offsets[i]))
        if self._cache_cap:
            if len(self._cache) >= self._cache_cap:
                self._cache.pop(next(iter(self._cache)))
            self._cache[i] = item
        return item

    def __del__(self):
        fh = getattr(self, "_fh", None)
        if fh:
            fh.close()


class IndexedDatasetBuilder:
    def __init__(self, path):
        self.path = path
        self._fh = open(f"{path}.data", "wb")
        self.offsets = [0]

    def add_item(self, item):
        self.offsets.append(self.offsets[-1] + self._fh.write(pickle.dumps(item)))

    def finalize(self):
        self._fh.close()
        with open(f"{self.path}.idx", "wb") as f:
            np.save(f, {"offsets": self.offsets})
