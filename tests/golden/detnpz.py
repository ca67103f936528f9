"""Deterministic .npz writer for the golden generators: np.savez_compressed stamps every zip member with the current time, so two
runs over identical arrays differ in bytes.  Fixed member order + fixed timestamp + fixed permissions -> same arrays, same file."""
import io
import zipfile

import numpy as np


def savez_det(path, **arrays):
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        for name in arrays:
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(arrays[name]), allow_pickle=False)
            zi = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            zi.compress_type = zipfile.ZIP_DEFLATED
            zi.external_attr = 0o644 << 16
            z.writestr(zi, buf.getvalue())
