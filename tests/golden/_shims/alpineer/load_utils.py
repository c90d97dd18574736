"""Stand-in for alpineer.load_utils (test infrastructure for tests/golden/make_golden.py only).

Only ``load_imgs_from_tree`` and only the slice of its xarray result the reference's pixel path touches
(``.values``, ``[0].values``, ``.loc[fov, :, :, channels].values``, ``.channels.values``).  Documented
assumption about the real package (not installable here): the stack takes the dtype of the image files
(float32 for float32 TIFFs) and lists a FOV's channels in natural order."""
import os

import numpy as np
from PIL import Image

from ark_analysis_amd.host_utils import natsorted


class _Coord:
    def __init__(self, values):
        self.values = np.array(values)


class _Loc:
    def __init__(self, owner):
        self._owner = owner

    def __getitem__(self, key):
        fov, _, _, chans = key
        o = self._owner
        f = list(o.fovs.values).index(fov)
        if isinstance(chans, slice):
            idx = list(range(len(o.channels.values)))
        else:
            idx = [list(o.channels.values).index(c) for c in chans]
        return _Stack(o.values[f][:, :, idx], None, [o.channels.values[i] for i in idx])


class _Stack:
    def __init__(self, values, fovs, channels):
        self.values = values
        self.fovs = _Coord(fovs) if fovs is not None else None
        self.channels = _Coord(channels)
        self.loc = _Loc(self)

    def __getitem__(self, i):
        return _Stack(self.values[i], None, self.channels.values)


def load_imgs_from_tree(data_dir, img_sub_folder=None, fovs=None, channels=None, max_image_size=None):
    sub = img_sub_folder or ""
    if channels is None:
        names = [os.path.splitext(f)[0] for f in os.listdir(os.path.join(data_dir, fovs[0], sub))
                 if f.endswith((".tiff", ".tif"))]
        channels = natsorted(names)
    planes = []
    for fov in fovs:
        per_chan = []
        for ch in channels:
            with Image.open(os.path.join(data_dir, fov, sub, ch + ".tiff")) as im:
                per_chan.append(np.array(im))
        planes.append(np.stack(per_chan, axis=-1))
    arr = np.stack(planes, axis=0)
    return _Stack(arr.astype(planes[0].dtype, copy=False), list(fovs), list(channels))
