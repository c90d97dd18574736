"""Single-channel TIFF images on disk, laid out the way the Pixie notebooks expect them:
``<tiff_dir>/<fov>/[<img_sub_folder>/]<channel>.tiff``.  The reference reads them through
``alpineer.load_utils.load_imgs_from_tree`` (scikit-image underneath); neither package is in this image, so
the few things the pixel path needs are done with Pillow: list a FOV's channels, read one channel, read a
stack of channels as ``[H, W, C]`` in the files' own dtype (float32 for MIBI/MPLEX exports)."""
import os
from typing import List, Optional, Sequence

import numpy as np

from .host_utils import natsorted

_EXTENSIONS = (".tiff", ".tif")


def _fov_folder(tiff_dir, fov: str, img_sub_folder: Optional[str]) -> str:
    return os.path.join(tiff_dir, fov, img_sub_folder or "")


def channel_names(tiff_dir, fov: str, img_sub_folder: Optional[str] = None) -> List[str]:
    """Channel names (file names without extension) present for ``fov``, naturally sorted."""
    folder = _fov_folder(tiff_dir, fov, img_sub_folder)
    return natsorted(os.path.splitext(f)[0] for f in os.listdir(folder)
                     if f.lower().endswith(_EXTENSIONS) and not f.startswith("."))


def _channel_file(folder: str, channel: str) -> str:
    for ext in _EXTENSIONS:
        path = os.path.join(folder, channel + ext)
        if os.path.exists(path):
            return path
    raise FileNotFoundError("The file/path, %s.tiff, could not be found in %s" % (channel, folder))


def read_channel(tiff_dir, fov: str, channel: str, img_sub_folder: Optional[str] = None) -> np.ndarray:
    """One channel image ``[H, W]`` in the file's dtype."""
    from PIL import Image
    with Image.open(_channel_file(_fov_folder(tiff_dir, fov, img_sub_folder), channel)) as im:
        return np.array(im)


def read_channels(tiff_dir, fov: str, channels: Sequence[str], img_sub_folder: Optional[str] = None) -> np.ndarray:
    """``[H, W, len(channels)]`` stack in the dtype of the first channel (the loader the reference uses
    allocates the stack with the dtype of a test image)."""
    planes = [read_channel(tiff_dir, fov, ch, img_sub_folder) for ch in channels]
    return np.stack(planes, axis=-1).astype(planes[0].dtype, copy=False)


def write_channel(path, image: np.ndarray) -> None:
    """Counterpart used by tests and the synthetic-cohort script."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(image)).save(path, format="TIFF")
