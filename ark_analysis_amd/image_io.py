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


# Uncompressed TIFF strips are read straight from the file into the destination array (``readinto``: one copy, no
# GIL while the kernel copies), not through Pillow's decoder: ``np.array(im)`` runs Pillow's raw codec chunk by
# chunk from Python, ~1.2 GB/s per image and mostly under the GIL -- 16 decoder threads then starve the thread
# that drives the GPU (measured: the per-FOV percentile call, 4 ms alone, took 150 ms beside them).
# rawmode -> (dtype in the file, dtype Pillow's array would have)
_RAW_MODES = {"F;32F": ("<f4", np.float32), "F;32BF": (">f4", np.float32), "I;32S": ("<i4", np.int32),
              "I;32BS": (">i4", np.int32), "I;16": ("<u2", np.uint16), "I;16B": (">u2", np.uint16),
              "I;16S": ("<i2", np.int16), "I;16BS": (">i2", np.int16), "L": ("u1", np.uint8)}
# (signed 16-bit stays int16, as the tifffile-based reader of the reference returns it -- cluster masks are saved
# that way; Pillow's own decoder would widen it to int32)


def _raw_layout(im):
    """``(file dtype, array dtype, [(row0, row1, file offset), ...])`` when the opened image is a single page of
    uncompressed full-width strips in a sample format listed above; None otherwise (Pillow then decodes it)."""
    tiles = getattr(im, "tile", None)
    if not tiles or getattr(im, "n_frames", 1) != 1:
        return None
    width = im.size[0]
    mode, strips = None, []
    for tile in tiles:
        codec, extents, offset, args = tile[0], tile[1], tile[2], tile[3]
        if codec != "raw" or not isinstance(args, tuple) or len(args) < 3 or args[1] != 0 or args[2] != 1:
            return None
        if args[0] not in _RAW_MODES or (mode is not None and args[0] != mode):
            return None
        if extents[0] != 0 or extents[2] != width:
            return None
        mode = args[0]
        strips.append((extents[1], extents[3], offset))
    file_dtype, array_dtype = _RAW_MODES[mode]
    return np.dtype(file_dtype), np.dtype(array_dtype), strips


def _read_strips(path, shape, layout, out):
    """Fills ``out`` ([H, W], C-contiguous, dtype = layout's array dtype) from the file; False if the file is
    shorter than its directory says (the caller falls back to Pillow, which raises the proper error)."""
    file_dtype, array_dtype, strips = layout
    direct = file_dtype == array_dtype and file_dtype.isnative
    with open(path, "rb", buffering=0) as f:
        for row0, row1, offset in strips:
            rows = out[row0:row1]
            f.seek(offset)
            if direct:
                if f.readinto(memoryview(rows).cast("B")) != rows.nbytes:
                    return False
            else:
                block = np.fromfile(f, dtype=file_dtype, count=rows.size)
                if block.size != rows.size:
                    return False
                rows[...] = block.reshape(rows.shape)
    return True


def read_image(path, out=None) -> np.ndarray:
    """Any single image file as an array in its own dtype (what ``np.array(PIL.Image.open(path))`` returns).
    ``out``: a C-contiguous ``[H, W]`` array to fill when shape and dtype agree (else a new array is returned)."""
    from PIL import Image
    with Image.open(path) as im:
        layout = _raw_layout(im)
        if layout is not None:
            shape = (im.size[1], im.size[0])
            dest = out if (out is not None and out.shape == shape and out.dtype == layout[1]
                           and out.flags.c_contiguous) else np.empty(shape, dtype=layout[1])
            if _read_strips(path, shape, layout, dest):
                return dest
        return np.array(im)


def read_channel(tiff_dir, fov: str, channel: str, img_sub_folder: Optional[str] = None, out=None) -> np.ndarray:
    """One channel image ``[H, W]`` in the file's dtype."""
    return read_image(_channel_file(_fov_folder(tiff_dir, fov, img_sub_folder), channel), out=out)


_DECODERS = None


def _decoder_pool():
    """Shared thread pool for reading / decoding the channel files of a FOV side by side."""
    global _DECODERS
    if _DECODERS is None:
        from concurrent.futures import ThreadPoolExecutor
        _DECODERS = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4), thread_name_prefix="pxsom-tiff")
    return _DECODERS


def read_channels(tiff_dir, fov: str, channels: Sequence[str], img_sub_folder: Optional[str] = None) -> np.ndarray:
    """``[H, W, len(channels)]`` stack in the dtype of the first channel (the loader the reference uses
    allocates the stack with the dtype of a test image).  The channel files are decoded side by side into
    one channel-planar buffer and the result is that buffer VIEWED as ``[H, W, C]``: interleaving 22 planes
    on the host costs several times the decode, so it is left to the device (``flowsom`` uploads planar
    storage as it lies and permutes there); for numpy the view behaves like any other array."""
    channels = list(channels)
    first = read_channel(tiff_dir, fov, channels[0], img_sub_folder)
    planar = np.empty((len(channels),) + first.shape, dtype=first.dtype)
    planar[0] = first

    def fill(j):
        got = read_channel(tiff_dir, fov, channels[j], img_sub_folder, out=planar[j])
        if got is not planar[j]:       # another dtype / shape, or a compressed file: converted as numpy assigns
            planar[j] = got

    list(_decoder_pool().map(fill, range(1, len(channels))))
    return planar.transpose(1, 2, 0)


def iter_stacks(tiff_dir, fovs: Sequence[str], channels: Sequence[str], img_sub_folder: Optional[str] = None,
                cache: Optional[dict] = None, fill: bool = True):
    """Yields ``(fov, stack)`` for every FOV, reading one FOV ahead on a background thread.  ``cache``
    (fov -> stack) is consulted first and, with ``fill``, filled while it stays under
    ``cache['__max_bytes__']``: the passes create_pixel_matrix makes over the same TIFFs (two percentile passes, then the tables) decode them once
    when the cohort fits the budget."""
    from concurrent.futures import ThreadPoolExecutor
    fovs = list(fovs)

    def load(fov):
        if cache is not None and fov in cache:
            return cache[fov]
        stack = read_channels(tiff_dir, fov, channels, img_sub_folder)
        if cache is not None and fill:
            used = cache.get("__bytes__", 0)
            if used + stack.nbytes <= cache.get("__max_bytes__", 0):
                cache[fov] = stack
                cache["__bytes__"] = used + stack.nbytes
        return stack

    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="pxsom-fov") as ahead:
        pending = ahead.submit(load, fovs[0]) if fovs else None
        for i, fov in enumerate(fovs):
            stack = pending.result()
            pending = ahead.submit(load, fovs[i + 1]) if i + 1 < len(fovs) else None
            yield fov, stack


def stack_cache(max_bytes: Optional[int] = None) -> dict:
    """A cache for :func:`iter_stacks`; default budget 8 GiB of host memory (``PXSOM_STACK_CACHE_GB``)."""
    if max_bytes is None:
        max_bytes = int(float(os.environ.get("PXSOM_STACK_CACHE_GB", "8")) * (1 << 30))
    return {"__max_bytes__": int(max_bytes), "__bytes__": 0}


def write_channel(path, image: np.ndarray) -> None:
    """Counterpart used by tests and the synthetic-cohort script."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(image)).save(path, format="TIFF")


_SAMPLE_FORMATS = {"u": 1, "i": 2, "f": 3}


def write_image(path, image: np.ndarray) -> None:
    """A 2-D uint8 / uint16 / int16 / int32 / float32 array as a baseline TIFF in its own dtype: little-endian,
    uncompressed, one strip, with the SampleFormat tag -- what the cluster masks are saved as (int16 stays int16,
    which Pillow cannot write).  :func:`read_image` reads it back unchanged."""
    import struct
    image = np.asarray(image)
    if image.ndim != 2 or image.dtype.name not in ("uint8", "uint16", "int16", "int32", "float32"):
        raise ValueError("write_image takes a 2-D uint8, uint16, int16, int32 or float32 array, got %s %s"
                         % (image.dtype, image.shape))
    data = np.ascontiguousarray(image, dtype=image.dtype.newbyteorder("<"))
    if data.nbytes >= 1 << 32:
        raise ValueError("image too large for a classic TIFF")
    height, width = data.shape
    short, long_ = 3, 4
    tags = [(256, long_, width), (257, long_, height), (258, short, data.dtype.itemsize * 8), (259, short, 1),
            (262, short, 1), (273, long_, 0), (277, short, 1), (278, long_, height), (279, long_, data.nbytes),
            (339, short, _SAMPLE_FORMATS[data.dtype.kind])]
    first_byte = 8 + 2 + 12 * len(tags) + 4
    directory = struct.pack("<H", len(tags))
    for tag, kind, value in tags:
        value = first_byte if tag == 273 else value
        directory += struct.pack("<HHI", tag, kind, 1) + (struct.pack("<HH", value, 0) if kind == short
                                                            else struct.pack("<I", value))
    with open(path, "wb") as f:
        f.write(b"II*\0" + struct.pack("<I", 8) + directory + struct.pack("<I", 0))
        f.write(memoryview(data).cast("B"))
