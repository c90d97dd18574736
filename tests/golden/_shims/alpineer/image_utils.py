"""Shim: the one function the mask-saving helpers of the reference call."""
import numpy as np


def save_image(fname, data, compression_level=6):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(data)).save(fname, format="TIFF")
