EXTENSION_TYPES = {"IMAGE": ["tiff", "tif", "png", "jpg", "jpeg", "ome.tiff"], "ARCHIVE": ["tar", "gz", "zip"], "DATA": ["csv", "feather", "bin", "json"]}
