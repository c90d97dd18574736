def find_boundaries(*a, **k):
    raise NotImplementedError("scikit-image is not in this image")
