class AnnCollection:
    pass
