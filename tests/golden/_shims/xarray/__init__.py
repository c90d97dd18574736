"""Import stand-in (test-infra only): names ark.utils.data_utils mentions in annotations."""


class DataArray:  # never instantiated by the functions the fixtures call
    pass
