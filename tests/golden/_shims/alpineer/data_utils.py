"""Import stand-in (test-infra only): ark.utils.data_utils imports the module, the fixture functions do not use it."""
