"""Shim (unused by the hot path)."""
