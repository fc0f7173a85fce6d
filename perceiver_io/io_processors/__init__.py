"""Shim package, see perceiver_io/__init__.py."""
