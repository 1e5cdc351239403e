"""Serving surface: the reference's ``moe_infinity/entrypoints``."""
