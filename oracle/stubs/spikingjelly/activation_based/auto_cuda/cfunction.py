"""import-only placeholder (reference: Spiking_submodules.py:5)."""
