"""import-only placeholder (reference: Spiking_modules.py:14)."""
