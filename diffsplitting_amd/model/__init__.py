"""Mirror of the reference model package (model/__init__.py:5-9)."""
