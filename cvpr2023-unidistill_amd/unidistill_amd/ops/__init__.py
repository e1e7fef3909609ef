"""Operator layer: thin torch wrappers (autograd Functions) over the C-ABI HIP library."""
