"""Cleanup (``sorrel/examples/cleanup``) on the batched engine."""
