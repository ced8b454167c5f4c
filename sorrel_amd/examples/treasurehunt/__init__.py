"""Treasurehunt: Sorrel's tutorial environment (``sorrel/examples/treasurehunt``) on the
batched engine -- also the canonical synthetic workload of the benchmark."""
