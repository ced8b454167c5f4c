"""Tag (``sorrel/examples/tag``): the first agent <-> agent interaction rule on the batched engine."""
