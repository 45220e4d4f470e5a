"""Factor types that produce the clique training batches of the range-only SLAM configurations."""
