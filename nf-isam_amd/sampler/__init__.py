"""Clique training-batch generation (the producer of the flow hot path's input)."""
