"""Drop-in subset of the reference package `utils` needed by the flow hot path."""
