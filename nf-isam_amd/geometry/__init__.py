"""Planar geometry needed by the factor types of the range-only SLAM configurations."""
