"""Drop-in for the hot-path part of the reference package `slam` (src/slam/): the density-model
adapter `slam.NFiSAM`, the variable types and the solver hook interface."""
