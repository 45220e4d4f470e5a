"""Drop-in for the reference package `flows` (src/flows/), backed by hand-written gfx950 kernels
(see nf-isam_amd/csrc and include/nfisam_hip.h).  Same module names, class names and call
signatures as the reference; tensors must live on a ROCm device — there is no CPU path."""
