"""flows.utils — rational-quadratic spline entry points of the reference module surface
(reference: src/flows/utils.py:17-164).

`unconstrained_RQS` on ROCm tensors runs the gfx950 spline kernel (same device code as the
fused flow kernels).  `searchsorted` is a one-line host/torch helper in the reference and stays
one here.  There is no CPU spline path."""
import torch

import nfisam_hip as _nh

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def searchsorted(bin_locations, inputs, eps=1e-6):
    """Index of the bin each input falls in; bumps the last knot by eps IN PLACE like the
    reference (utils.py:17-22)."""
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


def _check_defaults(min_bin_width, min_bin_height, min_derivative):
    if (min_bin_width, min_bin_height, min_derivative) != (DEFAULT_MIN_BIN_WIDTH, DEFAULT_MIN_BIN_HEIGHT,
                                                            DEFAULT_MIN_DERIVATIVE):
        raise NotImplementedError("the gfx950 spline kernel is compiled for the reference's default minimum "
                                  "bin width/height/derivative (1e-3), which the NF-iSAM path never overrides")


def unconstrained_RQS(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False,
                      tail_bound=1., is_circular=False, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                      min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
    """Elementwise RQ spline with linear tails (utils.py:25-66): inputs [M], widths/heights [M,K],
    derivatives [M,K-1] -> (outputs [M], logabsdet [M])."""
    _check_defaults(min_bin_width, min_bin_height, min_derivative)
    if is_circular:
        raise NotImplementedError("is_circular=True is never used on the NF-iSAM path (SURVEY.md §8 a5)")
    return _nh.rqs(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, bool(inverse),
                   float(tail_bound))


def RQS(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False, left=0.,
        right=1., bottom=0., top=1., min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
        min_derivative=DEFAULT_MIN_DERIVATIVE):
    """Bounded RQ spline on [left,right] x [bottom,top] with K+1 derivative logits (utils.py:69-164).
    Raises ValueError for inputs outside the domain, like the reference."""
    _check_defaults(min_bin_width, min_bin_height, min_derivative)
    if inputs.numel() and (torch.min(inputs) < left or torch.max(inputs) > right):
        raise ValueError("Input outside domain")
    return _nh.rqs_box(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, bool(inverse),
                       float(left), float(right), float(bottom), float(top))
