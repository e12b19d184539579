"""Primitive names, in the order architecture alphas index them (reference search/genotypes.py:5-11;
`alphas[...].argmax()` is used as an index into this list, train/model_seg.py:141)."""
PRIMITIVES = [
    'skip',
    'conv',
    'conv_downup',
    'conv_2x',
    'conv_2x_downup',
]
