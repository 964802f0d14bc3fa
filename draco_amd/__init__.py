"""draco_amd -- MI355X (gfx950) native m-mode map-making path.

Drop-in for the hot path of radiocosmology/draco (``MModeTransform``, the
``DirtyMapMaker`` / ``MaximumLikelihoodMapMaker`` / ``WienerMapMaker`` family and the
forward ``SimulateSidereal``): Python task classes with the reference's names,
``setup``/``process`` signatures and config attributes, calling hand-written HIP
kernels through the C ABI declared in ``include/draco_amd.h`` (``libdraco_amd.so``).

There is no CPU fallback: importing the compute layer without the built library, or
running it without a GPU, raises.
"""

__version__ = "0.1.0"
