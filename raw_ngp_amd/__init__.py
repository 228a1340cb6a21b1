"""raw_ngp_amd -- MI355X (gfx950) implementation of raw_ngp's data-parallel hot path.

Layout (mirrors the reference's operator packages so a raw_ngp-style renderer can import them
unchanged):  gridencoder/  shencoder/  freqencoder/  raymarching/  encoding.py  activation.py
nerf/{network,renderer}.py, all driving hand-written HIP kernels in csrc/ through the C ABI
declared in include/ngp_hip.h.  The HIP library is mandatory: nothing here falls back to a CPU
or PyTorch implementation of the kernels.
"""
__version__ = "0.1.0"
