"""varigraph_amd -- MI355X (gfx950) implementation of varigraph's per-sample genotyping hot path.

The product is the C-ABI shared library `libvgmi.so` (include/vgmi.h): hand-written HIP kernels
for read k-mer counting against the graph k-mer table, the per-node depth gather, and the
construct-side counting Bloom filter.  This package is the thin Python binding used by the tests,
`bench.py` and `__graft_entry__.py`; there is no CPU fallback -- importing `varigraph_amd.vgmi`
without the built library, or creating a context without a GPU, fails loudly.
"""
__all__ = ["build", "vgmi", "synth"]
