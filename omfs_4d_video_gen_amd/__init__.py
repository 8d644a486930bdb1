"""MI355X-native engine for the FLAME-rigged Gaussian head-avatar path of OMFS-4D-Video-Gen
(`02_Visual_Engine`: flame_fitter -> head_recon/train_ghost -> render_surgery).

Python call surfaces mirror the reference modules of the same name; the work behind them runs in
hand-written HIP kernels (csrc/, C ABI in include/omfs_splat.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
