"""back2future_amd: MI355X-native computeFlow hot path of JJanai/back2future.

    from back2future_amd import back2future
    computeFlow = back2future.init('Ours-Soft-ft-KITTI')
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
"""
__all__ = ["back2future", "weights"]
