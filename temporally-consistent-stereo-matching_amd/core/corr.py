"""Drop-in for the reference's core/corr.py: `CorrBlock1D` on the HIP library.

Build = fp32-MFMA row GEMM + one finalize pass that pools the 4 levels, stores them in the skewed
lookup layout, and (first frame) fuses `argmax_disp`; lookup = one coalesced kernel per call instead
of 4 grid_sample chains (csrc/tcs_corr.hip).  Same constructor and methods as core/corr.py:7-79.
"""
import torch

from tcs_mi355 import ops


class CorrBlock1D:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, thres=0.2, want_argmax=True, want_cost_volume=False):
        """`thres` is accepted and ignored like the reference (corr.py:73 hard-codes 0.3).
        want_argmax / want_cost_volume are build hints: the fused kernel emits the first-frame
        argmax and the masked [B,W2,H,W1] volume only when asked (both are recomputed on demand)."""
        if num_levels != 4:
            raise NotImplementedError("the HIP pyramid has 4 levels (corr_levels=4 in every shipped config)")
        self.num_levels, self.radius, self.thres = num_levels, radius, thres
        self._f1, self._f2 = fmap1.float().contiguous(), fmap2.float().contiguous()
        self._pyr = ops.corr_build(self._f1, self._f2, argmax=want_argmax, cost_volume=want_cost_volume)

    def __call__(self, coords):
        """coords [B,>=1,H,W] -> [B, 4*(2r+1), H, W] float (corr.py:33-52)."""
        return ops.corr_lookup(self._pyr, coords[:, :1].float().contiguous(), self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs row correlation of the normalised maps, [B,H,W1,1,W2] (corr.py:54-62)."""
        p = ops.corr_build(fmap1.float().contiguous(), fmap2.float().contiguous(), natural=True)
        B, H, W = p.B, p.H, p.W
        return p.natural[0].reshape(B, H, W, 1, W).clone()

    @property
    def corr_pyramid(self):
        """Natural-layout levels [B*H*W1,1,1,W2>>i] like the reference attribute (corr.py:20-23), i=0..3."""
        p = ops.corr_build(self._f1, self._f2, natural=True)
        return [t.reshape(p.B * p.H * p.W, 1, 1, -1) for t in p.natural]

    def get_cost_volume(self):
        if self._pyr.cost_volume is None:
            self._pyr = ops.corr_build(self._f1, self._f2, argmax=self._pyr.sparse is not None, cost_volume=True)
        return self._pyr.cost_volume

    def argmax_disp(self):
        """(sparse_disp, main_cost, mask), each [B,1,H,W] (corr.py:67-79)."""
        if self._pyr.sparse is None:
            self._pyr = ops.corr_build(self._f1, self._f2, argmax=True, cost_volume=self._pyr.cost_volume is not None)
        return self._pyr.sparse
