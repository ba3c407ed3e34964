"""vgtk.utils -- the gather helper of the operator layer (reference: vgtk/vgtk/utils.py:L25-27).  The reference's
LearningRateScheduler lives in the same file; it is training runtime and not part of this package."""
import vgtk.cuda.gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """[b,c,n] x [b,m] -> float32 [b,c,m] through the native gather (no autograd on this route, as in the reference)."""
    return cuda_gather.gather_points_forward(x.contiguous(), idx.int().contiguous())
