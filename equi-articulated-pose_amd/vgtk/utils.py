"""vgtk.utils -- what `vgtk/__init__.py` of the reference exports from vgtk/vgtk/utils.py:L25-68: the gather helper of the
operator layer, the (unimplemented there as here) batch_zip, and the step-wise learning-rate schedule the trainers build
(`vgtk.LearningRateScheduler`, SPConvNets/trainer_unsup_arti_align.py:L165) -- kept so that code importing the reference
package finds the same names."""
import vgtk.cuda.gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """[b,c,n] x [b,m] -> float32 [b,c,m] through the native gather (no autograd on this route, as in the reference)."""
    return cuda_gather.gather_points_forward(x.contiguous(), idx.int().contiguous())


def batch_zip(x, y, idx):
    raise NotImplementedError('batch zip cuda not implemented')


class LearningRateScheduler:
    """Learning rate held piecewise constant over windows of `decay_step` calls of step(): window n (n = 1, 2, ...) uses
    init_lr ('constant') or init_lr * decay_rate ** n ('exp_decay'); step() returns the rate in force and writes it into
    every parameter group of the optimizer when a window starts."""

    def __init__(self, optimizer, init_lr, lr_type, decay_step, **kwargs):
        self.optimizer, self.init_lr, self.lr_type, self.decay_step = optimizer, init_lr, lr_type, decay_step
        self.counter, self.lr = 0, init_lr
        self.schedule_func = getattr(self, '_' + lr_type)(**kwargs)

    def step(self):
        self.counter += 1
        window, into = divmod(self.counter, self.decay_step)
        if into == 0:
            self.lr = self.schedule_func(window)
            for group in self.optimizer.param_groups:
                group['lr'] = self.lr
        return self.lr

    def _constant(self, decay_rate=None):
        return lambda window: self.init_lr

    def _exp_decay(self, decay_rate):
        self.decay_rate = decay_rate
        return lambda window: self.init_lr * decay_rate ** window
