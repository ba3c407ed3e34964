"""vgtk.utils (reference: vgtk/vgtk/utils.py:L25-68)."""
import vgtk.cuda.gathering as cuda_gather


def batch_gather(x, idx, dim=1):
    """[b,c,n] x [b,m] -> float32 [b,c,m] (utils.py:L25-27; no autograd on this route)."""
    return cuda_gather.gather_points_forward(x.contiguous(), idx.int().contiguous())


def batch_zip(x, y, idx):
    raise NotImplementedError('batch zip is not implemented (neither is it in the reference)')


class LearningRateScheduler():
    """utils.py:L33-68: step-wise constant / exponential decay."""

    def __init__(self, optimizer, init_lr, lr_type, decay_step, **kwargs):
        self.counter = 0
        self.init_lr = init_lr
        self.lr = init_lr
        self.lr_type = lr_type
        self.optimizer = optimizer
        self.decay_step = decay_step
        self.schedule_func = getattr(self, f'_{lr_type}')(**kwargs)

    def step(self):
        self.counter += 1
        if self.counter % self.decay_step == 0:
            lr = self.schedule_func(self.counter // self.decay_step)
            for group in self.optimizer.param_groups:
                group['lr'] = lr
            self.lr = lr
        return self.lr

    def _constant(self, decay_rate=None):
        return lambda x: self.init_lr

    def _exp_decay(self, decay_rate):
        self.decay_rate = decay_rate
        return lambda x: self.init_lr * decay_rate ** x
