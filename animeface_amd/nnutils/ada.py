"""``ADA``: the augmentation pipe whose strength ``p`` follows the discriminator's overfitting signal
(reference nnutils/ada.py:4-36 and implementations/ADA/model.py:5-32 -- the two copies differ only in argument order).

Every ``interval`` calls of ``update_p(D(real))``:  p += sign(mean(sign(logits)) - threshold) * batch * interval / target_imgs,
clamped to [0, 1].  Under data parallelism the reference's rule needs the GLOBAL batch: the
accumulated ``signsum`` is all-reduced once per interval and ``batch_size`` is the global batch, so every rank keeps the same
``p`` and the schedule equals the single-process schedule at the same global batch."""
import torch
import torch.distributed as dist

from ..thirdparty.ada import AugmentPipe

_DEFAULT_AUGMENTS = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1,
                         brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)


class ADA(AugmentPipe):
    def __init__(self, batch_size: int, interval: int = 4, target_kimg: int = 500, threshold: float = 0.6, **augment_kwargs) -> None:
        super().__init__(**(augment_kwargs or _DEFAULT_AUGMENTS))
        self._batch_size = batch_size            # per-process batch; the world size is folded in at update time
        self._interval = interval
        self._target_img = target_kimg * 1000
        self._threshold = threshold
        self._num_iter = 0
        self.register_buffer('signsum', torch.zeros([]))
        self.p.copy_(torch.zeros([]))

    @property
    def _p_delta(self):
        return self._batch_size * self._world() * self._interval / self._target_img

    @staticmethod
    def _world():
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @torch.no_grad()
    def update_p(self, prob: torch.Tensor):
        self.signsum.add_(torch.sign(prob).sum().to(self.signsum.dtype))
        self._num_iter += 1
        if self._num_iter == self._interval:
            total = self.signsum.clone()
            if self._world() > 1:
                dist.all_reduce(total)
            signmean = total / (self._batch_size * self._world() * self._interval)
            adjust = torch.sign(signmean - self._threshold) * self._p_delta
            self.p.copy_((self.p + adjust).clamp_(0., 1.))
            self._num_iter = 0
            self.signsum.fill_(0.)
