"""Architecture-parameter optimiser (drop-in for the reference's search/architect.py, first-order path).

`Architect(model, args)` builds one Adam(lr=args.arch_learning_rate, betas=(0.5, 0.999)) per architecture set and
`step(input_train, target_train, input_valid, target_valid)` performs: zero grads -> `model._loss(valid)` + latency
penalty from `model.forward_latency((3,1024,2048))` mixed 1:497:2 over alpha/beta/ratio (architect.py:55-76) ->
backward -> Adam step.  The DARTS second-order code of the reference (:31-40,78-127) is unreachable there
(`config.unrolled = False`, and it references an undefined `Network`), so `unrolled=True` raises here.

Under data parallelism pass `grad_sync` (a callable run between backward and the optimizer step) to all-reduce the
~1.3 k architecture gradients.
"""
import torch
from torch import nn


class Architect(object):

    def __init__(self, model, args, distill=False, grad_sync=None):
        self.network_momentum = args.momentum
        self.network_weight_decay = args.weight_decay
        self.model = model
        self._args = args
        self._distill = distill
        self._kl = nn.KLDivLoss()
        self.optimizers = [torch.optim.Adam(arch_param, lr=args.arch_learning_rate, betas=(0.5, 0.999))
                           for arch_param in self.model._arch_parameters]
        self.latency_weight = args.latency_weight
        assert len(self.latency_weight) == len(self.optimizers)
        self.latency = 0
        self.latency_supernet = 0
        self.grad_sync = grad_sync
        self.loss_fn = None           # train_step.SupernetStep: `_loss` with adjacent passes evaluated together (same value and gradients)
        self.latency_input = (3, 1024, 2048)

    def step(self, input_train, target_train, input_valid, target_valid, eta=None, network_optimizer=None, unrolled=False):
        if unrolled:
            raise NotImplementedError("second-order (unrolled) DARTS update is dead code in the reference and is not provided")
        for optimizer in self.optimizers:
            optimizer.zero_grad()
        loss, loss_latency = self._backward_step(input_valid, target_valid)
        self.last_loss = loss.detach()                # `_loss` alone, before the latency penalty
        loss.backward()
        if not (isinstance(loss_latency, (int, float)) and loss_latency == 0):
            loss_latency.backward()
        if self.grad_sync is not None:
            self.grad_sync([p for group in self.model._arch_parameters for p in group])
        for optimizer in self.optimizers:
            optimizer.step()
        if hasattr(self.model, "note_arch_update"):
            self.model.note_arch_update()
        return loss + loss_latency

    def _backward_step(self, input_valid, target_valid):
        loss = self.loss_fn(input_valid, target_valid) if self.loss_fn is not None else self.model._loss(input_valid, target_valid)
        return loss, self._latency_loss()

    def _latency_loss(self):
        """The latency penalty of `_backward_step` (architect.py:60-76): 0, or a device scalar."""
        loss_latency = 0
        self.latency_supernet = 0
        self.model.prun_mode = None
        single_width = len(self.model._width_mult_list) == 1
        mix = ((1. / 500, (True, False, False)), (499. / 500, (False, True, False))) if single_width else \
              ((1. / 500, (True, False, False)), (497. / 500, (False, True, False)), (2. / 500, (False, False, True)))
        for idx in range(len(self.optimizers)):
            self.model.arch_idx = idx
            if self.latency_weight[idx] > 0:
                latency = 0
                for weight, (a, b, r) in mix:
                    latency = latency + weight * self.model.forward_latency(self.latency_input, alpha=a, beta=b, ratio=r)
                self.latency_supernet = latency
                loss_latency = loss_latency + latency * self.latency_weight[idx]
        return loss_latency
