"""Host control flow of the fold/epoch driver: early stopping (src/mdl/earlystopping.py:26-39) and the
ReduceLROnPlateau(factor=0.1, patience=2) schedule the reference builds at src/mdl/fnn.py:105 (stepped at
fnn.py:163).  Only scalars are involved, so this stays on the host."""
from __future__ import annotations

import math


class EarlyStopping:
    def __init__(self, patience=5, verbose=False, delta=0.0, trace_func=print):
        self.patience, self.verbose, self.delta, self.trace_func = patience, verbose, delta, trace_func
        self.counter, self.best_score, self.early_stop, self.val_loss_min = 0, None, False, math.inf

    def __call__(self, val_loss, model=None):
        score = -val_loss
        if self.best_score is None:
            self.best_score = score
            self._improved(val_loss)
        elif score < self.best_score + self.delta:
            self.counter += 1
            self.trace_func(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_score = score
            self._improved(val_loss)
            self.counter = 0
        return self

    def _improved(self, val_loss):
        if self.verbose:
            self.trace_func(f"Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f})")
        self.val_loss_min = val_loss


class PlateauLR:
    """torch.optim.lr_scheduler.ReduceLROnPlateau with mode='min', threshold=1e-4 (relative), cooldown 0, min_lr 0,
    eps 1e-8 — the defaults the reference relies on."""

    def __init__(self, lr, factor=0.1, patience=2, threshold=1e-4, eps=1e-8):
        self.lr, self.factor, self.patience, self.threshold, self.eps = float(lr), factor, patience, threshold, eps
        self.best, self.num_bad = math.inf, 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = self.lr * self.factor
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad = 0
        return self.lr
