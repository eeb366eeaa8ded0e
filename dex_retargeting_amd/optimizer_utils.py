"""Exponential low-pass filter, same semantics as the reference's LPFilter
(/root/reference/src/dex_retargeting/optimizer_utils.py:1-17): first sample passes through, then
``y += alpha * (x - y)``.  Works on (n,) vectors and, unchanged, on (B, n) batches."""


class LPFilter:
    def __init__(self, alpha):
        self.alpha = alpha
        self.y = None
        self.is_init = False

    def next(self, x):
        if not self.is_init:
            self.y = x
            self.is_init = True
            return self.y.copy()
        self.y = self.y + self.alpha * (x - self.y)
        return self.y.copy()

    def reset(self):
        self.y = None
        self.is_init = False
