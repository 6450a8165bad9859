"""src/utils/timer.py: wall-clock tic/toc (synchronises the GPU so the interval covers the kernels)."""
import time


class Timer(object):
    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.

    @staticmethod
    def _sync():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass

    def tic(self):
        self._sync()
        self.start_time = time.time()

    def toc(self, average=True):
        self._sync()
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff
