"""Interval timer for the train / detect scripts (the role src/utils/timer.py plays in the reference's
loops: `T.tic()` ... `T.toc(average=False)` around ten iterations, pascal_train_darknet.py:93-105).

GPU work is asynchronous, so a host wall clock around `sess.run`-like calls would time the enqueue, not the
kernels.  This timer brackets the interval with HIP events on the current stream: `toc` waits for the closing
event only (no device-wide synchronise) and returns seconds of device time between the two marks.  Without a
GPU it degrades to `time.perf_counter`."""
import time


class Timer(object):
    def __init__(self):
        self.intervals = []          # seconds, one entry per tic/toc pair
        self._open = None

    @staticmethod
    def _device():
        try:
            import torch
            return torch if torch.cuda.is_available() else None
        except ImportError:
            return None

    def tic(self):
        torch = self._device()
        if torch is None:
            self._open = ("host", time.perf_counter())
        else:
            mark = torch.cuda.Event(enable_timing=True)
            mark.record()
            self._open = ("hip", mark)

    def toc(self, average=True):
        if self._open is None:
            raise RuntimeError("toc() without tic()")
        kind, mark = self._open
        self._open = None
        if kind == "host":
            seconds = time.perf_counter() - mark
        else:
            torch = self._device()
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            end.synchronize()
            seconds = mark.elapsed_time(end) * 1e-3
        self.intervals.append(seconds)
        return self.average_time if average else seconds

    @property
    def diff(self):
        return self.intervals[-1] if self.intervals else 0.0

    @property
    def calls(self):
        return len(self.intervals)

    @property
    def total_time(self):
        return float(sum(self.intervals))

    @property
    def average_time(self):
        return self.total_time / len(self.intervals) if self.intervals else 0.0
