"""Stage timer with the reference's interface (F/tools/timer.py:8-63).

Same decorator protocol (`enabled`, `runtimes`, `average` in ms, `fps`, `clear`) and the same
six stage names are used by lib/pose_regressor.py, so `model.report_runtime()` prints the same
table.  Events are HIP events (torch.cuda.Event on ROCm), created lazily so that importing the
package on a CPU-only box does not touch the GPU runtime.
"""
import functools

import numpy as np
import torch


class TimerDecorator(object):
    """Decorator for timing functions"""

    def __init__(self, name):
        self.name = name
        self.runtimes = []
        self.enabled = False
        self.start = None
        self.end = None

    def __call__(self, function):

        @functools.wraps(function)
        def wrapper(*args, **kwargs):
            if not self.enabled or not torch.cuda.is_available():
                return function(*args, **kwargs)
            if self.start is None:
                self.start = torch.cuda.Event(enable_timing=True)
                self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
            result = function(*args, **kwargs)
            self.end.record()
            # the reference synchronises the whole device here (timer.py:37); the end event suffices
            self.end.synchronize()
            self.runtimes.append(self.start.elapsed_time(self.end))
            return result

        return wrapper

    @property
    def average(self):
        self._average = np.mean(np.array(self.runtimes))
        return self._average

    @property
    def fps(self):
        self._fps = 1000 / self.average
        return self._fps

    def clear(self):
        self.runtimes.clear()
