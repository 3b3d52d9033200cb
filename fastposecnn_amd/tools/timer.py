"""Stage timer with the reference's interface (F/tools/timer.py:8-63): a decorator object per stage with
`enabled`, `runtimes` (ms), `average`, `fps` and `clear()`.  lib/pose_regressor.py uses the same six stage names,
so `model.report_runtime()` prints the same table.

Differences by design: the two HIP events (torch.cuda.Event on ROCm) are created on first use, so importing the
package on a CPU-only box never touches the GPU runtime; a timed call waits on its own end event instead of
synchronising the whole device (the reference's `torch.cuda.synchronize()`, timer.py:37), so other streams keep
running; without a GPU the wrapper is a pass-through.
"""
import functools
import statistics

import torch


class TimerDecorator:

    def __init__(self, name):
        self.name = name
        self.enabled = False
        self.runtimes = []
        self._events = None

    def _timed(self, fn, args, kwargs):
        if self._events is None:
            self._events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        begin, finish = self._events
        begin.record()
        out = fn(*args, **kwargs)
        finish.record()
        finish.synchronize()
        self.runtimes.append(begin.elapsed_time(finish))
        return out

    def __call__(self, fn):
        @functools.wraps(fn)
        def timed_or_plain(*args, **kwargs):
            if self.enabled and torch.cuda.is_available():
                return self._timed(fn, args, kwargs)
            return fn(*args, **kwargs)
        return timed_or_plain

    @property
    def average(self):
        """Mean runtime in ms (NaN before the first timed call, like numpy's mean of nothing)."""
        return statistics.fmean(self.runtimes) if self.runtimes else float("nan")

    @property
    def fps(self):
        return 1000.0 / self.average

    def clear(self):
        self.runtimes.clear()

    def __deepcopy__(self, memo):
        # a stage timer is shared by every model that uses the stage (and HIP events cannot be copied)
        return self
