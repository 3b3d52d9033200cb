"""fastposecnn_amd — MI355X-native hot path of FastPoseCNN inference."""
# FrameStreamer keeps 4 frames on 4 HIP streams; the HIP runtime's default of 4 hardware queues would make
# two of them (plus the null stream) share a queue and serialise.  Must be set before the first HIP call.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
