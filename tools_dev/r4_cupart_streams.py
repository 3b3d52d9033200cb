"""Round-4 experiment (not adopted): the FrameStreamer's frame streams created with CU masks so that every
stream owns 1/n of the CUs of every XCD.  Dropped into FrameStreamer.__init__ in place of the torch.cuda.Stream list
(`self.net_streams = make_masked_streams(self.device, len(self.models))`), run by tools_dev/r4_cupart.sh with
FPC_CU_PARTITION=1.  Result on MI355X: a frame alone 2.14 ms on a quarter of the chip (0.97 ms on all of it), but masked
streams did not overlap: 3.1 ms per frame with four in flight, 322 img/s."""
import ctypes

import torch


def make_masked_streams(device, n):
    """n in (2, 4, 8): stream k gets mask words [k * 8 / n, (k + 1) * 8 / n) — a 32-bit word = 4 CUs in each of the 8 XCDs."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    streams, per = [], 8 // n
    with torch.cuda.device(device):
        for k in range(n):
            words = (ctypes.c_uint32 * 8)(*[0xFFFFFFFF if k * per <= w < (k + 1) * per else 0 for w in range(8)])
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
            if rc != 0 or not h.value:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
            streams.append(torch.cuda.ExternalStream(h.value, device=device))
    return streams
