"""Frame-streaming runtime: keeps several frames of `PoseRegressor.forward` in flight on separate HIP
streams so that the GPU always has wide AND latency-bound kernels to run side by side.

At batch 1 a frame is a chain of ~80 dependent launches whose deep-encoder / post-network kernels
occupy a fraction of the 256 CUs.  `FrameStreamer` alternates consecutive frames between `net_streams`
native plans (each with its own workspace; the parameters are shared) and runs the post-network
stages on one more stream:

    frame i   : network on stream N[i % k]  --event-->  aggregation / voting / RT on stream P
    frame i+1 : network on stream N[(i+1) % k]            (overlaps both of the above)

`submit(x)` enqueues one frame and returns a ticket without synchronising; `collect(ticket)` waits on
that frame's own event (the only host wait of the frame), trims the per-instance tensors to the
instance count found and returns the reference's forward() dict
{'logits', 'categorical', 'aggregated'}.  Every frame does all of its work; only its completion is
deferred.  Nothing like this exists in the reference (single stream, >= 3 host syncs per instance).
"""
import copy

import torch


class FrameStreamer:

    def __init__(self, model, net_streams=2, device=None):
        p = next(model.parameters())
        self.device = device if device is not None else p.device
        if self.device.type != "cuda":
            raise RuntimeError("FrameStreamer needs the model on a GPU")
        if model.training:
            raise RuntimeError("FrameStreamer is an inference runtime: call model.eval() first")
        self.models = [model]
        for _ in range(max(1, int(net_streams)) - 1):
            m = copy.copy(model)            # shares parameters / sub-modules, owns its native plans
            m._engines = {}
            m._fused = None
            self.models.append(m)
        self.net_streams = [torch.cuda.Stream(device=self.device) for _ in self.models]
        self.post_stream = torch.cuda.Stream(device=self.device)
        self._n = 0

    def submit(self, x, categorical_override=None):
        """Enqueue one frame (x f32 [B,3,H,W] on the device).  `categorical_override` replaces the
        network's categorical output as the input of the post-network stages (benchmark fixture)."""
        k = self._n % len(self.models)
        self._n += 1
        model, stream = self.models[k], self.net_streams[k]
        stream.wait_stream(torch.cuda.current_stream(self.device))      # x was produced on the caller's stream
        with torch.no_grad():
            with torch.cuda.stream(stream):
                model._inv_k(x.device)
                logits = model.pure_model_forward(x)
                cat = model.class_compression(logits)
                ev = torch.cuda.Event()
                ev.record()
            ticket = {"logits": logits, "categorical": cat, "post": None, "model": model, "net_event": ev}
            if model.HPARAM.PERFORM_AGGREGATION:
                src = categorical_override if categorical_override is not None else cat
                with torch.cuda.stream(self.post_stream):
                    self.post_stream.wait_event(ev)
                    for t in src.values():
                        t.record_stream(self.post_stream)
                    if model.HPARAM.PERFORM_HOUGH_VOTING:
                        ticket["post"] = model.post_network_enqueue(src)
                    else:
                        ticket["agg_only"] = model.aggregate(src)
        return ticket

    def collect(self, ticket):
        """Wait for the ticket's frame (only) and return forward()'s dict."""
        model = ticket["model"]
        agg = None
        if ticket["post"] is not None:
            with torch.cuda.stream(self.post_stream):
                agg = model.post_network_finish(ticket["post"])
        elif "agg_only" in ticket:
            self.post_stream.synchronize()
            agg = ticket["agg_only"]
        else:
            ticket["net_event"].synchronize()
        return {"logits": ticket["logits"], "categorical": ticket["categorical"], "aggregated": agg}
