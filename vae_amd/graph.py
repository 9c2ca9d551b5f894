"""Recording a whole step of the hot path into ONE HIP graph.

Small and medium configurations (BASELINE C1: B = 4096, L = 32; a VRNN step; Concat at any size) are launch-bound from
Python: 5-9 us kernels behind 30-70 us of interpreter and autograd time per call.  Every entry point of the library is
asynchronous on the caller's stream and allocates nothing, and in-kernel noise reads its Philox offset from a device
word under capture (vae_amd/rng.py), so the whole module-level step -- forward, loss, autograd backward -- records
into one graph whose replay costs one launch: C1 0.147 -> 0.048 ms, the C3 Concat step 0.19 -> 0.12 ms, the C5
sequence 7 -> 1.9 ms (profiles/r03/r03_configs.txt).

    step = vae_amd.graph.capture(lambda: train_step(static_x))     # train_step returns tensors (or None)
    for batch in loader:
        static_x.copy_(batch)
        outs = step()            # replays; `outs` are the SAME tensors every time (copy what must survive)

Rules that torch's CUDA graphs impose and this helper enforces or documents:
  * inputs are static tensors (copy new data into them), outputs are static tensors (read them after the replay);
  * leaves whose gradient was first accumulated on another stream must not be reused inside the capture (torch's
    AccumulateGrad would wait on that stream, which HIP rejects under capture): create the parameters / leaves, then
    call capture() before any eager backward on them -- the warm-up steps here run on the capture stream;
  * host-side schedule state (GumbelSoftmax.tau / .iteration, a KL annealer) advances at capture time only:
    re-capture when it has to change;  the Philox SEED is recorded too (vae_amd.rng.manual_seed restarts the counter).
"""
import torch


class CapturedStep(object):
    """Callable wrapper of a captured graph; `outputs` is what the recorded function returned."""

    def __init__(self, graph, outputs, stream):
        self.graph, self.outputs, self.stream = graph, outputs, stream

    def __call__(self):
        self.graph.replay()
        return self.outputs

    def release(self):
        """Drop the graph (and its private memory pool)."""
        self.graph = None
        self.outputs = None


def capture(fn, warmup=3, stream=None):
    """Run `fn` `warmup` times on a side stream (allocator, per-stream workspace, lazily created buffers), then record
    one more call into a HIP graph.  Returns a CapturedStep."""
    if not torch.cuda.is_available():
        raise RuntimeError("vae_amd.graph.capture needs a ROCm device")
    side = torch.cuda.Stream() if stream is None else stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, int(warmup))):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outputs = fn()
    return CapturedStep(graph, outputs, side)
