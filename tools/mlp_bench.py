"""MS MLP half of a swin block at config 2's stage shapes: one-launch kernel vs the three-launch form (us per call, L3-resident)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

DEV = "cuda:0"


class L:
    def __init__(self, W, a, b, ns):
        self.N, self.K = W.shape
        self.Wp = hip.split_weight(W.to(DEV).contiguous(), ns)
        self.alpha, self.beta, self.bias = a.to(DEV), b.to(DEV), None


def run(B, D, H, W, Cc, ns=2, reps=50, only_fused=False):
    Ch = 4 * Cc
    x0 = rnd((B, D, H, W, Cc), 1, -0.5, 1.0).to(DEV)
    fc1 = L(rnd((Ch, Cc), 2, -0.3, 0.3), rnd((Ch,), 3, 0.5, 1.5), rnd((Ch,), 4, -0.2, 0.2), ns)
    fc2 = L(rnd((Cc, Ch), 5, -0.1, 0.1), rnd((Cc,), 6, 0.5, 1.5), rnd((Cc,), 7, -0.2, 0.2), ns)
    p = hip.NeuronParams("lif", 2.0, 0.1, None)
    out = {}
    for name, three in ((("one launch", False),) if only_fused else (("one launch", False), ("three launches", True))):
        x = x0.clone()
        for _ in range(5):
            hip.ms_mlp(x, fc1, fc2, p, p, three_launches=three)
        torch.cuda.synchronize()
        # the calls are replayed from a HIP graph: the host side of a call (descriptor, workspace) costs more than a 30 us kernel
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    hip.ms_mlp(x, fc1, fc2, p, p, three_launches=three)
        torch.cuda.synchronize()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) * 1e3 / reps
    flop = 2 * 2 * B * D * H * W * Cc * Ch
    print(f"B={B} D={D} {H}x{W} C={Cc}: " + "  ".join(f"{k} {v:7.1f} us ({flop / v / 1e6:6.1f} TFLOP/s)" for k, v in out.items()), flush=True)


if __name__ == "__main__":
    run(1, 10, 72, 96, 96)
    run(1, 10, 36, 48, 192)
    run(4, 20, 120, 160, 96, reps=10)
    run(4, 20, 60, 80, 192, reps=10)
