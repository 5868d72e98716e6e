"""Spike-forced oracle replay: the parity statement for a chaotic spiking network (TEST INFRASTRUCTURE).

A free-running GPU forward cannot be compared with the reference's flow element by element: one spike that lands on the
other side of the threshold because a pre-activation differs in its last bit decorrelates everything downstream (DESIGN.md
section 2).  What CAN be proven, layer by layer, on the very forward whose flow is reported:

  1. the GPU forward runs with the engine's tape on: the u8 spikes of every neuron layer are kept;
  2. the oracle forward is replayed with `oracle.NEURON_HOOK`: at every neuron call the oracle has computed that layer's
     pre-activation FROM THE GPU'S OWN UPSTREAM SPIKES (they were substituted at the previous calls), so it differs from
     the GPU's pre-activation by floating-point rounding only; the GPU's spikes for this layer must then be
     `delta_consistent`: equal to the reference decision wherever the reference margin |h - v_th| exceeds delta, with
     the reset dynamics following the decision actually taken.  `unexplained` = decisions that are neither: must be 0;
  3. the flows the replay ends with (reference arithmetic on the GPU's last spikes) must equal the GPU's flows to
     floating-point tolerance.

Together: the GPU forward is an execution of the reference network in which every arithmetic result is within fp32
rounding of the reference's and every spike decision is the reference's except where the reference itself is within
delta of the threshold."""
import torch

from oracle import sdformer_oracle as O

# delta = DELTA_ULPS * 2^-23 * max(rms(pre-activation), v_th).  Measured on the MI355X over every configuration in
# tests/test_replay_gpu.py: the departures from the reference's spikes need at most 2.1 such ulps (report["needed_ulps"]);
# 16 leaves 8x headroom for other boxes' library / clock-dependent accumulation orders and still marks < 1e-5 of all decisions
# as ambiguous
DELTA_ULPS = 16.0


def to_reference_layout(spikes, layout, shape):
    t = spikes
    if layout == "BDHWC->TBCHW":
        t = t.permute(1, 0, 4, 2, 3)
    elif layout == "BDHWC->TBHWC":
        t = t.permute(1, 0, 2, 3, 4)
    elif layout != "flat":
        raise ValueError(layout)
    return t.reshape(shape)


def run(engine, x_gpu, chunk_cpu, sd, forward_oracle, delta_ulps=DELTA_ULPS):
    """-> (gpu_flows, replay_flows, per-layer report list).  `forward_oracle(chunk)` runs the oracle's forward."""
    return run_part(engine, lambda: engine.forward(x_gpu), lambda: forward_oracle(chunk_cpu), delta_ulps)


def run_part(engine, gpu_call, oracle_call, delta_ulps=DELTA_ULPS):
    """The same for any part of the network: `gpu_call()` runs engine code (taped), `oracle_call()` the oracle's restatement of
    the same part on the same input; -> (gpu result, replayed oracle result, report)."""
    engine.tape = []
    try:
        with torch.no_grad():
            flows = gpu_call()
        torch.cuda.synchronize()
        tape = {}
        for name, t, layout in engine.tape:
            assert name not in tape, f"neuron layer recorded twice: {name}"
            tape[name] = (t, layout)
    finally:
        engine.tape = None
    report, used = [], set()

    def hook(prefix, x, s, ncfg, sd_):
        if prefix not in tape:
            report.append({"layer": prefix, "forced": False, "n": s.numel()})
            return s
        t, layout = tape[prefix]
        used.add(prefix)
        got = to_reference_layout(t, layout, x.shape).cpu().to(x.dtype)
        scale = max(float(x.pow(2).mean().sqrt()), abs(float(ncfg.v_th)) if ncfg.neuron_type != "psn" else 0.0, 1e-30)
        ulp = 2.0 ** -23 * scale
        r = O.delta_consistent(x, got, ncfg, sd_, prefix, delta_ulps * ulp)
        r.update(layer=prefix, forced=True, needed_ulps=r["needed"] / ulp, scale=scale)
        report.append(r)
        return got

    O.NEURON_HOOK = hook
    try:
        with torch.no_grad():
            ref = oracle_call()
    finally:
        O.NEURON_HOOK = None
    missing = set(tape) - used
    assert not missing, f"taped layers the oracle never asked for: {sorted(missing)[:5]}"
    return flows, ref, report


def summarise(report):
    forced = [r for r in report if r["forced"]]
    tot = sum(r["n"] for r in forced)
    return {"layers_forced": len(forced), "layers_free": len(report) - len(forced), "decisions": tot,
            "flips": sum(r["flips"] for r in forced), "ambiguous": sum(r["ambiguous"] for r in forced),
            "unexplained": sum(r["unexplained"] for r in forced),
            "needed_ulps_max": max((r["needed_ulps"] for r in forced), default=0.0)}
