"""The three `spikingjelly.activation_based.functional` tree walks the reference harness performs on the
model (eval_DSEC_flow_SNN.py:101-119,155), implemented over this package's neuron holders.  The HIP engine
is stateless between forwards (membranes start from v_reset inside every kernel), so reset_net only has to
restore the holders' `v`."""


class functional:
    @staticmethod
    def reset_net(net):
        for m in net.modules():
            if hasattr(m, "reset"):
                m.reset()

    @staticmethod
    def set_step_mode(net, step_mode):
        for m in net.modules():
            if hasattr(m, "step_mode"):
                if step_mode != "m" and type(m).__name__ == "PSN":
                    raise ValueError("PSN is multi-step only")
                m.step_mode = step_mode

    @staticmethod
    def set_backend(net, backend, instance=None):
        for m in net.modules():
            if instance is not None and not isinstance(m, instance):
                continue
            if hasattr(m, "backend") and backend in getattr(m, "supported_backends", ()):
                m.backend = backend   # 'cupy' requests are honoured by the HIP kernels (the only backend there is)
