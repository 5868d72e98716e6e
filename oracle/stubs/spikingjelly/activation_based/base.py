"""Stub of spikingjelly.activation_based.base (own code)."""
import copy
import torch
import torch.nn as nn


class StepModule:
    def supported_step_mode(self):
        return ("s", "m")

    @property
    def step_mode(self):
        return getattr(self, "_step_mode", "s")

    @step_mode.setter
    def step_mode(self, value):
        if value not in self.supported_step_mode():
            raise ValueError(f"step_mode {value!r} not supported")
        self._step_mode = value


class SingleModule(StepModule):
    def supported_step_mode(self):
        return ("s",)


class MultiStepModule(StepModule):
    # tolerant of __init__ never being called (PSN only runs nn.Module.__init__)
    def supported_step_mode(self):
        return ("m",)

    @property
    def step_mode(self):
        return getattr(self, "_step_mode", "m")

    @step_mode.setter
    def step_mode(self, value):
        if value != "m":
            raise ValueError("multi-step only")
        self._step_mode = value


class MemoryModule(nn.Module, StepModule):
    def __init__(self):
        super().__init__()
        self._memories = {}
        self._memories_rv = {}
        self._backend = "torch"
        self._step_mode = "s"

    @property
    def supported_backends(self):
        return ("torch",)

    @property
    def backend(self):
        return self._backend

    @backend.setter
    def backend(self, value):
        if value not in self.supported_backends:
            raise NotImplementedError(value)
        self._backend = value

    def single_step_forward(self, x, *a, **k):
        raise NotImplementedError

    def multi_step_forward(self, x_seq, *a, **k):
        ys = [self.single_step_forward(x_seq[t], *a, **k) for t in range(x_seq.shape[0])]
        return torch.stack(ys)

    def forward(self, *a, **k):
        if self.step_mode == "s":
            return self.single_step_forward(*a, **k)
        return self.multi_step_forward(*a, **k)

    def register_memory(self, name, value):
        self._memories[name] = value
        self._memories_rv[name] = copy.deepcopy(value)

    def reset(self):
        for k in self._memories.keys():
            self._memories[k] = copy.deepcopy(self._memories_rv[k])

    def __getattr__(self, name):
        d = self.__dict__
        if "_memories" in d and name in d["_memories"]:
            return d["_memories"][name]
        return super().__getattr__(name)

    def __setattr__(self, name, value):
        mem = self.__dict__.get("_memories")
        if mem is not None and name in mem:
            mem[name] = value
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):
        mem = self.__dict__.get("_memories")
        if mem is not None and name in mem:
            del mem[name]
            del self._memories_rv[name]
        else:
            super().__delattr__(name)
