"""Parameter tree of the spiking conv / patch-embed layers (mirror of reference
models/STSwinNet_SNN/Spiking_modules.py; attribute names fix the state_dict keys of SURVEY.md 8b).

The layers are declarative: their forward arithmetic is scheduled by `sdformerflow_amd.engine`,
which fuses BN into the neuron kernels and never materialises the reference's permutes.
"""
import torch.nn as nn

from .Spiking_submodules import LIFNode, IFNode, PSN, ParametricLIFNode, SLTTLIFNode, GatedLIFNode


class _Surrogate:
    """Namespace so the YAML string `surrogate.ATan()` (configs/*.yml `surrogate_fun`) evaluates."""

    class ATan:
        def __init__(self, alpha=2.0, spiking=True):
            self.alpha, self.spiking = alpha, spiking

    class Sigmoid(ATan):
        def __init__(self, alpha=4.0, spiking=True):
            super().__init__(alpha, spiking)


surrogate = _Surrogate


class Spiking_neuron(nn.Module):
    """reference Spiking_modules.py:26-99 - the `neuron_type` switch.  lif / if / psn / plif / SLTTlif run on the HIP kernels
    (plif: the multiplicative charge through the `tau` field; SLTTlif: LIF forward); glif is torch element-wise ops on the
    tensor's device (module-level only - the fused engines refuse it)."""

    def __init__(self, num_steps, spike_norm=None, neuron_type="plif", v_th=1.0, v_reset=0, surrogate_fun="surrogate.ATan()",
                 tau=2.0, detach_reset=True):
        super().__init__()
        fn = eval(surrogate_fun) if isinstance(surrogate_fun, str) else surrogate_fun
        if neuron_type == "lif":
            self.spiking_neuron = LIFNode(tau=tau, v_threshold=v_th, v_reset=v_reset, surrogate_function=fn,
                                          detach_reset=detach_reset)
        elif neuron_type == "if":
            self.spiking_neuron = IFNode(v_threshold=v_th, v_reset=v_reset, surrogate_function=fn, detach_reset=detach_reset)
        elif neuron_type == "psn":
            self.spiking_neuron = PSN(T=num_steps, surrogate_function=fn)
        elif neuron_type == "SLTTlif":
            self.spiking_neuron = SLTTLIFNode(tau=tau, v_threshold=v_th, v_reset=v_reset, surrogate_function=fn,
                                              detach_reset=detach_reset)
        elif neuron_type == "plif":
            self.spiking_neuron = ParametricLIFNode(init_tau=tau, v_threshold=v_th, v_reset=v_reset, surrogate_function=fn,
                                                    detach_reset=detach_reset)
        elif neuron_type == "glif":
            self.spiking_neuron = GatedLIFNode(T=num_steps, init_v_subreset=None, init_tau=0.25, init_v_threshold=0.5,
                                               init_conduct=0.5, surrogate_function=fn)
        else:
            raise ValueError(f"neuron type {neuron_type!r} not in the list (lif, if, plif, SLTTlif, glif, psn)")

    def forward(self, x):
        return self.spiking_neuron(x)


class SpikingNormLayer(nn.Module):
    """reference Spiking_modules.py:101-146; only the shipped 'BN' variant is on the hot path."""

    def __init__(self, out_channels, num_steps=None, norm="BN", v_th=1.0):
        super().__init__()
        if norm != "BN":
            raise NotImplementedError(f"spike_norm {norm!r}: only 'BN' is used by the shipped configs")
        self.num_steps, self.norm = num_steps, norm
        self.norm_layer = nn.BatchNorm2d(out_channels)


def _neuron_kwargs(kw):
    return {k: v for k, v in kw.items() if k != "spike_norm"}


class SpikingConvEncoderLayer(nn.Module):
    """conv -> BN -> SN (reference :250-296)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, spike_norm=None, **spiking_kwargs):
        super().__init__()
        self.norm = spike_norm
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=spike_norm is None))
        if spike_norm is not None:
            self.norm_layer = SpikingNormLayer(out_channels, spiking_kwargs["num_steps"], spike_norm, spiking_kwargs["v_th"])
        self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))


class MS_SpikingConvEncoderLayer(nn.Module):
    """[SN ->] conv -> BN, membrane shortcut flavour (reference :298-347)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, first_layer=True, spike_norm=None,
                 **spiking_kwargs):
        super().__init__()
        self.first_layer, self.norm = first_layer, spike_norm
        if not first_layer:
            self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=spike_norm is None))
        if spike_norm is not None:
            self.norm_layer = SpikingNormLayer(out_channels, spiking_kwargs["num_steps"], spike_norm, spiking_kwargs["v_th"])


class MS_ResBlock(nn.Module):
    """SN-conv-BN-SN-conv-BN + identity (reference :880-933)."""

    def __init__(self, in_channels, out_channels, stride=1, connect_function="ADD", spike_norm=None, **spiking_kwargs):
        super().__init__()
        self.norm, self.connect_function = spike_norm, connect_function
        bias = spike_norm is None
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, stride, 1, bias=bias))
        self.conv2 = nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, 1, 1, bias=bias))
        if spike_norm is not None:
            self.norm1 = SpikingNormLayer(out_channels, None, "BN", spiking_kwargs["v_th"])
            self.norm2 = SpikingNormLayer(out_channels, None, "BN", spiking_kwargs["v_th"])
        self.sn1 = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))
        self.sn2 = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))

    def forward(self, x):
        """(T,B,C,H,W) membrane -> same shape, on the HIP engine (module_forward.ms_resblock_forward)."""
        from ..module_forward import ms_resblock_forward
        return ms_resblock_forward(self, x)


class MS_spiking_residual_feature_generator(nn.Module):
    """reference :935-973."""

    def __init__(self, dim, norm, num_resblocks=4, cnt_fun="ADD", **spiking_kwargs):
        super().__init__()
        self.dim, self.num_resblocks = dim, num_resblocks
        self.resblocks = nn.ModuleList([MS_ResBlock(dim, dim, 1, cnt_fun, norm, **spiking_kwargs)
                                        for _ in range(num_resblocks)])


class SpikingPEDLayer(nn.Module):
    """stride-2 projection with a 1x1 membrane shortcut (reference :772-825)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, norm=None, patch_resolution=(120, 160),
                 **spiking_kwargs):
        super().__init__()
        self.norm, self.patch = norm, patch_resolution
        bias = norm is None
        self.conv_res = nn.Conv2d(in_channels, out_channels, 1, 2, 0, bias=bias)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, 1, bias=bias)
        if norm is not None:
            self.norm_layer = nn.BatchNorm2d(out_channels)
        self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))


class MS_PED_Spiking_PatchEmbed_Conv_sfn(nn.Module):
    """The shipped patch embedding (reference :1710-1790): head conv, stride-2 conv, 2 MS res-blocks, PED projection."""
    use_MS, num_res, first_conv_k = True, 2, 3

    def __init__(self, img_size=(240, 320), patch_size=(2, 4, 4), in_chans=10, embed_dim=96, patch_norm=None, norm=None,
                 spiking_proj=False, spike_norm=None, **spiking_kwargs):
        super().__init__()
        self.patch_size, self.image_size = patch_size, img_size
        self.patches_resolution = [img_size[0] // patch_size[2] // 2, img_size[1] // patch_size[3] // 2]
        self.embed_dim, self.patch_norm, self.num_bins = embed_dim, patch_norm, in_chans
        self.num_steps = spiking_kwargs["num_steps"]
        self.num_ch = in_chans * 2 // self.num_steps
        self.spike_norm = spike_norm
        kw = dict(spiking_kwargs)
        self.head = SpikingConvEncoderLayer(self.num_ch, embed_dim // 2, 3, 1, 1, spike_norm=spike_norm, **kw)
        self.conv = MS_SpikingConvEncoderLayer(embed_dim // 2, embed_dim, self.first_conv_k, 2, self.first_conv_k // 2,
                                               spike_norm=spike_norm, **kw)
        self.residual_encoding = MS_spiking_residual_feature_generator(embed_dim, spike_norm, self.num_res, "ADD", **kw)
        self.proj = SpikingPEDLayer(embed_dim, embed_dim, 3, tuple(patch_size[2:]), 1, norm=spike_norm,
                                    patch_resolution=self.patches_resolution, **kw)

    def record_flops(self):
        """reference :1816-1840."""
        E, (H, W), (ph, pw) = self.embed_dim, self.image_size, self.patches_resolution
        rec = {"head": self.num_ch * E // 2 * 9 * H * W,
               "conv": E // 2 * E * self.first_conv_k * self.first_conv_k * H * W // 4}
        for i in range(self.num_res):
            rec[f"res{i}_conv0"] = rec[f"res{i}_conv1"] = E * E * 9 * H * W // 4
        rec["proj"] = E * E * 9 * ph * pw
        return rec

    def flops(self):
        """reference :1795-1814: the convolutions above plus one MAC per BN output element."""
        E, (H, W), (ph, pw) = self.embed_dim, self.image_size, self.patches_resolution
        bn = E // 2 * H * W + E * H * W + self.num_res * 2 * E * H * W // 4 + E * ph * pw
        return sum(self.record_flops().values()) + bn


class MS_SpikingTransposeDecoderLayer(nn.Module):
    """SN -> ConvTranspose 3x3 s2 -> BN (reference :397-474)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, spike_norm=None, scale=2, **spiking_kwargs):
        super().__init__()
        if scale != 2:
            raise NotImplementedError("scale-4 decoder is not used by the shipped configs")
        self.scale, self.norm = scale, spike_norm
        self.deconv = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=2, padding=kernel_size // 2,
                                                       output_padding=1, bias=spike_norm is None))
        if spike_norm is not None:
            self.norm_layer = SpikingNormLayer(out_channels, spiking_kwargs["num_steps"], spike_norm, spiking_kwargs["v_th"])
        self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))


class MS_SpikingPredLayer(nn.Module):
    """SN -> conv1x1 (+bias) flow prediction (reference :607-646)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, **spiking_kwargs):
        super().__init__()
        self.norm = None
        self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=True))


# ---------------------------------------------------------------------------------------------- SEW family (reference :827-878, :397-456, :571-603)
class SEWResBlock(MS_ResBlock):
    """conv-BN-SN-conv-BN-SN + identity (spike-element-wise ADD; reference :827-878).  Same parameters as `MS_ResBlock`; the
    order of neuron and convolution differs, which the SEW engine schedules."""

    def forward(self, x):
        raise NotImplementedError("SEWResBlock runs inside SEWFlowEngine (the whole-model forward); it has no module-level forward")


class SpikingTransposeDecoderLayer(MS_SpikingTransposeDecoderLayer):
    """ConvTranspose 3x3 s2 -> BN -> SN (reference :397-456): the neuron comes last."""


class SpikingPredLayer(nn.Module):
    """Plain conv1x1 (+bias) flow prediction on spikes (reference :571-603): no neuron of its own."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, **spiking_kwargs):
        super().__init__()
        self.norm = None
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=True))
