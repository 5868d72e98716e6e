"""Public model classes of the drop-in boundary (mirror of reference
models/STSwinNet_SNN/Spiking_STSwinNet.py:254-325 and the kwargs plumbing of
models/STSwinNet/STSwinNet.py:323-366, models/STSwinNet_SNN/SNN_models.py:30-164).

    model = MS_SpikingformerFlowNet_en4(config["model"].copy(), config["swin_transformer"].copy())
    model.to("cuda"); model.init_weights(); model.eval()
    out = model(chunk)            # chunk (B, bins, 2, H, W)  ->  {"flow": [ (B,2,H,W) ] * E, "attn": None}

Forward runs on the HIP engine; there is no CPU fallback (it raises on CPU tensors).
"""
import torch
import torch.nn as nn

from .Spiking_modules import (MS_ResBlock, MS_SpikingPredLayer, MS_SpikingTransposeDecoderLayer, SEWResBlock, SpikingPredLayer,
                              SpikingTransposeDecoderLayer)
from .Spiking_submodules import IFNode
from .Spiking_swin_transformer3D import MS_Spiking_SwinTransformer3D_v2, Spiking_SwinTransformer3D_v2


class MS_spiking_former_encoder(nn.Module):
    """reference Spiking_STSwinNet.py:8-88."""
    swin_type = MS_Spiking_SwinTransformer3D_v2

    def __init__(self, arc_type="swinv2", patch_embed_type="PatchEmbedLocal", img_size=(240, 320), patch_size=(32, 2, 2),
                 in_chans=128, embed_dim=96, depths=(2, 2, 6), num_heads=(3, 6, 12), window_size=(2, 7, 7),
                 pretrained_window_size=(0, 0, 0), mlp_ratio=4.0, patch_norm=False, out_indices=(0, 1, 2), frozen_stages=-1,
                 norm=None, spikformer_norm=None, pol_in_channel=False, **spiking_kwargs):
        super().__init__()
        self.num_encoders = len(depths)
        self.out_channels = [embed_dim * 2 ** i for i in range(self.num_encoders)]
        self.swin3d = self.swin_type(
            arc_type=arc_type, embed_type=patch_embed_type, img_size=img_size, patch_size=patch_size, in_chans=in_chans,
            embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
            pretrained_window_size=pretrained_window_size, mlp_ratio=mlp_ratio, drop_rate=0.0, attn_drop_rate=0.0,
            drop_path_rate=0.2, norm_layer=spikformer_norm, out_indices=out_indices, frozen_stages=frozen_stages, norm=norm,
            **spiking_kwargs)


class spiking_former_encoder(MS_spiking_former_encoder):
    """SEW encoder (reference Spiking_STSwinNet.py:8-86)."""
    swin_type = Spiking_SwinTransformer3D_v2


class MS_Spikingformer_MultiResUNet(nn.Module):
    """Spiking U-Net: swin encoder, 2 MS res-blocks, transposed-conv decoders with per-scale flow predictions
    (reference Spiking_STSwinNet.py:90-252, SNN_models.py:118-164)."""
    encoder_block, res_type, transpose_type, pred_type = MS_spiking_former_encoder, MS_ResBlock, MS_SpikingTransposeDecoderLayer, MS_SpikingPredLayer

    def __init__(self, unet_kwargs, stt_kwargs):
        super().__init__()
        self.base_num_channels = unet_kwargs["base_num_channels"]
        self.num_encoders = unet_kwargs["num_encoders"]
        self.num_residual_blocks = unet_kwargs["num_residual_blocks"]
        self.num_output_channels = unet_kwargs["num_output_channels"]
        self.kernel_size = unet_kwargs["kernel_size"]
        if unet_kwargs["skip_type"] != "concat" or unet_kwargs.get("use_upsample_conv", True):
            raise NotImplementedError("shipped SNN configs use skip_type=concat with transposed-conv decoders")
        self.spiking_kwargs = dict(unet_kwargs["spiking_neuron"])
        self.steps = self.spiking_kwargs["num_steps"]
        self.num_bins_events = unet_kwargs["num_bins"]
        self.depths = [int(i) for i in stt_kwargs["swin_depths"]]
        self.num_heads = [int(i) for i in stt_kwargs["swin_num_heads"]]
        assert len(self.depths) == self.num_encoders and len(self.num_heads) == self.num_encoders
        self.window_size = [int(i) for i in stt_kwargs["window_size"]]
        self.input_size = stt_kwargs["input_size"]
        spik_norm = stt_kwargs["norm"] if "norm" in stt_kwargs else self.spiking_kwargs["spike_norm"]
        cm = unet_kwargs.get("channel_multiplier", 2)
        self.encoder_output_sizes = [int(self.base_num_channels * cm ** i) for i in range(self.num_encoders)]
        self.encoder_input_sizes = [self.base_num_channels] + self.encoder_output_sizes[:-1]
        self.max_num_channels = self.encoder_output_sizes[-1]
        kw = self.spiking_kwargs
        self.resblocks = nn.ModuleList([self.res_type(self.max_num_channels, self.max_num_channels, connect_function="ADD", **kw)
                                        for _ in range(self.num_residual_blocks)])
        self.decoders = nn.ModuleList()
        for i, (cin, cout) in enumerate(zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes))):
            self.decoders.append(self.transpose_type(2 * cin + (0 if i == 0 else self.num_output_channels), cout,
                                                                 kernel_size=self.kernel_size, scale=2, **kw))
        self.preds = nn.ModuleList([self.pred_type(c, self.num_output_channels, 1, **kw)
                                    for c in reversed(self.encoder_input_sizes)])
        self.encoders = self.encoder_block(
            arc_type=stt_kwargs["use_arc"][0], patch_embed_type=stt_kwargs["use_arc"][1], img_size=self.input_size,
            patch_size=[int(i) for i in stt_kwargs["swin_patch_size"]], in_chans=self.num_bins_events,
            embed_dim=self.base_num_channels, depths=self.depths, num_heads=self.num_heads, window_size=self.window_size,
            pretrained_window_size=[int(i) for i in stt_kwargs["pretrained_window_size"]], mlp_ratio=stt_kwargs["mlp_ratio"],
            out_indices=[int(i) for i in stt_kwargs["swin_out_indices"]], norm=None, spikformer_norm=spik_norm,
            pol_in_channel=False, **kw)
        self.preds_out = nn.ModuleList([IFNode(v_threshold=float("inf"), v_reset=0.0) for _ in range(self.num_encoders)])

    def record_flops(self):
        """Analytic MAC record of the reference (Spiking_STSwinNet.py:211-237)."""
        rec = {"en": self.encoders.swin3d.record_flops()}
        H, W = self.encoders.swin3d.patch_embed.patches_resolution
        H, W = H // 2 ** (self.num_encoders - 1), W // 2 ** (self.num_encoders - 1)
        C = self.max_num_channels
        for i in range(self.num_residual_blocks):
            rec[f"res{i}conv0"] = rec[f"res{i}conv1"] = C * C * 9 * H * W
        for i, (cin, cout) in enumerate(zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes))):
            H, W = 2 * H, 2 * W
            rec[f"decoder{i}"] = (2 * cin + (0 if i == 0 else self.num_output_channels)) * cout * H * W * self.kernel_size ** 2
            rec[f"pred{i}"] = cout * self.num_output_channels * H * W
        return rec

    def flops(self):
        """reference :184-209: encoder + res-blocks + decoders + predictions, one MAC per BN output element."""
        f = self.encoders.swin3d.flops()
        H, W = self.encoders.swin3d.patch_embed.patches_resolution
        H, W = H // 2 ** (self.num_encoders - 1), W // 2 ** (self.num_encoders - 1)
        f += 2 * self.max_num_channels ** 2 * 9 * H * W * self.num_residual_blocks
        for i, (cin, cout) in enumerate(zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes))):
            H, W = 2 * H, 2 * W
            f += (2 * cin + (0 if i == 0 else self.num_output_channels)) * cout * H * W * self.kernel_size ** 2 + cout * H * W
            f += cout * self.num_output_channels * H * W + self.num_output_channels * H * W
        return f


class Spikingformer_MultiResUNet(MS_Spikingformer_MultiResUNet):
    """SEW U-Net (reference Spiking_STSwinNet.py:90-182; SNN_models.py:12-32): SEW encoder, SEW res-blocks, transposed-conv
    decoders with the neuron after the norm, plain 1x1 predictions."""
    encoder_block, res_type, transpose_type, pred_type = spiking_former_encoder, SEWResBlock, SpikingTransposeDecoderLayer, SpikingPredLayer


class MS_SpikingformerFlowNet(nn.Module):
    """MS-shortcut SDformerFlow, 3 encoders (reference Spiking_STSwinNet.py:313-317)."""
    num_en = 3
    unet_type = MS_Spikingformer_MultiResUNet

    def __init__(self, unet_kwargs, stt_kwargs):
        super().__init__()
        unet_kwargs = dict(unet_kwargs)
        self.mask = unet_kwargs.get("mask_output")
        self.norm_input = unet_kwargs.get("norm_input", False)
        self.encoding = unet_kwargs.get("encoding")
        self.num_bins = unet_kwargs["num_bins"]
        self.num_encoders = self.num_en
        unet_kwargs.update({"num_encoders": self.num_en, "num_residual_blocks": 2, "num_output_channels": 2,
                            "skip_type": "concat", "channel_multiplier": 2,
                            "use_upsample_conv": unet_kwargs.get("use_upsample_conv", True)})
        self.sttmultires_unet = self.unet_type(unet_kwargs, dict(stt_kwargs))
        self._engine, self._stamp_tensors = None, None
        # Weight planes of the spike GEMMs / convolutions (binary spikes are exact in 16-bit floats, accumulation is fp32):
        #   2 = fp16 hi + lo of the power-of-two-scaled weight: 22 of the 24 significand bits at 2/3 of the matrix work.
        #       Measured against fp64 the layer outputs are as close as torch's own fp32 convolution, and every
        #       teacher-forced stage reproduces the oracle as well as the exact-weight mode does (tests/test_engine_gpu.py).
        #   3 = bf16 hi + mid + lo: the fp32 weights exactly.
        self.gemm_nsplit = 2

    def init_weights(self):
        """Linear: kaiming-normal fan_out; BN: 1/0; Conv2d: xavier-uniform (reference :264-276)."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
        self.apply(_init)
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine, self._stamp_tensors = None, None
        return super()._apply(fn, *a, **k)

    def _weights_stamp(self):
        """Cheap version stamp of everything the packed engine copies: in-place updates (optimizer.step(), the train-mode
        BN kernels' running statistics, load_state_dict's copy_()) bump a tensor's `_version`.  Host-side only (no device
        sync, ~0.1 ms); the tensor list is cached and dropped whenever the tree is re-laid (`_apply`: .to() / .cuda())."""
        if self._stamp_tensors is None:
            self._stamp_tensors = list(self.parameters()) + list(self.buffers())
        # (data_ptr, version): `p.data = other` moves the pointer, in-place updates bump the version.  What NO stamp can see: writes
        # through `p.data.copy_()` / `p.data.mul_()` (EMA weight swaps) and a Parameter OBJECT replaced in a submodule - call
        # invalidate_engine() after those.
        return (self.gemm_nsplit,) + tuple((t.data_ptr(), t._version) for t in self._stamp_tensors)

    def invalidate_engine(self):
        """For callers that swap tensors out behind the module's back (`p.data = ...`), which no version counter sees."""
        self._engine, self._stamp_tensors = None, None

    def train(self, mode=True):
        """Entering training invalidates the packed inference plan (weights and BN statistics are about to change)."""
        if mode:
            self._engine = None
        return super().train(mode)

    def engine(self):
        """The packed HIP execution plan: split weight planes, folded eval-BN (alpha, beta), PSN matrices.  Rebuilt whenever
        the weights moved or changed since it was packed - `load_state_dict`, `.to()`, `init_weights`, `train()`, and any
        in-place update seen through the version stamp (eval -> train_step -> eval must not run on stale planes)."""
        stamp = self._weights_stamp()
        if self._engine is None or self._engine_stamp != stamp:
            self._engine = self._make_engine()
            self._engine_stamp = stamp
        return self._engine

    def _make_engine(self):
        from ..engine import MSFlowEngine
        return MSFlowEngine(self)

    def flops(self):
        """reference :307-308."""
        return self.sttmultires_unet.flops()

    def record_flops(self):
        """reference :310-311."""
        return self.sttmultires_unet.record_flops()

    def forward(self, x, log=False):
        if self.training:                     # train-mode forward under autograd (batch-stat BN, HIP neurons both ways)
            from ..train import forward_train
            return {"flow": forward_train(self, x), "attn": None}
        with torch.no_grad():
            scores = [] if log else None          # reference :283-284; see engine.forward for what the reference's own chain does
            flows = self.engine().forward(x, scores)
        return {"flow": flows, "attn": scores}

    def forward_replicas(self, x):
        """x (R, bins, 2, H, W) = R INDEPENDENT samples -> {"flow": [(R, 2, H, W)] * E}: flow[i] is bit-equal to
        `self(x[i:i+1])["flow"]` - R batch-1 forwards of the reference (`eval_DSEC_flow_SNN.py:219` with `batch_size: 1`) served by one
        launch sequence whose kernels see R times the rows.  (`self(x)` on a batch keeps the reference's own batch semantics, which
        couples the samples through `window_partition_v2`'s raw view.)  Eval mode only."""
        if self.training:
            raise RuntimeError("forward_replicas is an inference entry point: call model.eval()")
        from .. import hip

        def one_by_one():
            outs = [self.engine().forward(x[i:i + 1], None) for i in range(x.shape[0])]
            return [torch.cat([o[lvl] for o in outs], 0) for lvl in range(len(outs[0]))]
        with torch.no_grad():
            if self.gemm_nsplit != 2 and x.shape[0] > 1:
                # the 16-bit-plane modes (3 = exact, 1 = bf16) run on the streaming kernels, whose split-K plans and tile families follow
                # the row count: R-fold rows would change the summation order of a layer - the samples go one by one (same results)
                flows = one_by_one()
            else:
                try:
                    flows = self.engine().forward(x, None, replicas=True)
                except hip.ReplicaGeometryError:
                    # an odd window count per sample at some stage (e.g. 256 x 320: 175 windows at stage 0): the replica tables cannot
                    # keep a sample's head scramble inside one attention step - one by one (nothing persistent was touched before the raise)
                    flows = one_by_one()
        return {"flow": flows, "attn": None}


class MS_SpikingformerFlowNet_en4(MS_SpikingformerFlowNet):
    """MS-shortcut SDformerFlow, 4 encoders - the shipped model (reference :319-325)."""
    num_en = 4


class SpikingformerFlowNet(MS_SpikingformerFlowNet):
    """SEW-shortcut SDformerFlow, 3 encoders (reference Spiking_STSwinNet.py:254-311): the stream between blocks carries sums
    of spikes.  Inference runs on `engine_sew.SEWFlowEngine`; training of this family is not built."""
    num_en = 3
    unet_type = Spikingformer_MultiResUNet

    def _make_engine(self):
        from ..engine_sew import SEWFlowEngine
        return SEWFlowEngine(self)

    def forward(self, x, log=False):
        if self.training:
            raise NotImplementedError("the SEW family is forward-only here (the training path covers the shipped MS models)")
        if log:
            raise NotImplementedError("SEW attention maps (B_, nH, N, N) live in registers of the fused window-attention kernel and "
                                      "are never materialised; log=True is built for the MS (QK token-gate) family")
        return super().forward(x, log)

    def forward_replicas(self, x):
        """The SEW engine has no one-launch-sequence form (its window attention and integer stream keep the reference's batch view): the
        samples go one by one - same contract as the MS models' call, flow[i] == self(x[i:i+1])["flow"]."""
        if self.training:
            raise RuntimeError("forward_replicas is an inference entry point: call model.eval()")
        with torch.no_grad():
            outs = [self.engine().forward(x[i:i + 1], None) for i in range(x.shape[0])]
        return {"flow": [torch.cat([o[lvl] for o in outs], 0) for lvl in range(len(outs[0]))], "attn": None}
