#!/usr/bin/env python3
"""Where does a replica forward leave the separate batch-1 forwards?  Stage by stage on the engine's own methods.
usage: replica_debug.py [lif|psn] R [H W] [en3] [T20]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_replica_batch import _model
from sdformerflow_amd.harness import prepare_chunk
from sdformerflow_amd.synthetic import synth_voxel
kind, R = sys.argv[1], int(sys.argv[2])
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (288, 384)
en4 = "en3" not in sys.argv
TT = 20 if "T20" in sys.argv else 10
model = _model(kind, H, W, en4, T=TT)
eng = model.engine()
xs = [prepare_chunk(synth_voxel(1, TT, H, W, seed=300 + 7 * i)).to("cuda:0") for i in range(R)]


def cmp(tag, rep, ones):
    bad = [i for i in range(R) if not torch.equal(rep[i], ones[i][0])]
    d = max(float((rep[i].float() - ones[i][0].float()).abs().max()) for i in range(R))
    print(f"{tag:40s} {'ok' if not bad else 'DIFFERS in samples ' + str(bad)}  max |diff| {d:.3g}", flush=True)
    return not bad


with torch.no_grad():
    ones = [eng.patch_embed(x) for x in xs]
    eng.replicas = True
    rep = eng.patch_embed(torch.cat(xs, 0))
    cmp("patch_embed", rep, ones)
    rep = torch.cat(ones, 0).clone()                                     # continue from identical inputs at every step
    feats_r, feats_1 = [], [[] for _ in range(R)]
    for s, blocks in enumerate(eng.stages):
        for i in range(len(blocks)):
            eng.replicas = False
            ones = [eng.swin_block(o.clone(), s, i) for o in ones]
            eng.replicas = True
            a = eng.attention(rep.clone(), blocks[i])
            eng.replicas = False
            a1 = [eng.attention(o.clone(), blocks[i]) for o in [rep[j:j + 1] for j in range(R)]]
            cmp(f"stage {s} block {i} attention (from rep)", a, a1)
            eng.replicas = True
            rep = eng.swin_block(rep, s, i)
            cmp(f"stage {s} block {i}", rep, ones)
            rep = torch.cat(ones, 0).clone()
        feats_r.append(rep)
        for j in range(R):
            feats_1[j].append(ones[j])
        if s < len(eng.merges):
            eng.replicas = False
            ones = [eng.patch_merge(o, s) for o in ones]
            eng.replicas = True
            rep = eng.patch_merge(rep, s)
            cmp(f"merge {s}", rep, ones)
            rep = torch.cat(ones, 0).clone()
    # the emitted-spike hand-overs of forward(): last MLP of a stage -> patch merging / the bottleneck's first neuron
    for s, blocks in enumerate(eng.stages):
        nxt_sn = eng.merges[s][1] if s < len(eng.merges) else eng.unet_res[0].sn1
        y1 = [feats_1[j][s].clone() for j in range(R)]
        yr = torch.cat(y1, 0).clone()
        eng.replicas = False
        em1 = [eng._next_spikes(y, blocks[-1], nxt_sn) for y in y1]
        o1 = [eng.swin_block(y, s, len(blocks) - 1, **({"emit_next": e} if e is not None else {})) for y, e in zip(y1, em1)]
        eng.replicas = True
        emr = eng._next_spikes(yr, blocks[-1], nxt_sn)
        orr = eng.swin_block(yr, s, len(blocks) - 1, **({"emit_next": emr} if emr is not None else {}))
        cmp(f"stage {s} last block again (emitting)", orr, o1)
        if emr is not None and all(e is not None for e in em1):
            cmp(f"stage {s} emitted spikes", emr[0], [e[0] for e in em1])
            if s < len(eng.merges):
                eng.replicas = False
                m1 = [eng.patch_merge(o, s, spikes=e[0]) for o, e in zip(o1, em1)]
                eng.replicas = True
                mr = eng.patch_merge(orr, s, spikes=emr[0])
                cmp(f"merge {s} on emitted spikes", mr, m1)
        else:
            print(f"stage {s}: emission rep={emr is not None} single={[e is not None for e in em1]}")
    eng.replicas = False
    tails = [eng.unet_tail(f, out_size=(H, W)) for f in feats_1]
    flows1 = None
    eng.replicas = True
    pr = eng.unet_tail(feats_r, out_size=(H, W))
    fr = list(eng._flows)
    eng.replicas = False
    for lvl in range(len(fr)):
        f1 = []
        for j in range(R):
            eng.unet_tail(feats_1[j], out_size=(H, W))
            f1.append(eng._flows[lvl])
        if fr[lvl] is not None:
            cmp(f"tail flow level {lvl}", fr[lvl], f1)
